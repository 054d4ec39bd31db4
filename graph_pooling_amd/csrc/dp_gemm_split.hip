// Batched GEMM between two GENERAL fp32 operands on the bf16 matrix cores, at fp32-grade accuracy.
//
//   C[b] = op(A[b]) op(B[b]) (+ C[b])
//
// The contractions of the pooling step and the assign head at large shapes — A' = (A^T S)^T S, dS += T dA',
// V = S dA'^T, the assign_pred products (encoders.py:1273, 1279 and their backward; ~1.2 GFLOP per graph at
// N = 1024, K = 256) — have no exactly-representable operand (unlike the 0/1 adjacency), so they ran on the fp32
// MFMA (v_mfma_f32_16x16x4_f32: 157 TFLOP/s peak, 74-84 measured).  Here BOTH operands are split in registers,
// on the way from HBM to LDS, into three bf16 planes  x = h + m + l  (exactly, round-to-nearest at each step),
// and the product is taken as the six plane products whose weight is >= 2^-16 of the leading one:
//
//   a b  ~=  h_a h_b + (h_a m_b + m_a h_b) + (h_a l_b + l_a h_b + m_a m_b)
//
// Each bf16 x bf16 product is exact in fp32 and is accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (16x the fp32
// MFMA rate, so 16/6 of its throughput); the dropped terms (m l, l m, l l) are <= 2^-23 |a b|, the size of one fp32
// rounding.  Results are deterministic; they differ from the fp32-MFMA kernel's by reduction order and those terms.
//
// Tile 128 x 128 x 32 per workgroup (4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 MFMA tiles, 96 MFMAs per wave and
// k-slab).  LDS image of either operand: [plane][row][k], k contiguous, row stride 40 bf16 (80 B: the 16 rows of a
// fragment read start 20 banks apart, every ds_read_b128 conflict-free).  Global loads keep the operand's own
// orientation coalesced:
//   k-contiguous operand (A not transposed / B transposed): a thread owns four (row, k..k+3) quads;
//   row-contiguous operand (A transposed / B not transposed): a thread owns ONE 4 k x 4 row block, loaded as four
//     16-byte row segments and transposed in registers — so every LDS write is an 8-byte (row, k..k+3) store in both
//     cases, never a 2-byte scatter.
// The next slab's global loads are in flight while the current one is multiplied (register double buffer).
#include "dp_common.h"

namespace dp {

namespace {

typedef float sg_f32x4 __attribute__((ext_vector_type(4)));
typedef float sg_f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef short sg_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short sg_u16x4 __attribute__((ext_vector_type(4)));

constexpr int SG_BM = 128, SG_BN = 128, SG_KT = 32;
constexpr int SG_LDK = 40;                          // bf16 elements per LDS row
constexpr int SG_PLANE = SG_BM * SG_LDK;            // elements of one plane of a 128-row operand tile
constexpr int sg_lds_bytes(int bn) { return 3 * (SG_BM + bn) * SG_LDK * 2; }   // A and B tiles, three planes, 2 bytes
constexpr int SG_LDS_BYTES = sg_lds_bytes(SG_BN);

struct SplitGemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    int lda, ldb, ldc;
    long sA, sB, sC;
    int tA, tB;
    float beta;       // 0 or 1
    int tilesN;
    // optional second output: the exact 3-plane bf16 split of C in the layout the bf16 aggregation reads (GemmDesc)
    unsigned short* split_out;
    int split_ct, split_k8, split_c0;
    const float* bias;   // [N] added to every row, or null
    int act;             // 1: ReLU
};

// 16 fp32 values of one operand slab share: q[i] = four consecutive elements along the operand's contiguous dimension
template <int NQ>
struct SgRegs {
    sg_f32x4 q[NQ];
};

// ---- global -> registers.  KC: k is the contiguous dimension (element (r, k) at p[r * ld + k]); else p[k * ld + r].
// Every load is ONE unconditional 16-byte request from a clamped, always-valid address — no branch and no select sits
// between the eight requests of a slab, so they share one memory round trip (a per-quad `if (inside) load4 else
// load1 x 4` made hipcc wait for each quad in turn: 9.8k cycles per k-slab against a 3k MFMA budget).
// EDGE = false: the whole 128 x 32 slab lies inside the operand.  EDGE = true (tiles on the operand's rim, the last
// partial k-slab): a quad that sticks out of the contiguous dimension is read shifted back so that it ENDS at the
// rim, and sg_fix() — after the loads are in flight — undoes the shift and zeroes what lies outside.
// NQ = quads per thread and operand slab: 4 with 256 threads, 2 with 512.
// NT = threads of the workgroup (256 or 512); a 128-row tile is NQ = 1024 / NT quads per thread, a 64-row tile half that.
template <bool KC, int NQ, int NT>
__device__ __forceinline__ void sg_coords(int i, int r0, int k0, int& c, int& o) {
    const int t = threadIdx.x;
    int r, k;
    if (KC) {
        const int slot = t + NT * i;            // (row, k-quad): 8 quads per row
        r = r0 + (slot >> 3);
        k = k0 + (slot & 7) * 4;
    } else if (NQ == 4) {
        r = r0 + (t >> 3) * 4;                  // one 4 k x 4 row block per thread: k-block t & 7, row block t >> 3
        k = k0 + (t & 7) * 4 + i;
    } else if (NQ == 2) {
        r = r0 + (t >> 4) * 4;                  // one 2 k x 4 row block per thread: k-pair t & 15, row block t >> 4
        k = k0 + (t & 15) * 2 + i;
    } else {
        r = r0 + (t >> 5) * 4;                  // one 1 k x 4 row block per thread: k t & 31, row block t >> 5
        k = k0 + (t & 31);
    }
    c = KC ? k : r;                             // contiguous coordinate of the quad's first element
    o = KC ? r : k;                             // the other coordinate
}
template <bool KC, bool EDGE, int NQ, int NT>
__device__ __forceinline__ void sg_load(const float* __restrict__ p, int ld, int r0, int k0, int rmax, int kmax,
                                        SgRegs<NQ>& s) {
    const int cmax = KC ? kmax : rmax, omax = KC ? rmax : kmax;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int c, o;
        sg_coords<KC, NQ, NT>(i, r0, k0, c, o);
        if (EDGE) {
            c = min(c, cmax - 4);
            o = min(o, omax - 1);
        }
        s.q[i] = *reinterpret_cast<const sg_f32x4_u*>(p + (long)o * ld + c);
    }
}
template <bool KC, int NQ, int NT>
__device__ __forceinline__ void sg_fix(int r0, int k0, int rmax, int kmax, SgRegs<NQ>& s) {
    const int cmax = KC ? kmax : rmax, omax = KC ? rmax : kmax;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int c, o;
        sg_coords<KC, NQ, NT>(i, r0, k0, c, o);
        const int shift = c - min(c, cmax - 4);          // 0 inside; 1..3 on the rim; >= 4 wholly outside
        const bool dead = o >= omax;
        const sg_f32x4 v = s.q[i];
        sg_f32x4 w;
        w[0] = shift == 0 ? v[0] : shift == 1 ? v[1] : shift == 2 ? v[2] : shift == 3 ? v[3] : 0.f;
        w[1] = shift == 0 ? v[1] : shift == 1 ? v[2] : shift == 2 ? v[3] : 0.f;
        w[2] = shift == 0 ? v[2] : shift == 1 ? v[3] : 0.f;
        w[3] = shift == 0 ? v[3] : 0.f;
        s.q[i] = dead ? (sg_f32x4){0.f, 0.f, 0.f, 0.f} : w;
    }
}

// v = h + m + l exactly: round to bf16 (gfx950's v_cvt_pk_bf16_f32, round-to-nearest-even), subtract — the residual
// of a nearest rounding is exact in fp32 — and repeat.  Three conversions, two shifts back and two subtractions per
// element (the integer formulation of dp_common.h's bf16_split3 is ~16 vector instructions: at 32 elements per thread
// and k-slab it, not the MFMA, set the pace of this kernel).
__device__ __forceinline__ void sg_split(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 hb = (__bf16)v;
    const float r1 = v - (float)hb;
    const __bf16 mb = (__bf16)r1;
    const float r2 = r1 - (float)mb;
    const __bf16 lb = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, hb);
    m = __builtin_bit_cast(unsigned short, mb);
    l = __builtin_bit_cast(unsigned short, lb);
}

// Two values at once, the way the hardware converts them: v_cvt_pk_bf16_f32 rounds a PAIR into one packed word (the
// LDS image wants consecutive k packed anyway), the two halves widen back with one shift and one mask, and the
// residuals of a pair are one v_pk_add_f32.  9 vector instructions per pair; the element-wise form above compiled to
// 7-8 per ELEMENT plus the packing (cvt_pk with an unused half, v_or_b32_sdwa to assemble the word).
typedef float sg_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sg_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sg_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    sg_f32x2 v = {a, b};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, sg_bf16x2));
    sg_f32x2 hv = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    v -= hv;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(v, sg_bf16x2));
    sg_f32x2 mv = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    v -= mv;
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(v, sg_bf16x2));
}
typedef unsigned sg_u32x2 __attribute__((ext_vector_type(2)));

// ---- registers -> LDS planes [plane][row][k]
template <bool KC, int NQ, int NT, int PLANE>
__device__ __forceinline__ void sg_store(unsigned short* __restrict__ img, const SgRegs<NQ>& s) {
    const int t = threadIdx.x;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int slot = t + NT * i;
            const int r = slot >> 3, k = (slot & 7) * 4;
            unsigned h0, m0, l0, h1, m1, l1;
            sg_split_pair(s.q[i][0], s.q[i][1], h0, m0, l0);
            sg_split_pair(s.q[i][2], s.q[i][3], h1, m1, l1);
            unsigned short* d = img + r * SG_LDK + k;
            *reinterpret_cast<sg_u32x2*>(d) = (sg_u32x2){h0, h1};
            *reinterpret_cast<sg_u32x2*>(d + PLANE) = (sg_u32x2){m0, m1};
            *reinterpret_cast<sg_u32x2*>(d + 2 * PLANE) = (sg_u32x2){l0, l1};
        }
    } else if (NQ == 4) {
        // q[i][j] = element (row rb*4 + j, k kb*4 + i): write row j's four k values as one quad
        const int r = (t >> 3) * 4, k = (t & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h0, m0, l0, h1, m1, l1;
            sg_split_pair(s.q[0][j], s.q[1][j], h0, m0, l0);
            sg_split_pair(s.q[2][j], s.q[3][j], h1, m1, l1);
            unsigned short* d = img + (r + j) * SG_LDK + k;
            *reinterpret_cast<sg_u32x2*>(d) = (sg_u32x2){h0, h1};
            *reinterpret_cast<sg_u32x2*>(d + PLANE) = (sg_u32x2){m0, m1};
            *reinterpret_cast<sg_u32x2*>(d + 2 * PLANE) = (sg_u32x2){l0, l1};
        }
    } else if (NQ == 1) {
        // q[0][j] = element (row rb*4 + j, k): one bf16 per plane and row (the 64-row B tile of the 8-wave form)
        const int r = (t >> 5) * 4, k = t & 31;
        unsigned h0, m0, l0, h1, m1, l1;
        sg_split_pair(s.q[0][0], s.q[0][1], h0, m0, l0);
        sg_split_pair(s.q[0][2], s.q[0][3], h1, m1, l1);
        unsigned short* d = img + r * SG_LDK + k;
        d[0] = (unsigned short)h0;            d[SG_LDK] = (unsigned short)(h0 >> 16);
        d[2 * SG_LDK] = (unsigned short)h1;   d[3 * SG_LDK] = (unsigned short)(h1 >> 16);
        d += PLANE;
        d[0] = (unsigned short)m0;            d[SG_LDK] = (unsigned short)(m0 >> 16);
        d[2 * SG_LDK] = (unsigned short)m1;   d[3 * SG_LDK] = (unsigned short)(m1 >> 16);
        d += PLANE;
        d[0] = (unsigned short)l0;            d[SG_LDK] = (unsigned short)(l0 >> 16);
        d[2 * SG_LDK] = (unsigned short)l1;   d[3 * SG_LDK] = (unsigned short)(l1 >> 16);
    } else {
        // q[i][j] = element (row rb*4 + j, k kb*2 + i): row j's two k values as one 4-byte store
        const int r = (t >> 4) * 4, k = (t & 15) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h, m, l;
            sg_split_pair(s.q[0][j], s.q[NQ - 1][j], h, m, l);
            unsigned short* d = img + (r + j) * SG_LDK + k;
            *reinterpret_cast<unsigned*>(d) = h;
            *reinterpret_cast<unsigned*>(d + PLANE) = m;
            *reinterpret_cast<unsigned*>(d + 2 * PLANE) = l;
        }
    }
}

#define SG_MFMA(a, b, c) \
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sg_bf16x8, a), __builtin_bit_cast(sg_bf16x8, b), c, 0, 0, 0)

// NW waves per workgroup: 4 (2 x 2, 64 x 64 per wave) or 8 (4 x 2, 32 x 64 per wave: half the split work per thread and
// four waves per SIMD to hide its dependent convert / subtract chains).
// BN = 128, or 64 for products with a narrow output (N <= 64: the pooling products X' = S^T Z, dZ += S dX' at 60
// columns): the waves then stack along M only (NW x 1, 128 / NW rows x 64 columns each).
template <bool TA, bool TB, int NW, int BN>
__device__ __forceinline__ void sg_body(const SplitGemmArgs& a, int tile, int b, unsigned short* lds) {
    constexpr int NT = NW * 64;
    constexpr int WC = BN / 64;                 // wave columns: 2 or 1
    constexpr int WR = NW / WC;                 // wave rows
    constexpr int MI = 8 / WR;                  // 16-row MFMA tiles per wave
    constexpr int NQ = 1024 / NT;               // quads per thread and 128-row operand slab
    constexpr int NQB = NQ * BN / 128;          // ... and B slab
    constexpr int PLANE_B = BN * SG_LDK;
    const int m0 = (tile / a.tilesN) * SG_BM, n0 = (tile % a.tilesN) * BN;
    const float* A = a.A + (long)b * a.sA;
    const float* B = a.B + (long)b * a.sB;
    float* C = a.C + (long)b * a.sC;
    unsigned short* As = lds;
    unsigned short* Bs = lds + 3 * SG_PLANE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WC, wc = wave % WC;    // WR wave rows of 16 * MI rows, WC wave columns of 64
    const int l15 = lane & 15, kq = lane >> 4;

    sg_f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (sg_f32x4){0.f, 0.f, 0.f, 0.f};

    // A is k-contiguous when NOT transposed (A[m][k]); B is k-contiguous when transposed (B[n][k])
    // The next slab's loads are in flight in registers while the current one is multiplied.  Measured and dropped:
    // two slabs in flight (312 VGPRs -> one workgroup per CU: 95-105 TFLOP/s against 124-140; forced into 256 VGPRs
    // with LDS-only barriers instead of __syncthreads' vmcnt(0): no change) — memory latency is not what this kernel
    // waits for.  Phase ablations at A' = Tt^T S (276 us): without the MFMAs 205 us, without split + LDS stores 146 us,
    // without the global loads 217 us: the phases of a workgroup hardly overlap, and the split + store phase (~250
    // vector instructions per thread and slab in dependent convert / shift / subtract chains, two waves per SIMD to
    // hide them) costs as much as everything else together.  Eight waves per workgroup (half the elements per thread,
    // four waves per SIMD) is the next form to try.
    SgRegs<NQ> ra0;
    SgRegs<NQB> rb0;
    const int nk = (a.K + SG_KT - 1) / SG_KT;
    const bool rimA = m0 + SG_BM > a.M, rimB = n0 + BN > a.N;      // wave-uniform
    auto fetch = [&](int kt, SgRegs<NQ>& ra, SgRegs<NQB>& rb) {
        const int k0 = kt * SG_KT;
        const bool tail = k0 + SG_KT > a.K;
        if (rimA || tail) sg_load<!TA, true, NQ, NT>(A, a.lda, m0, k0, a.M, a.K, ra);
        else sg_load<!TA, false, NQ, NT>(A, a.lda, m0, k0, a.M, a.K, ra);
        if (rimB || tail) sg_load<TB, true, NQB, NT>(B, a.ldb, n0, k0, a.N, a.K, rb);
        else sg_load<TB, false, NQB, NT>(B, a.ldb, n0, k0, a.N, a.K, rb);
    };
    auto fix = [&](int kt, SgRegs<NQ>& ra, SgRegs<NQB>& rb) {        // after the wait the LDS write needs anyway
        const int k0 = kt * SG_KT;
        const bool tail = k0 + SG_KT > a.K;
        if (rimA || tail) sg_fix<!TA, NQ, NT>(m0, k0, a.M, a.K, ra);
        if (rimB || tail) sg_fix<TB, NQB, NT>(n0, k0, a.N, a.K, rb);
    };
    auto multiply = [&]() {
        // ---- 64 x 64 x 32 per wave.  The six plane products of a column tile run product by product over the four
        // row tiles: consecutive MFMAs write different accumulators (a chain of six on one accumulator stalls on
        // each result); smallest terms first.
        sg_s16x8 af[3][MI];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[p][i] = *reinterpret_cast<const sg_s16x8*>(As + p * SG_PLANE + (wr * 16 * MI + i * 16 + l15) * SG_LDK + kq * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sg_s16x8 bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[p] = *reinterpret_cast<const sg_s16x8*>(Bs + p * PLANE_B + (wc * 64 + j * 16 + l15) * SG_LDK + kq * 8);
#pragma unroll
            for (int i = 0; i < MI; ++i) SG_MFMA(af[1][i], bf[1], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MI; ++i) SG_MFMA(af[0][i], bf[2], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MI; ++i) SG_MFMA(af[2][i], bf[0], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MI; ++i) SG_MFMA(af[0][i], bf[1], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MI; ++i) SG_MFMA(af[1][i], bf[0], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MI; ++i) SG_MFMA(af[0][i], bf[0], acc[i][j]);
        }
    };
    fetch(0, ra0, rb0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();                       // the previous slab's readers are done
        fix(kt, ra0, rb0);
        sg_store<!TA, NQ, NT, SG_PLANE>(As, ra0);
        sg_store<TB, NQB, NT, PLANE_B>(Bs, rb0);
        __syncthreads();
        if (kt + 1 < nk) fetch(kt + 1, ra0, rb0);
        multiply();
    }
    // ---- C tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wc * 64 + j * 16 + l15;
            if (col >= a.N) continue;
            if (a.bias) {
                const float bv = a.bias[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += bv;
            }
            if (a.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
            }
            if (a.split_out) {
                // this lane holds 4 consecutive rows of one column: half of a k8 group -> one 8-byte store per plane
                // (rows M .. 8 * split_k8 - 1 are the operand's zero padding)
                const int row0 = m0 + wr * 16 * MI + i * 16 + kq * 4;
                if (row0 < a.split_k8 * 8) {
                    const int vc = a.split_c0 + col;
                    unsigned short* vb = a.split_out + (long)b * 3 * a.split_ct * a.split_k8 * 128;
                    sg_u16x4 h, m, l;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned short hh, mm, ll;
                        sg_split(row0 + r < a.M ? acc[i][j][r] : 0.f, hh, mm, ll);
                        h[r] = hh; m[r] = mm; l[r] = ll;
                    }
                    const long o = vs_index(0, a.split_ct, a.split_k8, vc >> 4, row0 >> 3, vc & 15, row0 & 7);
                    const long pl = (long)a.split_ct * a.split_k8 * 128;
                    *reinterpret_cast<sg_u16x4*>(vb + o) = h;
                    *reinterpret_cast<sg_u16x4*>(vb + o + pl) = m;
                    *reinterpret_cast<sg_u16x4*>(vb + o + 2 * pl) = l;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 16 * MI + i * 16 + kq * 4 + r;
                if (row >= a.M) continue;
                float* cp = C + (long)row * a.ldc + col;
                *cp = a.beta != 0.f ? acc[i][j][r] + a.beta * *cp : acc[i][j][r];
            }
        }
}

template <int NW, int BN = SG_BN>
__global__ __launch_bounds__(NW * 64) void k_gemm_split_bf16(SplitGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sg_lds[];
    // XCD-aware work mapping (as k_aggregate): workgroups are dealt to the 8 XCDs round-robin in dispatch order, each XCD
    // with its own L2.  The tiles of ONE batch element read the same operand panels (every A panel tilesN times, every B
    // panel tilesM times): give each XCD a contiguous run of (batch, tile) items so those re-reads hit its L2 instead of
    // going to HBM a second time from another XCD.
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3, qd = nwg >> 3, rm = nwg & 7;
    const int w = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    const int tile = w % (int)gridDim.x, b = w / (int)gridDim.x;
    if (a.tA) {
        if (a.tB) sg_body<true, true, NW, BN>(a, tile, b, sg_lds);
        else sg_body<true, false, NW, BN>(a, tile, b, sg_lds);
    } else {
        if (a.tB) sg_body<false, true, NW, BN>(a, tile, b, sg_lds);
        else sg_body<false, false, NW, BN>(a, tile, b, sg_lds);
    }
}

}  // namespace

// Shapes the split kernel takes AND that are worth a 128 x 128 tile: both output extents at least most of a tile,
// a contraction long enough to amortise the prologue, and enough tiles to fill the chip.
bool gemm_split_usable(const GemmDesc& d, int batch, int ksplit) {
    if (knobs().no_split_gemm) return false;
    if (d.atomic || d.fix_part || ksplit > 1) return false;      // (sK only matters under split-K)
    if ((d.bias || d.act) && d.beta != 0.f) return false;
    if (d.split_out && d.beta != 0.f) return false;
    if (d.alpha != 1.f || !(d.beta == 0.f || d.beta == 1.f)) return false;
    // N in [48, 64]: the 128 x 64 tile (8-wave form only)
    const bool narrow = d.N >= 48 && d.N <= 64 && !knobs().split_gemm_w4 && !d.split_out;
    if (d.M < 96 || (d.N < 80 && !narrow) || d.K < 40) return false;       // (the rim loads also need every extent >= 4)
    const int bn = narrow ? 64 : SG_BN;
    const long tiles = (long)((d.M + SG_BM - 1) / SG_BM) * ((d.N + bn - 1) / bn) * batch;
    return tiles >= 256;
}

void gemm_split_bf16(Seq& q, const GemmDesc& d, int batch) {
    if (!q.ok() || batch <= 0 || d.M <= 0 || d.N <= 0) return;
    if (d.M < 4 || d.N < 4 || d.K < 4) {          // no 16-byte quad fits an extent: the fp32-MFMA kernel takes it
        bgemm_group(q, &d, 1, batch, 1);
        return;
    }
    const bool w8 = !knobs().split_gemm_w4;
    if (w8 && d.N <= 64) {
        static DynLdsOnce attr64;
        ensure_dyn_lds(q, attr64, reinterpret_cast<const void*>(&k_gemm_split_bf16<8, 64>), sg_lds_bytes(64),
                       "k_gemm_split_bf16<8,64>");
        if (!q.ok()) return;
        SplitGemmArgs a{d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.sA, d.sB, d.sC, d.tA ? 1 : 0, d.tB ? 1 : 0,
                        d.beta, 1, d.split_out, d.split_ct, d.split_k8, d.split_c0, d.bias, d.act};
        const int tiles = (d.M + SG_BM - 1) / SG_BM;
        hipLaunchKernelGGL((k_gemm_split_bf16<8, 64>), dim3(tiles, batch), dim3(512), sg_lds_bytes(64), q.stream, a);
        q.check_launch("gemm_split_bf16");
        return;
    }
    static DynLdsOnce attr4, attr8;
    if (w8) ensure_dyn_lds(q, attr8, reinterpret_cast<const void*>(&k_gemm_split_bf16<8>), SG_LDS_BYTES, "k_gemm_split_bf16<8>");
    else ensure_dyn_lds(q, attr4, reinterpret_cast<const void*>(&k_gemm_split_bf16<4>), SG_LDS_BYTES, "k_gemm_split_bf16<4>");
    if (!q.ok()) return;
    SplitGemmArgs a{d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.sA, d.sB, d.sC, d.tA ? 1 : 0, d.tB ? 1 : 0, d.beta,
                    (d.N + SG_BN - 1) / SG_BN, d.split_out, d.split_ct, d.split_k8, d.split_c0, d.bias, d.act};
    const int tiles = ((d.M + SG_BM - 1) / SG_BM) * a.tilesN;
    if (w8) hipLaunchKernelGGL(k_gemm_split_bf16<8>, dim3(tiles, batch), dim3(512), SG_LDS_BYTES, q.stream, a);
    else hipLaunchKernelGGL(k_gemm_split_bf16<4>, dim3(tiles, batch), dim3(256), SG_LDS_BYTES, q.stream, a);
    q.check_launch("gemm_split_bf16");
}

}  // namespace dp
