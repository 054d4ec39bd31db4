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
constexpr int SG_PLANE = SG_BM * SG_LDK;            // elements of one plane of one operand (BM == BN)
constexpr int SG_LDS_BYTES = 2 * 3 * SG_PLANE * 2;  // A and B, three planes, 2 bytes

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
};

// 16 fp32 values of one operand slab share: q[i] = four consecutive elements along the operand's contiguous dimension
struct SgRegs {
    sg_f32x4 q[4];
};

// ---- global -> registers.  KC: k is the contiguous dimension (element (r, k) at p[r * ld + k]); else p[k * ld + r].
// Loads are unconditional on clamped addresses; quads that stick out of the operand are re-read element-wise (tile
// edges only).
template <bool KC>
__device__ __forceinline__ void sg_load(const float* __restrict__ p, int ld, int r0, int k0, int rmax, int kmax,
                                        SgRegs& s) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r, k;                                   // first element of quad i
        if (KC) {
            const int slot = t + 256 * i;           // (row, k-quad): 8 quads per row
            r = r0 + (slot >> 3);
            k = k0 + (slot & 7) * 4;
        } else {
            r = r0 + (t >> 3) * 4;                  // one 4 k x 4 row block per thread: k-block t & 7, row block t >> 3
            k = k0 + (t & 7) * 4 + i;
        }
        const int c = KC ? k : r, o = KC ? r : k;   // contiguous / other coordinate
        const int cmax = KC ? kmax : rmax, omax = KC ? rmax : kmax;
        const float* row = p + (long)min(o, omax - 1) * ld;
        if (c + 4 <= cmax) {
            s.q[i] = *reinterpret_cast<const sg_f32x4_u*>(row + c);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s.q[i][j] = c + j < cmax ? row[c + j] : 0.f;
        }
        if (o >= omax) s.q[i] = (sg_f32x4){0.f, 0.f, 0.f, 0.f};     // past K (or past the rows: never stored)
    }
}

__device__ __forceinline__ void sg_split(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    bf16_split3(v, h, m, l);
}

// ---- registers -> LDS planes [plane][row][k]
template <bool KC>
__device__ __forceinline__ void sg_store(unsigned short* __restrict__ img, const SgRegs& s) {
    const int t = threadIdx.x;
    if (KC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = t + 256 * i;
            const int r = slot >> 3, k = (slot & 7) * 4;
            sg_u16x4 h, m, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned short hh, mm, ll;
                sg_split(s.q[i][j], hh, mm, ll);
                h[j] = hh; m[j] = mm; l[j] = ll;
            }
            unsigned short* d = img + r * SG_LDK + k;
            *reinterpret_cast<sg_u16x4*>(d) = h;
            *reinterpret_cast<sg_u16x4*>(d + SG_PLANE) = m;
            *reinterpret_cast<sg_u16x4*>(d + 2 * SG_PLANE) = l;
        }
    } else {
        // q[i][j] = element (row rb*4 + j, k kb*4 + i): write row j's four k values as one quad
        const int r = (t >> 3) * 4, k = (t & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sg_u16x4 h, m, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned short hh, mm, ll;
                sg_split(s.q[i][j], hh, mm, ll);
                h[i] = hh; m[i] = mm; l[i] = ll;
            }
            unsigned short* d = img + (r + j) * SG_LDK + k;
            *reinterpret_cast<sg_u16x4*>(d) = h;
            *reinterpret_cast<sg_u16x4*>(d + SG_PLANE) = m;
            *reinterpret_cast<sg_u16x4*>(d + 2 * SG_PLANE) = l;
        }
    }
}

#define SG_MFMA(a, b, c) \
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sg_bf16x8, a), __builtin_bit_cast(sg_bf16x8, b), c, 0, 0, 0)

template <bool TA, bool TB>
__device__ __forceinline__ void sg_body(const SplitGemmArgs& a, int tile, unsigned short* lds) {
    const int b = blockIdx.y;
    const int m0 = (tile / a.tilesN) * SG_BM, n0 = (tile % a.tilesN) * SG_BN;
    const float* A = a.A + (long)b * a.sA;
    const float* B = a.B + (long)b * a.sB;
    float* C = a.C + (long)b * a.sC;
    unsigned short* As = lds;
    unsigned short* Bs = lds + 3 * SG_PLANE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, kq = lane >> 4;

    sg_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (sg_f32x4){0.f, 0.f, 0.f, 0.f};

    // A is k-contiguous when NOT transposed (A[m][k]); B is k-contiguous when transposed (B[n][k])
    SgRegs ra, rb;
    sg_load<!TA>(A, a.lda, m0, 0, a.M, a.K, ra);
    sg_load<TB>(B, a.ldb, n0, 0, a.N, a.K, rb);
    const int nk = (a.K + SG_KT - 1) / SG_KT;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();                       // the previous slab's readers are done
        sg_store<!TA>(As, ra);
        sg_store<TB>(Bs, rb);
        __syncthreads();
        if (kt + 1 < nk) {
            sg_load<!TA>(A, a.lda, m0, (kt + 1) * SG_KT, a.M, a.K, ra);
            sg_load<TB>(B, a.ldb, n0, (kt + 1) * SG_KT, a.N, a.K, rb);
        }
        // ---- 64 x 64 x 32 per wave
        sg_s16x8 af[3][4];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[p][i] = *reinterpret_cast<const sg_s16x8*>(As + p * SG_PLANE + (wr * 64 + i * 16 + l15) * SG_LDK + kq * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sg_s16x8 bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[p] = *reinterpret_cast<const sg_s16x8*>(Bs + p * SG_PLANE + (wc * 64 + j * 16 + l15) * SG_LDK + kq * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // smallest terms first
                SG_MFMA(af[1][i], bf[1], acc[i][j]);
                SG_MFMA(af[0][i], bf[2], acc[i][j]);
                SG_MFMA(af[2][i], bf[0], acc[i][j]);
                SG_MFMA(af[0][i], bf[1], acc[i][j]);
                SG_MFMA(af[1][i], bf[0], acc[i][j]);
                SG_MFMA(af[0][i], bf[0], acc[i][j]);
            }
        }
    }
    // ---- C tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wc * 64 + j * 16 + l15;
            if (col >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 64 + i * 16 + kq * 4 + r;
                if (row >= a.M) continue;
                float* cp = C + (long)row * a.ldc + col;
                *cp = a.beta != 0.f ? acc[i][j][r] + a.beta * *cp : acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256) void k_gemm_split_bf16(SplitGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sg_lds[];
    const int tile = blockIdx.x;
    if (a.tA) {
        if (a.tB) sg_body<true, true>(a, tile, sg_lds);
        else sg_body<true, false>(a, tile, sg_lds);
    } else {
        if (a.tB) sg_body<false, true>(a, tile, sg_lds);
        else sg_body<false, false>(a, tile, sg_lds);
    }
}

}  // namespace

// Shapes the split kernel takes AND that are worth a 128 x 128 tile: both output extents at least most of a tile,
// a contraction long enough to amortise the prologue, and enough tiles to fill the chip.
bool gemm_split_usable(const GemmDesc& d, int batch, int ksplit) {
    if (knobs().no_split_gemm) return false;
    if (d.bias || d.act || d.atomic || d.split_out || d.fix_part || d.sK != 0 || ksplit > 1) return false;
    if (d.alpha != 1.f || !(d.beta == 0.f || d.beta == 1.f)) return false;
    if (d.M < 96 || d.N < 96 || d.K < 64) return false;
    const long tiles = (long)((d.M + SG_BM - 1) / SG_BM) * ((d.N + SG_BN - 1) / SG_BN) * batch;
    return tiles >= 256;
}

void gemm_split_bf16(Seq& q, const GemmDesc& d, int batch) {
    if (!q.ok() || batch <= 0 || d.M <= 0 || d.N <= 0) return;
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_gemm_split_bf16), SG_LDS_BYTES, "k_gemm_split_bf16");
    if (!q.ok()) return;
    SplitGemmArgs a{d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.sA, d.sB, d.sC, d.tA ? 1 : 0, d.tB ? 1 : 0, d.beta,
                    (d.N + SG_BN - 1) / SG_BN};
    const int tiles = ((d.M + SG_BM - 1) / SG_BM) * a.tilesN;
    hipLaunchKernelGGL(k_gemm_split_bf16, dim3(tiles, batch), dim3(256), SG_LDS_BYTES, q.stream, a);
    q.check_launch("gemm_split_bf16");
}

}  // namespace dp
