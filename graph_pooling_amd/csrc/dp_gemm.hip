// Batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, k-ordered fma
// chain — bit-for-bit an fmaf chain, so parity with the CPU oracle is reduction-order only).
//
//   C[b] = act(alpha * op(A[b]) * op(B[b]) + beta * C[b] + bias)
//
// This is the general contraction of the DiffPool path: X·W transforms, A·P aggregation
// (encoders.py:965,968), the pooling products S^T Z, S^T A, (S^T A) S (encoders.py:1278-1279) and
// every backward contraction.  Arbitrary M, N, K and leading dimensions (the path's sizes are
// 500, 89, 50, 60 ... — nothing is a multiple of anything), zero-filled edges.
//
// Tiling: 256 threads = 4 waves arranged WAVES_M x WAVES_N over a BM x BN workgroup tile; each wave
// owns (BM/WAVES_M) x (BN/WAVES_N) as MFMA 16x16 tiles.  The DiffPool batches are small (B = 20 graphs
// of <= 500 nodes), so the launcher picks the LARGEST tile that still yields >= ~2 workgroups per CU:
// the kernels are latency-bound, and bytes in flight (Little's law) is what buys bandwidth.
// K is walked in 32-wide slabs staged through LDS with a two-deep register prefetch (the slab two
// steps ahead is in flight while the current one is multiplied).  16-byte global loads when the
// operand's alignment allows.  The LDS image of an operand keeps the operand's own memory
// orientation so global reads and LDS writes are both lane-contiguous:
//   k-contiguous operand  -> image [r][k], row stride 34  (frag read bank = 2r + k : conflict-free)
//   r-contiguous operand  -> image [k][r], row stride = 16 mod 32 (k rows land on disjoint bank halves)
#include <cstdlib>

#include "dp_common.h"

namespace dp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    long sA, sB, sC;
    float alpha, beta;
    int act;
    int tilesN;
    int tA, tB;
    int ksplit;       // K is cut into `ksplit` ranges handled by different workgroups
    int ntiles;       // output tiles of this problem; its workgroups are (range, tile) pairs: local = ks*ntiles + tile
    long sK;          // != 0: range ks stores its partial at C + ks*sK (slab rows, summed by the caller's reduce)
    int atomic;       // 1: ranges add into C with float atomics (C pre-zeroed / accumulated into; no bias/act)
    unsigned short* split_out;            // optional 3-plane bf16 copy of C (see GemmDesc)
    int split_ct, split_k8, split_c0;
    float* fix_part;                      // deterministic split-K (see GemmDesc)
    int* fix_cnt;
    const float* rp_xhat;                 // row partials of the final C (see GemmDesc)
    int rp_ldx;
    float* rp_part;
    int rp_G, rp_g;
    int stamp_slot;   // diagnostic build only
};

constexpr int KT = 32;

#ifdef DP_STAMP
// Diagnostic build only (csrc/build.sh with DP_STAMP=1, tools/gemm_stamps.py): per launch slot, the earliest start and
// the latest end over all workgroups plus the phases of workgroup (0,0), on the 100 MHz realtime counter (the shader
// clock differs between XCDs).  No stamp executes in the product build.
__device__ unsigned long long g_gemm_stamps[64][8];
// (one atomic per workgroup on a shared word distorts everything: only the first and the last workgroup stamp)
#define GEMM_STAMP_MIN(slot)                                                                   \
    do {                                                                                       \
        if (blockIdx.x == 0 && blockIdx.y == 0) g_gemm_stamps[slot][0] = wall_clock64();       \
    } while (0)
#define GEMM_STAMP_MAX(slot)                                                                   \
    do {                                                                                       \
        if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) g_gemm_stamps[slot][1] = wall_clock64(); \
    } while (0)
#define GEMM_STAMP(slot, i)                                                                               \
    do {                                                                                                  \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_gemm_stamps[slot][i] = wall_clock64(); \
    } while (0)
#else
#define GEMM_STAMP_MIN(slot) \
    do {                     \
    } while (0)
#define GEMM_STAMP_MAX(slot) \
    do {                     \
    } while (0)
#define GEMM_STAMP(slot, i) \
    do {                    \
    } while (0)
#endif

template <int R, bool KCONTIG>
struct LdsImage {
    // r-contiguous rows must start 16 banks apart (stride = 16 mod 32) and stay 16-byte aligned
    static constexpr int STRIDE = KCONTIG ? (KT + 2) : ((R % 32 == 16) ? R + 32 : R + 16);
    static constexpr int SIZE = KCONTIG ? R * STRIDE : KT * STRIDE;
    __device__ static inline int addr(int r, int k) { return KCONTIG ? r * STRIDE + k : k * STRIDE + r; }
};

// One R x KT operand slab held in registers as float4 "slots" (R*KT/4 slots over 256 threads).
template <int R>
struct Slab {
    static constexpr int SLOTS = (R * KT / 4 + 255) / 256;
    f32x4 v[SLOTS];
};

// `KCONTIG`: element (r,k) at p[r*ld + k], else at p[k*ld + r].  A slot is four elements along the operand's
// contiguous dimension c (extent cmax); o is the other dimension (extent omax).
//
// slab_load only ISSUES loads: one 16-byte load per slot from a clamped, always-valid address, no predicate and no
// branch around it (a branch around a load makes hipcc wait for that load before the join; the version with
// per-element predicated loads for rows that are not 16-byte aligned -- F = 89, K = 50, 90 in the DD model -- had 168
// `s_waitcnt vmcnt(0)` in one kernel).  The rows need not be 16-byte aligned: global dwordx4 loads only need dword
// alignment on gfx9 under HSA.  A slot that sticks out past cmax is read shifted back so that it ENDS at cmax;
// slab_store undoes the shift and zeroes what lies outside, after the wait the LDS write needs anyway.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

template <int R, bool KCONTIG>
__device__ __forceinline__ void slab_coords(int slot, int r0, int k0, int& c, int& o) {
    int r, k;
    if (KCONTIG) {
        r = slot / (KT / 4);
        k = (slot % (KT / 4)) * 4;
    } else {
        k = slot / (R / 4);
        r = (slot % (R / 4)) * 4;
    }
    c = KCONTIG ? k0 + k : r0 + r;
    o = KCONTIG ? r0 + r : k0 + k;
}

// QUAD = false serves operands with an extent below 4 in the contiguous dimension (no 16-byte load fits).
template <int R, bool KCONTIG, bool QUAD>
__device__ inline void slab_load(const float* __restrict__ p, int ld, int r0, int k0, int rmax, int kmax, Slab<R>& s) {
    const int t = threadIdx.x;
    const int cmax = KCONTIG ? kmax : rmax, omax = KCONTIG ? rmax : kmax;
#pragma unroll
    for (int i = 0; i < Slab<R>::SLOTS; ++i) {
        const int slot = min(t + i * 256, R * KT / 4 - 1);
        int c, o;
        slab_coords<R, KCONTIG>(slot, r0, k0, c, o);
        const float* q = p + (long)min(o, omax - 1) * ld;
        if (QUAD) {
            s.v[i] = *reinterpret_cast<const f32x4_u*>(q + min(c, cmax - 4));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s.v[i][j] = q[min(c + j, cmax - 1)];
        }
    }
}

template <int R, bool KCONTIG, bool QUAD>
__device__ inline void slab_store(float* lds, const Slab<R>& s, int r0, int k0, int rmax, int kmax) {
    using L = LdsImage<R, KCONTIG>;
    const int t = threadIdx.x;
    const int cmax = KCONTIG ? kmax : rmax, omax = KCONTIG ? rmax : kmax;
#pragma unroll
    for (int i = 0; i < Slab<R>::SLOTS; ++i) {
        const int slot = t + i * 256;
        if (slot < R * KT / 4) {
            int c, o;
            slab_coords<R, KCONTIG>(slot, r0, k0, c, o);
            f32x4 v = s.v[i];
            const bool ov = o < omax;
            if (QUAD) {
                // the load was shifted back by sh elements (0 for a whole slot, >= 4: all of it lies outside); no
                // branch on the shape here, so the wait for the slab stays a counted one
                const int sh = c - min(c, cmax - 4);
                f32x4 w;
                w[0] = sh == 0 ? v[0] : sh == 1 ? v[1] : sh == 2 ? v[2] : sh == 3 ? v[3] : 0.f;
                w[1] = sh == 0 ? v[1] : sh == 1 ? v[2] : sh == 2 ? v[3] : 0.f;
                w[2] = sh == 0 ? v[2] : sh == 1 ? v[3] : 0.f;
                w[3] = sh == 0 ? v[3] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ov ? w[j] : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (ov && c + j < cmax) ? v[j] : 0.f;
            }
            if (KCONTIG) {
                const int r = slot / (KT / 4), k = (slot % (KT / 4)) * 4;
                float* d = lds + L::addr(r, k);   // 8-byte aligned (row stride 34 floats)
                d[0] = v[0];
                d[1] = v[1];
                d[2] = v[2];
                d[3] = v[3];
            } else {
                const int k = slot / (R / 4), r = (slot % (R / 4)) * 4;
                *reinterpret_cast<f32x4*>(lds + L::addr(r, k)) = v;   // row stride is a multiple of 16 B
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, bool QUAD>
__device__ inline void gemm_body(const GemmArgs& a, int tile, int ks, float* lds) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 16, NI = WN / 16;
    static_assert(MI >= 1 && NI >= 1, "wave tile must be at least 16x16");
    using LA = LdsImage<BM, !TA>;   // A operand: not transposed -> stored [M][K] (k-contiguous)
    using LB = LdsImage<BN, TB>;    // B operand: not transposed -> stored [K][N] (r-contiguous)
    float* As = lds;
    float* Bs = lds + ((LA::SIZE + 3) & ~3);

    const int tm = tile / a.tilesN, tn = tile % a.tilesN;
    const int b = blockIdx.y;
    const float* A = a.A + (long)b * a.sA;
    const float* B = a.B + (long)b * a.sB;
    float* C = a.C + (long)b * a.sC + (long)ks * a.sK;
    const int m0 = tm * BM, n0 = tn * BN;
    // this workgroup's K range (whole slabs of KT)
    const int kchunk = ((a.K + a.ksplit * KT - 1) / (a.ksplit * KT)) * KT;
    const int kbeg = ks * kchunk;
    const int kend = min(a.K, kbeg + kchunk);
    if (kbeg >= kend && ks > 0 && a.sK == 0 && !a.fix_part) return;   // nothing to add to a shared C

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int l15 = lane & 15, l4 = lane >> 4;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Epilogue operands do not depend on the product: bias and (for beta != 0) the old C tile are asked for here,
    // clamped and unpredicated, and are long back when the K loop ends.
    const bool rmw = !a.atomic && a.beta != 0.f && (ks == 0 || a.sK != 0 || a.fix_part);
    float bv[NI], cold[MI][NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = min(n0 + wc * WN + j * 16 + l15, a.N - 1);
        bv[j] = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cold[i][j][r] = 0.f;
                if (rmw) cold[i][j][r] = C[(long)min(m0 + wr * WM + i * 16 + l4 * 4 + r, a.M - 1) * a.ldc + col];
            }
    }

    // row-partial hook: the xhat tile is asked for with the other epilogue operands (wave column 0 holds every column)
    float xh[MI][NI][4];
    if (a.rp_part) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = min(n0 + wc * WN + j * 16 + l15, a.N - 1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xh[i][j][r] = a.rp_xhat[((long)b * a.M + min(m0 + wr * WM + i * 16 + l4 * 4 + r, a.M - 1)) * a.rp_ldx + col];
        }
    }

    const int nk = kend > kbeg ? (kend - kbeg + KT - 1) / KT : 0;
    // Two K slabs of each operand are in flight ahead of the multiply.  (Four -- every slab of the 1-4 slab ranges the
    // model hands over -- measured no better: DD 0.376 vs 0.371 ms per step, and 184 VGPRs on the 128x32 tile.)
    constexpr int DEPTH = 2;
    Slab<BM> ra[DEPTH];
    Slab<BN> rb[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) {
            slab_load<BM, !TA, QUAD>(A, a.lda, m0, kbeg + d * KT, a.M, kend, ra[d]);
            slab_load<BN, TB, QUAD>(B, a.ldb, n0, kbeg + d * KT, a.N, kend, rb[d]);
        }

    GEMM_STAMP(a.stamp_slot, 3);     // operand and epilogue loads issued
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            float af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = As[LA::addr(wr * WM + i * 16 + l15, kk + l4)];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = Bs[LB::addr(wc * WN + j * 16 + l15, kk + l4)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    for (int kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (kt + d < nk) {
                if (kt + d > 0) __syncthreads();          // the previous slab's readers are done
                slab_store<BM, !TA, QUAD>(As, ra[d], m0, kbeg + (kt + d) * KT, a.M, kend);
                slab_store<BN, TB, QUAD>(Bs, rb[d], n0, kbeg + (kt + d) * KT, a.N, kend);
                __syncthreads();
                if (kt + d == 0) GEMM_STAMP(a.stamp_slot, 4);   // first slabs arrived and published
                if (kt + d + DEPTH < nk) {
                    slab_load<BM, !TA, QUAD>(A, a.lda, m0, kbeg + (kt + d + DEPTH) * KT, a.M, kend, ra[d]);
                    slab_load<BN, TB, QUAD>(B, a.ldb, n0, kbeg + (kt + d + DEPTH) * KT, a.N, kend, rb[d]);
                }
                compute();
            }
        }
    }

    GEMM_STAMP(a.stamp_slot, 5);     // K loop done
    if (a.fix_part && a.ksplit > 1) {
        // ---- deterministic split-K: partial out, ticket, the last range sums all partials in range order
        const long pstride = (long)a.M * a.N;
        float* Pb = a.fix_part + (long)b * a.ksplit * pstride;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = n0 + wc * WN + j * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wr * WM + i * 16 + l4 * 4 + r;
                    if (row < a.M && col < a.N)      // agent-scope (write-through) store: no cache flush needed later
                        __hip_atomic_store(Pb + (long)ks * pstride + (long)row * a.N + col, acc[i][j][r],
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        // the partial is at the device's coherence point once its stores have been acknowledged; a full fence
        // (`__threadfence()`: L2 write-back + invalidate per workgroup) made this launch 70 us slower.
        // MEMORY-MODEL NOTE: there is no release/acquire pair in this source.  The ordering rests on gfx950 ISA
        // behaviour, not on the HIP memory model: (1) agent-scope relaxed atomic stores are write-through (sc1) and
        // `s_waitcnt vmcnt(0)` returns only when every one of them has been acknowledged at the device's coherence
        // point; (2) the workgroup barrier orders all waves' stores before thread 0's ticket; (3) the ticket is an
        // agent-scope atomic RMW performed at that same coherence point, so whoever draws the last ticket does so
        // after every partial is visible there; (4) the reader's agent-scope relaxed atomic loads are served past
        // the XCD-local L2.  The portable form (ticket fetch_add with __ATOMIC_ACQ_REL at agent scope) adds a
        // buffer_wbl2 / buffer_inv pair per workgroup — the 70 us above.  The guard for this assumption is
        // tests/test_gpu_model.py::test_forward_is_bit_reproducible_with_split_k_pooling (40 runs, bit-identical)
        // and tools/race_hunt.py (2 500 runs).  The whole-level kernels' grid barrier (dp_small.hip) uses the same
        // store / wait / ticket / load pattern.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // (also: every wave is done reading the operand slabs)
        int* ticket = reinterpret_cast<int*>(lds);
        if (threadIdx.x == 0)
            *ticket = atomicAdd(a.fix_cnt + (long)b * a.ntiles + tile, 1);
        __syncthreads();
        if (*ticket != a.ksplit - 1) return;               // uniform: somebody else arrives last
        // range by range (fixed order), every element of the tile in flight per range: ksplit round trips
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k2 = 0; k2 < a.ksplit; ++k2) {
            f32x4 t[MI][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int col = min(n0 + wc * WN + j * 16 + l15, a.N - 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = min(m0 + wr * WM + i * 16 + l4 * 4 + r, a.M - 1);
                        t[i][j][r] = __hip_atomic_load(Pb + (long)k2 * pstride + (long)row * a.N + col, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);      // past this XCD's caches
                    }
                }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] += t[i][j];
        }
    }
    float rp0[MI][4], rp1[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) rp0[i][r] = rp1[i][r] = 0.f;
    // C/D map of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wc * WN + j * 16 + l15;
            if (col >= a.N) continue;
            if (a.split_out) {
                // this lane holds 4 consecutive rows of one column: half of a k8 group -> one 8-byte store per plane
                const int row0 = m0 + wr * WM + i * 16 + l4 * 4;
                if (row0 < a.split_k8 * 8) {
                    const int vc = a.split_c0 + col;
                    unsigned short* vb = a.split_out + (long)b * 3 * a.split_ct * a.split_k8 * 128;
                    unsigned short h[4], m[4], l[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = (row0 + r < a.M) ? a.alpha * acc[i][j][r] + bv[j] : 0.f;
                        bf16_split3(v, h[r], m[r], l[r]);
                    }
                    const long o = vs_index(0, a.split_ct, a.split_k8, vc >> 4, row0 >> 3, vc & 15, row0 & 7);
                    const long pl = (long)a.split_ct * a.split_k8 * 128;
                    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<u16x4*>(vb + o) = (u16x4){h[0], h[1], h[2], h[3]};
                    *reinterpret_cast<u16x4*>(vb + o + pl) = (u16x4){m[0], m[1], m[2], m[3]};
                    *reinterpret_cast<u16x4*>(vb + o + 2 * pl) = (u16x4){l[0], l[1], l[2], l[3]};
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * WM + i * 16 + l4 * 4 + r;
                if (row >= a.M) continue;
                float* cp = C + (long)row * a.ldc + col;
                if (a.atomic) {
                    atomicAdd(cp, a.alpha * acc[i][j][r]);
                    continue;
                }
                float v = a.alpha * acc[i][j][r] + bv[j];
                if (rmw) v += a.beta * cold[i][j][r];
                if (a.act == 1) v = fmaxf(v, 0.f);
                *cp = v;
                if (a.rp_part) {
                    rp0[i][r] += v;
                    rp1[i][r] += v * xh[i][j][r];
                }
            }
        }
    if (a.rp_part) {
        // a row's columns all sit in wave column 0 (N <= WN, checked on the host): 16 lanes per row, no LDS
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s0 = row16_sum(rp0[i][r]), s1 = row16_sum(rp1[i][r]);
                const int row = m0 + wr * WM + i * 16 + l4 * 4 + r;
                if (wc == 0 && tn == 0 && l15 == 0 && row < a.M) {
                    float* pp = a.rp_part + (((long)b * a.M + row) * a.rp_G + a.rp_g) * 2;
                    pp[0] = s0;
                    pp[1] = s1;
                }
            }
    }
    GEMM_STAMP(a.stamp_slot, 6);
}

// One launch, up to GEMM_GROUP_MAX independent problems (same batch count): the workgroup finds its
// problem from the tile prefix, then runs the body for that problem's transposes.
struct GemmGroupArgs {
    uint4* zero_p;       // non-null: workgroup (0, 0) clears zero_n16 16-byte words (Seq::fold_zero_p)
    int zero_n16;
    const int* pred;     // non-null: run only when *pred != 0
    int count;
    int stamp_slot;      // diagnostic build only
    int tile0[GEMM_GROUP_MAX + 1];
    GemmArgs p[GEMM_GROUP_MAX];
};

template <int R, bool KC>
constexpr int lds_size() { return LdsImage<R, KC>::SIZE; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void bgemm_kernel(GemmGroupArgs g) {
    __shared__ __attribute__((aligned(16))) float
        lds[cmax(lds_size<BM, true>(), lds_size<BM, false>()) + cmax(lds_size<BN, true>(), lds_size<BN, false>()) + 8];
    if (g.zero_p && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < g.zero_n16; i += 256) g.zero_p[i] = make_uint4(0, 0, 0, 0);
    if (g.pred && __builtin_amdgcn_readfirstlane(*g.pred) == 0) return;
    if (threadIdx.x == 0) GEMM_STAMP_MIN(g.stamp_slot);
    GEMM_STAMP(g.stamp_slot, 2);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_GROUP_MAX; ++i)
        if (i < g.count && (int)blockIdx.x >= g.tile0[i]) pi = i;
    GemmArgs a = g.p[pi];            // a copy: one burst of scalar loads instead of a kernarg read per use
    a.stamp_slot = g.stamp_slot;
    // the launch holds exactly the (range, tile) pairs that have work: no workgroup starts only to find out it is idle
    // (a launch-wide range count made 2/3 of some grouped launches idle workgroups, ~5 ns each: 30 us per DD step)
    const int local = blockIdx.x - g.tile0[pi];
    const int ks = local / a.ntiles, tile = local - ks * a.ntiles;
    // 16-byte operand loads need four elements along each operand's contiguous dimension
    const bool quad = (a.tA ? a.M : a.K) >= 4 && (a.tB ? a.K : a.N) >= 4;
#define DP_GEMM_BODY(TA, TB)                                                          \
    do {                                                                              \
        if (quad) gemm_body<BM, BN, WAVES_M, WAVES_N, TA, TB, true>(a, tile, ks, lds);    \
        else gemm_body<BM, BN, WAVES_M, WAVES_N, TA, TB, false>(a, tile, ks, lds);        \
    } while (0)
    if (a.tA) {
        if (a.tB) DP_GEMM_BODY(true, true);
        else DP_GEMM_BODY(true, false);
    } else {
        if (a.tB) DP_GEMM_BODY(false, true);
        else DP_GEMM_BODY(false, false);
    }
#undef DP_GEMM_BODY
    if (threadIdx.x == 0) GEMM_STAMP_MAX(g.stamp_slot);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static void launch_tile(Seq& q, GemmGroupArgs& g, int batch) {
    int total = 0;
    for (int i = 0; i < g.count; ++i) {
        GemmArgs& a = g.p[i];
        a.tilesN = (a.N + BN - 1) / BN;
        g.tile0[i] = total;
        // with a split output the zero rows M..8*k8-1 of the operand must be written too
        const int mrows = a.split_out && a.split_k8 * 8 > a.M ? a.split_k8 * 8 : a.M;
        a.ntiles = ((mrows + BM - 1) / BM) * a.tilesN;
        total += a.ntiles * a.ksplit;
    }
    g.tile0[g.count] = total;
#ifdef DP_STAMP
    static int launch_no = 0;
    g.stamp_slot = launch_no++ & 63;
    {
        unsigned long long init[8] = {~0ULL, 0, 0, 0, 0, 0, 0,
                                      (unsigned long long)g.p[0].M | ((unsigned long long)g.p[0].N << 16) |
                                          ((unsigned long long)g.p[0].K << 32) | ((unsigned long long)g.count << 56)};
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_gemm_stamps), init, sizeof(init), g.stamp_slot * sizeof(init),
                                     hipMemcpyHostToDevice, q.stream);
    }
#endif
    hipLaunchKernelGGL((bgemm_kernel<BM, BN, WAVES_M, WAVES_N>), dim3(total, batch), dim3(256), 0,
                       q.stream, g);
}

void bgemm_group(Seq& q, const GemmDesc* d_in, int count, int batch, int ksplit) {
    if (!q.ok() || batch <= 0 || count <= 0) return;
    if (ksplit < 1) ksplit = 1;
    // big products between two general fp32 operands go to the split-bf16 kernel (dp_gemm_split.hip), one launch each;
    // what is left shares the grouped fp32-MFMA launch below
    GemmDesc rest[GEMM_GROUP_MAX];
    const GemmDesc* d = d_in;
    if (!q.pred && count <= GEMM_GROUP_MAX) {
        int nrest = 0;
        bool any = false;
        for (int i = 0; i < count; ++i) {
            if (gemm_split_usable(d_in[i], batch, d_in[i].nosplit ? 1 : ksplit)) {
                gemm_split_bf16(q, d_in[i], batch);
                any = true;
            } else {
                rest[nrest++] = d_in[i];
            }
        }
        if (any) {
            if (nrest == 0 || !q.ok()) return;
            d = rest;
            count = nrest;
        }
    }
    if (batch > 65535 || count > GEMM_GROUP_MAX) {
        set_error("bgemm_group: batch %d / count %d out of range", batch, count);
        q.err = DP_ERR_INVALID_ARG;
        return;
    }
    // Big batches: a group is one launch with ONE tile shape, and a 20-row or 20-column problem in a 64 x 64 tile spends
    // 3.2x its flops on padding — at B = 256, n = 1024 that padding, not memory, set the time of the GraphConv backward
    // groups (363 us for [20x20x1024 TN] [1024x20x20 NT] [20x256x1024 TN] [1024x20x256 NT]).  When the group is worth
    // more than a few launch floors it is split by shape class (short M / narrow N / the rest), one launch per class.
    if (count > 1 && !q.pred) {
        double flops = 0;
        for (int i = 0; i < count; ++i) flops += 2.0 * d[i].M * d[i].N * d[i].K * batch;
        auto cls = [](const GemmDesc& s) { return s.M <= 32 ? 0 : s.N <= 32 ? 1 : 2; };
        bool mixed = false;
        for (int i = 1; i < count; ++i) mixed = mixed || cls(d[i]) != cls(d[0]);
        if (mixed && flops > 0.5e9) {
            GemmDesc part[GEMM_GROUP_MAX];
            for (int c = 0; c < 3; ++c) {
                int np = 0;
                for (int i = 0; i < count; ++i)
                    if (cls(d[i]) == c) part[np++] = d[i];
                if (np > 0) bgemm_group(q, part, np, batch, ksplit);
            }
            return;
        }
    }
    GemmGroupArgs g{};
    g.pred = q.pred;
    if (q.fold_zero_p && !q.pred) {      // (a predicated launch may not run at all: leave it for an unconditional one)
        g.zero_p = static_cast<uint4*>(q.fold_zero_p);
        g.zero_n16 = q.fold_zero_n16;
    }
    int maxN = 0, maxM = 0;
    bool hooked = false;      // a problem carries the row-partial hook: its row must stay inside ONE wave (WN >= 32)
    for (int i = 0; i < count; ++i) {
        const GemmDesc& s = d[i];
        if (s.M <= 0 || s.N <= 0) continue;
        GemmArgs& a = g.p[g.count++];
        a = GemmArgs{s.A, s.B, s.C, s.bias, s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.sA, s.sB, s.sC, s.alpha, s.beta,
                     s.act, 0, s.tA ? 1 : 0, s.tB ? 1 : 0, s.nosplit ? 1 : ksplit, ksplit, s.sK, s.atomic,
                     s.split_out, s.split_ct, s.split_k8, s.split_c0, s.fix_part, s.fix_cnt, s.rp_xhat, s.rp_ldx,
                     s.rp_part, s.rp_G, s.rp_g};
        if (s.rp_part) {
            if (!gemm_rowpart_ok(s.N) || s.atomic || !s.nosplit || s.split_out || s.fix_part) {
                set_error("bgemm_group: row-partial hook on an unsupported problem (N=%d)", s.N);
                q.err = DP_ERR_INVALID_ARG;
                return;
            }
            hooked = true;
        }
        if (a.ksplit > 1 && a.sK == 0) {
            // ranges that start past K have nothing to add to a shared C (atomic or ticket combine): not launched
            const int kchunk = ((a.K + a.ksplit * KT - 1) / (a.ksplit * KT)) * KT;
            const int eff = (a.K + kchunk - 1) / kchunk;
            a.ksplit = eff < 1 ? 1 : eff;
        }
        if (s.N > maxN) maxN = s.N;
        if (s.M > maxM) maxM = s.M;
    }
    if (g.count == 0) return;
    if (g.zero_p) q.fold_zero_p = nullptr, q.fold_zero_n16 = 0;
    if (knobs().gemm_trace) {   // host-side shape log, one line per launch
        fprintf(stderr, "bgemm batch=%d ksplit=%d:", batch, ksplit);
        for (int i = 0; i < g.count; ++i)
            fprintf(stderr, " [%dx%dx%d %c%c%s%s]", g.p[i].M, g.p[i].N, g.p[i].K, g.p[i].tA ? 'T' : 'N',
                    g.p[i].tB ? 'T' : 'N', d[i].nosplit ? " nosplit" : "", d[i].atomic ? " atomic" : "");
        fprintf(stderr, "\n");
    }
    // Largest tile that still gives >= TARGET workgroups (256 CUs x 2); smallest tile otherwise.
    const long TARGET = knobs().gemm_target_wgs > 0 ? knobs().gemm_target_wgs : 512L;   // tuning knob
    auto wgs = [&](int bm, int bn) {
        long t = 0;
        for (int i = 0; i < g.count; ++i)
            t += (long)((g.p[i].M + bm - 1) / bm) * ((g.p[i].N + bn - 1) / bn) * batch;   // split-K not counted:
        // it exists to add parallelism to small-output problems, not to license bigger tiles
        return t;
    };
    // ... and never a tile taller than the tallest problem (rows of padding are MFMA time)
    if (maxN <= 16) {
        launch_tile<64, 16, 4, 1>(q, g, batch);
    } else if (maxN <= 32) {
        if (maxM > 64 && wgs(128, 32) >= TARGET) launch_tile<128, 32, 4, 1>(q, g, batch);
        else if (hooked || (maxM > 32 && wgs(64, 32) >= TARGET)) launch_tile<64, 32, 4, 1>(q, g, batch);
        else launch_tile<32, 32, 2, 2>(q, g, batch);
    } else {
        if (hooked || (maxM > 32 && wgs(64, 64) >= TARGET)) launch_tile<64, 64, 2, 2>(q, g, batch);
        else if (maxM > 16 && wgs(32, 64) >= TARGET) launch_tile<32, 64, 1, 4>(q, g, batch);
        else if (maxM > 16 && maxM <= 32) launch_tile<32, 64, 1, 4>(q, g, batch);
        else launch_tile<16, 64, 1, 4>(q, g, batch);
    }
    q.check_launch("bgemm");
}

void bgemm(Seq& q, const float* A, const float* B, float* C, const float* bias, int batch, int M, int N,
           int K, int lda, int ldb, int ldc, long sA, long sB, long sC, bool tA, bool tB, float alpha,
           float beta, int act) {
    GemmDesc d{A, B, C, bias, M, N, K, lda, ldb, ldc, sA, sB, sC, tA, tB, alpha, beta, act, 0, 0, nullptr, 0, 0, 0};
    bgemm_group(q, &d, 1, batch, 1);
}

#ifdef DP_STAMP
extern "C" __attribute__((visibility("default"))) int dp_debug_gemm_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 64 * 8);
}
#endif

}  // namespace dp
