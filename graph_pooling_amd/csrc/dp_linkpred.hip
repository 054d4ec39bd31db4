// Link-prediction auxiliary loss of DiffPool (SoftPoolingGcnEncoder.loss, encoders.py:1309-1331):
//   P = min(S S^T, 1);  l = -A log(P + eps) - (1 - A) log(1 - P + eps), eps = 1e-7;
//   zero outside the n_b x n_b block; loss = sum(l) / sum_b n_b^2.
//
// The reference materialises ~10 [B,N,N] fp32 temporaries for this (80 % of its CPU step).  Here
// nothing of size N^2 is ever written:
//   forward : one workgroup per 64x64 tile of one graph forms P_tile = S_r S_c^T on the fp32 MFMA from two
//             S row-blocks staged in LDS, reads the A tile once, and reduces the loss terms with wave shuffles
//             to one partial per workgroup (tiles outside the n_b x n_b block exit immediately);
//   backward: one workgroup per 64-row block of dS walks the column tiles, RECOMPUTES each P tile, forms
//             E = dl/dP(r,c) + dl/dP(c,r) (the A^T tile comes through LDS so both reads are coalesced) and
//             accumulates dS_r += E S_c on the MFMA — dS = (D + D^T) S without ever storing D.
#include "dp_common.h"

namespace dp {

#define LINK_EPS 1e-7f
typedef float lk_f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float lk_block_sum(float v, float* red) {
    v = wave64_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// LDS images of S row blocks are [rows][KP] with KW = 16 KT padded columns (zero beyond K) and KP = KW + 2, so the
// MFMA fragment read (16 rows x 2 k per 32-lane group) touches 32 distinct banks.
// (force-inlined: as a real call — which hipcc chose for KT >= 12 — its pointers are generic and every access a flat_load)
template <int KT, int ROWS>
__device__ __forceinline__ void lk_stage(const float* Sb, int lds_ld, int r0, int n, int K, float* dst) {
    constexpr int KW = KT * 16, KP = KW + 2;
    // loads first, selects and LDS writes after: a select right behind its load makes every load a round trip of
    // its own (the first version: one `s_waitcnt vmcnt(0)` per element, 32 serial round trips in the forward kernel)
    constexpr int NV = ROWS * KW / 256;
    float v[NV];
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int e = threadIdx.x + 256 * m;
        const int i = e / KW, k = e % KW;
        v[m] = Sb[(long)min(r0 + i, n - 1) * lds_ld + min(k, K - 1)];
    }
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int e = threadIdx.x + 256 * m;
        const int i = e / KW, k = e % KW;
        dst[i * KP + k] = (r0 + i < n && k < K) ? v[m] : 0.f;
    }
}

// P tile of a wave: rows wr*32.., cols wc*16*NJ.. ; acc[mi][ni] are 16x16 tiles
template <int KT, int NJ>
__device__ __forceinline__ void lk_ptile(const float* Sr, const float* Sc, int K, int wr, int wc, int l15, int kq,
                                lk_f32x4 (&acc)[2][NJ]) {
    constexpr int KP = KT * 16 + 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (lk_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {          // columns K..KW are zero in both images
        float a[2], b[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = Sr[(wr * 32 + i * 16 + l15) * KP + k0 + kq];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = Sc[(wc * 16 * NJ + j * 16 + l15) * KP + k0 + kq];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------ forward
template <int KT>
__global__ __launch_bounds__(256) void k_link_fwd(const float* S, int lds_ld, const float* adj, const int* num_nodes,
                                                  int n, int K, int tiles, float* partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KP = KT * 16 + 2;
    const int b = blockIdx.y;
    const int tr = blockIdx.x / tiles, tc = blockIdx.x % tiles;
    const int nb = num_nodes ? min(num_nodes[b], n) : n;
    const int r0 = tr * 64, c0 = tc * 64;
    if (r0 >= nb || c0 >= nb) {
        if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = 0.f;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, kq = lane >> 4;
    // the A values this thread needs, issued first (clamped addresses, no branches) so they fly under the staging
    const float* Ab = adj + (long)b * n * n;
    float av[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(r0 + wr * 32 + i * 16 + kq * 4 + r, n - 1);
                const int c = min(c0 + wc * 32 + j * 16 + l15, n - 1);
                av[i][j][r] = Ab[(long)row * n + c];
            }
    float* Sr = lds;
    float* Sc = lds + 64 * KP;
    float* red = Sc + 64 * KP;
    const float* Sb = S + (long)b * n * lds_ld;
    lk_stage<KT, 64>(Sb, lds_ld, r0, n, K, Sr);
    lk_stage<KT, 64>(Sb, lds_ld, c0, n, K, Sc);
    __syncthreads();
    lk_f32x4 acc[2][2];
    lk_ptile<KT, 2>(Sr, Sc, K, wr, wc, l15, kq, acc);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 32 + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + wr * 32 + i * 16 + kq * 4 + r;
                const float pv = fminf(acc[i][j][r], 1.f);
                const float a = av[i][j][r];
                const float l = -a * logf(pv + LINK_EPS) - (1.f - a) * logf(1.f - pv + LINK_EPS);
                sum += (row < nb && c < nb) ? l : 0.f;
            }
        }
    const float s = lk_block_sum(sum, red);
    if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = s;
}

// loss = sum(partials) / sum_b n_b^2  (single block; deterministic);  also the backward scale dloss / sum n_b^2
// norm (optional device scalar) replaces the local sum_b n_b^2: data-parallel ranks pass (global sum) / world so that
// the mean of the per-rank losses and gradients equals the single-batch loss of encoders.py:1326,1331
__global__ __launch_bounds__(256) void k_link_final(const float* partial, int count, const int* num_nodes, int B,
                                                    int n, float* out, float* scale_out, const float* dloss,
                                                    const float* norm) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial ? partial[i] : 0.f;
    const float s = lk_block_sum(acc, red);
    if (threadIdx.x == 0) {
        double nn = 0.0;
        for (int b = 0; b < B; ++b) {
            const double v = num_nodes ? (double)min(num_nodes[b], n) : (double)n;
            nn += v * v;
        }
        if (norm) nn = (double)norm[0];
        if (out) out[0] = (float)((double)s / nn);
        if (scale_out) scale_out[0] = (float)((dloss ? (double)dloss[0] : 1.0) / nn);
    }
}

// ------------------------------------------------------------------ backward
// d l / d P at (a, p_raw); torch.min(a, b) backward: full gradient where a < b, half on ties, none above
__device__ inline float lk_dldp(float av, float raw) {
    const float pv = fminf(raw, 1.f);
    const float gate = raw < 1.f ? 1.f : (raw == 1.f ? 0.5f : 0.f);
    return gate * (-av / (pv + LINK_EPS) + (1.f - av) / (1.f - pv + LINK_EPS));
}

// KT = ceil(K / 16) output column tiles per row block; the column tiles walked are CW = 32 NJ wide.
// The next column tile's S rows and both A tiles are fetched into registers while the current one is computed on.
template <int KT, int NJ>
__global__ __launch_bounds__(256) void k_link_bwd(const float* S, int lds_ld, const float* adj, const int* num_nodes,
                                                  const float* scale_ptr, float* dS, int ldds, int n, int K,
                                                  int accumulate, float* part, int split_tiles) {
    // part != null: blockIdx.z walks only `split_tiles` column tiles and stores its partial row block to
    // part[z][b][row][K]; k_link_reduce sums the splits in a fixed order (small batches: 160 workgroups walking 8
    // tiles each left most of the chip idle)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CW = 32 * NJ, KW = KT * 16, KP = KW + 2;
    constexpr int SA = CW + 4;      // E / A tile row stride: 4 SA = 16 mod 32, so (4 kq + r) rows x 16 cols spread over banks
    constexpr int ST = 65;          // transposed A tile
    constexpr int NS = CW * KW / 256, NA = CW / 4;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * 64;
    const int nb = num_nodes ? min(num_nodes[b], n) : n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, kq = lane >> 4;
    float* dSb = dS + (long)b * n * ldds;
    if (r0 >= nb) {                                   // rows outside the graph: gradient is zero
        if (part) return;                             // (the reduce pass writes those zeros)
        if (!accumulate)
            for (int e = threadIdx.x; e < 64 * KW; e += 256) {
                const int i = e / KW, k = e % KW;
                if (r0 + i < n && k < K) dSb[(long)(r0 + i) * ldds + k] = 0.f;
            }
        return;
    }
    float* Sr = lds;                 // [64][KP]
    float* Sc = Sr + 64 * KP;        // [CW][KP]
    float* Ar = Sc + CW * KP;        // [64][SA]   A[r0 + i][c0 + j], overwritten in place by E[i][j]
    float* At = Ar + 64 * SA;        // [CW][ST]   A[c0 + j][r0 + i]
    const float* Sb = S + (long)b * n * lds_ld;
    const float* Ab = adj + (long)b * n * n;
    const float scale = scale_ptr[0];

    float sc[NS], ar[NA], at[NA];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            const int e = threadIdx.x + 256 * m;
            const int i = e / KW, k = e % KW;
            sc[m] = Sb[(long)min(c0 + i, n - 1) * lds_ld + min(k, K - 1)];    // raw: zeroed when it is written to LDS
        }
#pragma unroll
        for (int m = 0; m < NA; ++m) {
            const int e = threadIdx.x + 256 * m;
            ar[m] = Ab[(long)min(r0 + e / CW, n - 1) * n + min(c0 + e % CW, n - 1)];
            at[m] = Ab[(long)min(c0 + (e >> 6), n - 1) * n + min(r0 + (e & 63), n - 1)];
        }
    };
    const int cbeg = part ? (int)blockIdx.z * split_tiles * CW : 0;
    const int cend = part ? min(nb, cbeg + split_tiles * CW) : nb;
    fetch(min(cbeg, max(nb - 1, 0)));
    lk_stage<KT, 64>(Sb, lds_ld, r0, n, K, Sr);
    // this wave's share of the output block: rows wave*16.., all KT column tiles
    lk_f32x4 out[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) out[t] = (lk_f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = cbeg; c0 < cend; c0 += CW) {
        __syncthreads();                              // previous iteration's readers of Sc / Ar / At are done
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            const int e = threadIdx.x + 256 * m;
            const int i = e / KW, k = e % KW;
            Sc[i * KP + k] = (c0 + i < n && k < K) ? sc[m] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < NA; ++m) {
            const int e = threadIdx.x + 256 * m;
            Ar[(e / CW) * SA + e % CW] = ar[m];
            At[(e >> 6) * ST + (e & 63)] = at[m];
        }
        __syncthreads();
        if (c0 + CW < cend) fetch(c0 + CW);
        lk_f32x4 acc[2][NJ];
        lk_ptile<KT, NJ>(Sr, Sc, K, wr, wc, l15, kq, acc);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int cj = wc * 16 * NJ + j * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ri = wr * 32 + i * 16 + kq * 4 + r;
                    const float raw = acc[i][j][r];
                    const float e = scale * (lk_dldp(Ar[ri * SA + cj], raw) + lk_dldp(At[cj * ST + ri], raw));
                    Ar[ri * SA + cj] = (r0 + ri < nb && c0 + cj < nb) ? e : 0.f;
                }
            }
        __syncthreads();
        // out[16 rows of this wave][K] += E[rows][CW] · Sc[CW][K]
#pragma unroll 4
        for (int k0 = 0; k0 < CW; k0 += 4) {
            const float ev = Ar[(wave * 16 + l15) * SA + k0 + kq];
#pragma unroll
            for (int t = 0; t < KT; ++t)
                out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev, Sc[(k0 + kq) * KP + t * 16 + l15], out[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int col = t * 16 + l15;
        if (col >= K) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + wave * 16 + kq * 4 + r;
            if (row < n) {
                if (part) {
                    part[(((long)blockIdx.z * gridDim.y + b) * n + row) * K + col] = out[t][r];
                } else {
                    float* p = dSb + (long)row * ldds + col;
                    *p = accumulate ? *p + out[t][r] : out[t][r];
                }
            }
        }
    }
}

// dS[b, row, :] = (accumulate ? dS : 0) + sum_z part[z][b][row][:]   (rows >= n_b: the sum is empty)
__global__ __launch_bounds__(256) void k_link_reduce(const float* part, int splits, const int* num_nodes, float* dS,
                                                     int ldds, int B, int n, int K, int accumulate) {
    const long total = (long)B * n * K;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k = (int)(e % K);
        const long row = e / K;
        const int b = (int)(row / n), r = (int)(row % n);
        const int nb = num_nodes ? min(num_nodes[b], n) : n;
        float v = 0.f;
        if ((r / 64) * 64 < nb)                       // the row block ran: every split stored its partial
            for (int z = 0; z < splits; ++z) v += part[(long)z * total + e];
        float* p = dS + row * ldds + k;
        *p = accumulate ? *p + v : v;
    }
}

static int lk_kt(int K) {
    const int kt = (K + 15) / 16;
    static const int steps[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int s : steps)
        if (kt <= s) return s;
    return 0;
}

// every instantiation is raised once per device to what it can ever need (static + dynamic LDS stay within 160 KiB)

template <int KT>
static void launch_link_fwd(Seq& q, const float* S, int lds_ld, const float* adj, const int* num_nodes, int B, int n,
                            int K, int tiles, float* partial) {
    constexpr size_t bytes = ((size_t)2 * 64 * (KT * 16 + 2) + 4) * sizeof(float);
    static DynLdsOnce attr;
    if (bytes > 64 * 1024) ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_link_fwd<KT>), (int)bytes, "k_link_fwd");
    if (!q.ok()) return;
    hipLaunchKernelGGL((k_link_fwd<KT>), dim3(tiles * tiles, B), dim3(256), bytes, q.stream, S, lds_ld, adj, num_nodes,
                       n, K, tiles, partial);
}

void linkpred_fwd(Seq& q, const float* S, int lds_ld, const float* adj, const int* num_nodes, float* loss_out, int B,
                  int n, int K, const float* norm) {
    if (q.err) return;
    const int tiles = (n + 63) / 64;
    float* partial = q.alloc<float>((size_t)B * tiles * tiles);
    if (!q.ok()) return;
    const int kt = lk_kt(K);
    if (!kt) {
        set_error("linkpred: K=%d clusters exceed the fused tile kernel (max 256)", K);
        q.err = DP_ERR_UNSUPPORTED;
        return;
    }
#define LK_FWD(T) \
    case T: launch_link_fwd<T>(q, S, lds_ld, adj, num_nodes, B, n, K, tiles, partial); break;
    switch (kt) { LK_FWD(1) LK_FWD(2) LK_FWD(3) LK_FWD(4) LK_FWD(6) LK_FWD(8) LK_FWD(12) LK_FWD(16) }
#undef LK_FWD
    q.check_launch("link_fwd");
    hipLaunchKernelGGL(k_link_final, dim3(1), dim3(256), 0, q.stream, partial, B * tiles * tiles, num_nodes, B, n,
                       loss_out, (float*)nullptr, (const float*)nullptr, norm);
    q.check_launch("link_final");
}

template <int KT, int NJ>
static void launch_link_bwd(Seq& q, const float* S, int lds_ld, const float* adj, const int* num_nodes,
                            const float* scale, float* dS, int ldds, int B, int n, int K, int accumulate, float* part,
                            int splits, int split_tiles) {
    constexpr int CW = 32 * NJ, KP = KT * 16 + 2;
    constexpr size_t bytes = ((size_t)(64 + CW) * KP + 64 * (CW + 4) + CW * 65) * sizeof(float);
    static_assert(bytes <= 160 * 1024, "link_bwd LDS");
    static DynLdsOnce attr;
    if (bytes > 64 * 1024)
        ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_link_bwd<KT, NJ>), (int)bytes, "k_link_bwd");
    if (!q.ok()) return;
    hipLaunchKernelGGL((k_link_bwd<KT, NJ>), dim3((n + 63) / 64, B, part ? splits : 1), dim3(256), bytes, q.stream, S,
                       lds_ld, adj, num_nodes, scale, dS, ldds, n, K, accumulate, part, split_tiles);
}

void linkpred_bwd(Seq& q, const float* S, int lds_ld, const float* adj, const int* num_nodes, const float* dloss,
                  float* dS, int ldds, int B, int n, int K, int accumulate, const float* norm) {
    if (q.err) return;
    float* scale = q.alloc<float>(64);
    const int kt = lk_kt(K);
    // small batches: split the column tiles over extra workgroups (>= ~512 in all) and sum the partials afterwards
    const int cw = kt >= 12 ? 32 : 64;
    const int col_tiles = (n + cw - 1) / cw;
    const long base_wgs = (long)((n + 63) / 64) * B;
    int splits = (int)((512 + base_wgs - 1) / base_wgs);
    splits = splits < 1 ? 1 : (splits > col_tiles ? col_tiles : splits);
    const int split_tiles = (col_tiles + splits - 1) / splits;
    splits = (col_tiles + split_tiles - 1) / split_tiles;
    float* part = splits > 1 ? q.alloc<float>((size_t)splits * B * n * K) : nullptr;
    if (!q.ok()) return;
    if (!kt) {
        set_error("linkpred backward: K=%d clusters exceed the fused tile kernel (max 256)", K);
        q.err = DP_ERR_UNSUPPORTED;
        return;
    }
    hipLaunchKernelGGL(k_link_final, dim3(1), dim3(256), 0, q.stream, (const float*)nullptr, 0, num_nodes, B, n,
                       (float*)nullptr, scale, dloss, norm);
    q.check_launch("link_scale");
#define LK_BWD(T, J)                                                                                             \
    case T:                                                                                                      \
        launch_link_bwd<T, J>(q, S, lds_ld, adj, num_nodes, scale, dS, ldds, B, n, K, accumulate, part, splits,  \
                              split_tiles);                                                                      \
        break;
    switch (kt) { LK_BWD(1, 2) LK_BWD(2, 2) LK_BWD(3, 2) LK_BWD(4, 2) LK_BWD(6, 2) LK_BWD(8, 2) LK_BWD(12, 1) LK_BWD(16, 1) }
#undef LK_BWD
    q.check_launch("link_bwd");
    if (part) {
        const long total = (long)B * n * K;
        long blocks = (total + 255) / 256;
        blocks = blocks > 2048 ? 2048 : blocks;
        hipLaunchKernelGGL(k_link_reduce, dim3((int)blocks), dim3(256), 0, q.stream, part, splits, num_nodes, dS, ldds, B,
                           n, K, accumulate);
        q.check_launch("link_reduce");
    }
}

}  // namespace dp
