// Row-wise / reduction kernels of the DiffPool path.  Rows here are short (20..276 floats),
// so a row is handled by a 16-lane team (4 teams per wavefront): loads stay coalesced across
// the consecutive rows of a wave and reductions are four xor-shuffles.
#include "dp_common.h"

namespace dp {

#define L2_EPS 1e-12f
#define BN_EPS 1e-5f

// Wide rows (a group wider than 128 columns, every group width a multiple of 4): the lane's share of a row moves as
// 16-byte quads — lane tl holds quads tl, tl + 16, ... — a quarter of the memory instructions of the 4-byte form,
// which set the pace of these kernels at 256-column rows (k_rownorm_fwd 330 us for 580 MB at the ER shape).  Global
// dwordx4 accesses need only dword alignment on gfx9 under HSA, so row starts need not be 16-byte aligned.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ inline float team_sum(float v) {
    return row16_sum(v);
}
__device__ inline float team_max(float v) {
    return row16_max(v);
}

static inline bool row_quads_ok(const RowGroups& g) {
    if (knobs().no_row_quads) return false;
    for (int i = 0; i < g.G; ++i)
        if (g.w[i] < 4 || (g.w[i] & 3)) return false;
    return true;
}
static inline int team_grid(long items) {
    long blocks = (items + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ------------------------------------------------------------------ rownorm fwd
// u = U[row, c0+c] (+ P[row, c0+c]) (+ bias[c]);  y = u / max(||u||, 1e-12)   (encoders.py:966-972)
// optional BN partials of relu(y): (row mean, row M2) for a Chan-combine over the batch.
struct RownormFwdArgs {
    const float* U;
    int ldu;
    const float* P;
    GroupCPtrs bias;
    RowGroups g;
    GroupPtrs yout;
    float* invn;
    float* part;
    long rows;
    int normalize, stats_mode;
};

// NK > 0: group widths <= 16 NK — the row (+ add_self operand, + bias) is read ONCE into registers; the three passes
// of the generic form (norm, write, statistics) re-read it from memory: 342 us against ~150 for the 276-wide assign
// layer at the ER shape.
template <int NK, bool Q4 = false>
__global__ __launch_bounds__(256) void k_rownorm_fwd(RownormFwdArgs a) {
    const int tl = threadIdx.x & 15;
    const long team = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long nteams = (long)gridDim.x * 16;
    const long items = a.rows * a.g.G;
    for (long it = team; it < items; it += nteams) {
        const long row = it / a.g.G;
        const int g = (int)(it % a.g.G);
        const int c0 = a.g.c0[g], w = a.g.w[g];
        const float* u = a.U + row * a.ldu + c0;
        const float* p = a.P ? a.P + row * a.ldu + c0 : nullptr;
        const float* bias = a.bias.p[g];
        float* y = a.yout.p[g] + row * a.yout.ld[g];
        if constexpr (Q4) {
            constexpr int NQ = NK > 0 ? NK : 1;      // quads per lane
            const int nq = w >> 2;
            f4u v[NQ];
            const float* pp = p ? p : u;             // valid addresses either way: no branch around a load
            const float* bb = bias ? bias : u;
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int qd = min(tl + 16 * k, nq - 1) * 4;
                const f4u t0 = *reinterpret_cast<const f4u*>(u + qd);
                const f4u t1 = *reinterpret_cast<const f4u*>(pp + qd);
                const f4u t2 = *reinterpret_cast<const f4u*>(bb + qd);
                f4u t = t0;
                if (p) t += t1;
                if (bias) t += t2;
                v[k] = tl + 16 * k < nq ? t : (f4u){0.f, 0.f, 0.f, 0.f};
            }
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < NQ; ++k) ss += v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2] + v[k][3] * v[k][3];
            ss = team_sum(ss);
            const float inv = a.normalize ? 1.f / fmaxf(sqrtf(ss), L2_EPS) : 1.f;
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                v[k] *= inv;
                if (tl + 16 * k < nq) *reinterpret_cast<f4u*>(y + (tl + 16 * k) * 4) = v[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) s1 += a.stats_mode == 1 ? fmaxf(v[k][j], 0.f) : v[k][j];
            }
            if (tl == 0 && a.invn) a.invn[it] = inv;
            if (a.stats_mode && a.part) {
                s1 = team_sum(s1);
                const float mean = s1 / (float)w;
                float m2 = 0.f;
#pragma unroll
                for (int k = 0; k < NQ; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = a.stats_mode == 1 ? fmaxf(v[k][j], 0.f) : v[k][j];
                        t -= mean;
                        m2 += tl + 16 * k < nq ? t * t : 0.f;
                    }
                m2 = team_sum(m2);
                if (tl == 0) {
                    a.part[it * 2 + 0] = mean;
                    a.part[it * 2 + 1] = m2;
                }
            }
            continue;
        }
        if (NK > 0) {
            constexpr int NKK = NK > 0 ? NK : 1;
            float v[NKK];
#pragma unroll
            for (int k = 0; k < NKK; ++k) {
                const int c = min(tl + 16 * k, w - 1);
                float t = u[c];
                if (p) t += p[c];
                if (bias) t += bias[c];
                v[k] = tl + 16 * k < w ? t : 0.f;
            }
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < NKK; ++k) ss += v[k] * v[k];
            ss = team_sum(ss);
            const float inv = a.normalize ? 1.f / fmaxf(sqrtf(ss), L2_EPS) : 1.f;
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < NKK; ++k) {
                v[k] *= inv;
                if (tl + 16 * k < w) y[tl + 16 * k] = v[k];
                s1 += a.stats_mode == 1 ? fmaxf(v[k], 0.f) : v[k];
            }
            if (tl == 0 && a.invn) a.invn[it] = inv;
            if (a.stats_mode && a.part) {
                s1 = team_sum(s1);
                const float mean = s1 / (float)w;
                float m2 = 0.f;
#pragma unroll
                for (int k = 0; k < NKK; ++k) {
                    float t = a.stats_mode == 1 ? fmaxf(v[k], 0.f) : v[k];
                    t -= mean;
                    m2 += tl + 16 * k < w ? t * t : 0.f;
                }
                m2 = team_sum(m2);
                if (tl == 0) {
                    a.part[it * 2 + 0] = mean;
                    a.part[it * 2 + 1] = m2;
                }
            }
            continue;
        }
        float ss = 0.f;
        for (int c = tl; c < w; c += 16) {
            float v = u[c];
            if (p) v += p[c];
            if (bias) v += bias[c];
            ss += v * v;
        }
        ss = team_sum(ss);
        float inv = 1.f;
        if (a.normalize) inv = 1.f / fmaxf(sqrtf(ss), L2_EPS);
        float s1 = 0.f;
        for (int c = tl; c < w; c += 16) {
            float v = u[c];
            if (p) v += p[c];
            if (bias) v += bias[c];
            v *= inv;
            y[c] = v;
            s1 += a.stats_mode == 1 ? fmaxf(v, 0.f) : v;
        }
        if (tl == 0 && a.invn) a.invn[it] = inv;
        if (a.stats_mode && a.part) {
            s1 = team_sum(s1);
            const float mean = s1 / (float)w;
            float m2 = 0.f;
            for (int c = tl; c < w; c += 16) {
                float v = u[c];
                if (p) v += p[c];
                if (bias) v += bias[c];
                v *= inv;
                if (a.stats_mode == 1) v = fmaxf(v, 0.f);
                v -= mean;
                m2 += v * v;
            }
            m2 = team_sum(m2);
            if (tl == 0) {
                a.part[it * 2 + 0] = mean;
                a.part[it * 2 + 1] = m2;
            }
        }
    }
}

void rownorm_fwd(Seq& q, const float* U, int ldu, const float* P, GroupCPtrs bias, RowGroups g, GroupPtrs yout,
                 float* invn, float* part, long rows, int normalize, int stats_mode) {
    if (!q.ok() || rows <= 0) return;
    RownormFwdArgs a{U, ldu, P, bias, g, yout, invn, part, rows, normalize, stats_mode};
    const int maxw = g.G == 2 && g.w[1] > g.w[0] ? g.w[1] : g.w[0];
    const dim3 grid(team_grid(rows * g.G));
    if (maxw > 128 && maxw <= 512 && row_quads_ok(g)) {
        if (maxw <= 256) hipLaunchKernelGGL((k_rownorm_fwd<4, true>), grid, dim3(256), 0, q.stream, a);
        else if (maxw <= 320) hipLaunchKernelGGL((k_rownorm_fwd<5, true>), grid, dim3(256), 0, q.stream, a);
        else hipLaunchKernelGGL((k_rownorm_fwd<8, true>), grid, dim3(256), 0, q.stream, a);
        q.check_launch("rownorm_fwd");
        return;
    }
    if (maxw <= 32) hipLaunchKernelGGL(k_rownorm_fwd<2>, grid, dim3(256), 0, q.stream, a);
    else if (maxw <= 64) hipLaunchKernelGGL(k_rownorm_fwd<4>, grid, dim3(256), 0, q.stream, a);
    else if (maxw <= 128) hipLaunchKernelGGL(k_rownorm_fwd<8>, grid, dim3(256), 0, q.stream, a);
    else if (maxw <= 320) hipLaunchKernelGGL(k_rownorm_fwd<20>, grid, dim3(256), 0, q.stream, a);
    else hipLaunchKernelGGL(k_rownorm_fwd<0>, grid, dim3(256), 0, q.stream, a);
    q.check_launch("rownorm_fwd");
}

// ------------------------------------------------------------------ widening product + GraphConv tail
// y = l2norm((A x) W + bias) for a layer run in the reference's own order (dp_model.hip layer_agg_first): the
// aggregated input row (20-40 floats) is all a row needs, so the product rides in the row kernel — the row's inputs
// sit in two registers per lane and reach every lane of the 16-lane team by DPP row_share, the weights (22 KB at
// 20 -> 256) are staged once per workgroup in LDS and read as 16-byte quads — and the 276-column pre-activation is
// never written: 132 us (GEMM, store-bound at 64-byte segments) + 136 us (k_rownorm_fwd) -> one pass at the output's
// write rate.  Row groups as in k_rownorm_fwd's quad form; din <= 32 per group; no BatchNorm statistics (last layer).
struct WidenFwdArgs {
    RownormFwdArgs r;        // bias, output groups, yout, invn, rows, normalize (U, P, part unused)
    const float* Uin;        // [rows, ldin]: group g's inputs at columns c0in[g] .. c0in[g] + din[g]
    int ldin;
    int c0in[2], din[2];
    const float* W[2];       // [din_g][w_g] row-major
};
template <int K>
__device__ __forceinline__ float team_share(float v) {      // lane K of the 16-lane row, to every lane of the row
    return dpp_src<0x150 + K, 0xF, true>(0.f, v);
}
typedef float f4a __attribute__((ext_vector_type(4)));       // 16-byte aligned: LDS rows of W
// KMAX = 20 (hidden_dim's default: no padding) or 32: the contraction is unrolled KMAX deep with no branch (a guarded
// step per k put every LDS read in its own basic block, each with a full wait: 468 us); weight rows din .. KMAX - 1
// are zero in LDS, and a team's lanes past din hold zeros.
template <int NQ, int KMAX>
__device__ __forceinline__ void widen_item(const WidenFwdArgs& a, const float* Wg, long row, int g, int tl) {
    const int G = a.r.g.G;
    const int w = a.r.g.w[g], nq = w >> 2, din = a.din[g];
    const float* uin = a.Uin + row * a.ldin + a.c0in[g];
    const float* bias = a.r.bias.p[g];
    float* y = a.r.yout.p[g] + row * a.r.yout.ld[g];
    float u0 = uin[min(tl, din - 1)], u1 = uin[min(tl + 16, din - 1)];
    u0 = tl < din ? u0 : 0.f;
    u1 = tl + 16 < din ? u1 : 0.f;
    f4a v[NQ];
    const float* wq[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int qoff = min(tl + 16 * k, nq - 1) * 4;
        wq[k] = Wg + qoff;
        const f4u bq = *reinterpret_cast<const f4u*>(bias ? bias + qoff : a.W[g]);   // (a valid address either way)
        v[k] = bias ? (f4a)bq : (f4a){0.f, 0.f, 0.f, 0.f};
    }
#define DP_WIDEN_STEP(KK)                                                                    \
    if (KK < KMAX) {                                                                         \
        const float uk = KK < 16 ? team_share<KK & 15>(u0) : team_share<KK & 15>(u1);        \
        _Pragma("unroll") for (int k = 0; k < NQ; ++k)                                       \
            v[k] += uk * *reinterpret_cast<const f4a*>(wq[k] + KK * w);                      \
    }
    DP_WIDEN_STEP(0) DP_WIDEN_STEP(1) DP_WIDEN_STEP(2) DP_WIDEN_STEP(3) DP_WIDEN_STEP(4) DP_WIDEN_STEP(5)
    DP_WIDEN_STEP(6) DP_WIDEN_STEP(7) DP_WIDEN_STEP(8) DP_WIDEN_STEP(9) DP_WIDEN_STEP(10) DP_WIDEN_STEP(11)
    DP_WIDEN_STEP(12) DP_WIDEN_STEP(13) DP_WIDEN_STEP(14) DP_WIDEN_STEP(15) DP_WIDEN_STEP(16) DP_WIDEN_STEP(17)
    DP_WIDEN_STEP(18) DP_WIDEN_STEP(19) DP_WIDEN_STEP(20) DP_WIDEN_STEP(21) DP_WIDEN_STEP(22) DP_WIDEN_STEP(23)
    DP_WIDEN_STEP(24) DP_WIDEN_STEP(25) DP_WIDEN_STEP(26) DP_WIDEN_STEP(27) DP_WIDEN_STEP(28) DP_WIDEN_STEP(29)
    DP_WIDEN_STEP(30) DP_WIDEN_STEP(31)
#undef DP_WIDEN_STEP
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        if (tl + 16 * k >= nq) v[k] = (f4a){0.f, 0.f, 0.f, 0.f};
        ss += v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2] + v[k][3] * v[k][3];
    }
    ss = team_sum(ss);
    const float inv = a.r.normalize ? 1.f / fmaxf(sqrtf(ss), L2_EPS) : 1.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        if (tl + 16 * k < nq) *reinterpret_cast<f4u*>(y + (tl + 16 * k) * 4) = (f4u)(v[k] * inv);
    if (tl == 0 && a.r.invn) a.r.invn[row * G + g] = inv;
}
// KMAX = 20 (hidden_dim's default: no padding) or 32: the contraction is unrolled KMAX deep with no branch (a guarded
// step per k put every LDS read in its own basic block, each with a full wait: 468 us); weight rows din .. KMAX - 1
// are zero in LDS, and a team's lanes past din hold zeros.  Items run group by group (all rows of group 0, then all of
// group 1), so the teams of a wave share a group and a group of at most 64 columns takes the one-quad form: its
// 20-column rows cost a quarter of the LDS reads of the 256-column ones instead of the same.
template <int NQ, int KMAX>
__global__ __launch_bounds__(256) void k_widen_fwd(WidenFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int G = a.r.g.G;
    const int wcnt0 = KMAX * a.r.g.w[0], wcnt1 = G == 2 ? KMAX * a.r.g.w[1] : 0;
    for (int e = threadIdx.x * 4; e < wcnt0 + wcnt1; e += 1024) {          // (both counts are multiples of 4)
        const int g = e < wcnt0 ? 0 : 1;
        const int o = g ? e - wcnt0 : e;
        const bool live = o < a.din[g] * a.r.g.w[g];
        const f4u t = *reinterpret_cast<const f4u*>(a.W[g] + (live ? o : 0));
        *reinterpret_cast<f4a*>(wl + e) = live ? t : (f4u){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int tl = threadIdx.x & 15;
    const long team = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long nteams = (long)gridDim.x * 16;
    const long items = a.r.rows * G;
    for (long it = team; it < items; it += nteams) {
        const int g = it >= a.r.rows ? 1 : 0;
        const long row = g ? it - a.r.rows : it;
        const float* Wg = wl + (g ? wcnt0 : 0);
        if (a.r.g.w[g] <= 64) widen_item<1, KMAX>(a, Wg, row, g, tl);
        else widen_item<NQ, KMAX>(a, Wg, row, g, tl);
    }
}
bool widen_fwd_supported(RowGroups g, const int din[2]) {
    if (knobs().no_widen_fusion || !row_quads_ok(g)) return false;
    int maxw = 0, wsum = 0;
    for (int i = 0; i < g.G; ++i) {
        if (din[i] < 1 || din[i] > 32) return false;
        maxw = g.w[i] > maxw ? g.w[i] : maxw;
        wsum += g.w[i];
    }
    // (weight rows are padded to 32 in LDS when some din > 20)
    return maxw > 128 && maxw <= 320 && (size_t)32 * wsum * sizeof(float) <= 60 * 1024;
}
void widen_fwd(Seq& q, const float* Uin, int ldin, const int c0in[2], const int din[2], const float* const W[2],
               GroupCPtrs bias, RowGroups g, GroupPtrs yout, float* invn, long rows, int normalize) {
    if (!q.ok() || rows <= 0) return;
    WidenFwdArgs a{};
    a.r = RownormFwdArgs{nullptr, 0, nullptr, bias, g, yout, invn, nullptr, rows, normalize, 0};
    a.Uin = Uin;
    a.ldin = ldin;
    int maxw = 0, maxd = 0, wsum = 0;
    for (int i = 0; i < 2; ++i) {
        a.c0in[i] = c0in[i];
        a.din[i] = i < g.G ? din[i] : 0;
        a.W[i] = W[i];
        if (i < g.G) {
            maxw = g.w[i] > maxw ? g.w[i] : maxw;
            maxd = din[i] > maxd ? din[i] : maxd;
            wsum += g.w[i];
        }
    }
    const dim3 grid(team_grid(rows * g.G));
    const int kmax = maxd <= 20 ? 20 : 32;
    const size_t lds = (size_t)kmax * wsum * sizeof(float);
    if (maxw <= 256 && kmax == 20) hipLaunchKernelGGL((k_widen_fwd<4, 20>), grid, dim3(256), lds, q.stream, a);
    else if (maxw <= 256) hipLaunchKernelGGL((k_widen_fwd<4, 32>), grid, dim3(256), lds, q.stream, a);
    else if (kmax == 20) hipLaunchKernelGGL((k_widen_fwd<5, 20>), grid, dim3(256), lds, q.stream, a);
    else hipLaunchKernelGGL((k_widen_fwd<5, 32>), grid, dim3(256), lds, q.stream, a);
    q.check_launch("widen_fwd");
}

// ------------------------------------------------------------------ bn apply fwd
// apply_bn (encoders.py:1048-1052): per node index n, statistics over (batch, feature), biased variance,
// eps 1e-5.  Each row team combines the B per-row partials (row mean, row M2) of its node index itself
// (Chan's parallel variance), so no separate finalize launch is needed; the b == 0 team stores
// (mu, rstd) for the backward pass.  x = (relu(y) - mu_n) * rstd_n, written into the concat buffer.
struct BnApplyArgs {
    const float* Y;
    int ldy;
    const float* part;   // [B, n, G, 2] or null (no BN)
    float* stats;        // [n, G, 2] out
    RowGroups g;
    GroupPtrs xout;
    int B, n;
    int relu;
    int stats_ready;     // big batches: (mu, rstd) were finalised by k_bn_finalize, read them instead of combining
    int Bs;              // graphs the statistics span = partial blocks in `part` (B, or B x ranks under sync-BN)
};
// big batches: one team per (node, group) combines the B partials once (the apply kernel would otherwise repeat
// that B-term reduction in every one of the B rows of the node)
__global__ __launch_bounds__(256) void k_bn_finalize(BnApplyArgs a) {
    const int tl = threadIdx.x & 15;
    const long it = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (it >= (long)a.n * a.g.G) return;
    const int g = (int)(it % a.g.G);
    const int w = a.g.w[g];
    const long pstride = (long)a.n * a.g.G * 2;
    const float* p = a.part + it * 2;
    float sm = 0.f;
    for (int b = tl; b < a.Bs; b += 16) sm += p[b * pstride];
    const float mu = team_sum(sm) / (float)a.Bs;
    float s2 = 0.f;
    for (int b = tl; b < a.Bs; b += 16) {
        const float d = p[b * pstride] - mu;
        s2 += p[b * pstride + 1] + (float)w * d * d;
    }
    const float var = team_sum(s2) / ((float)a.Bs * (float)w);
    if (tl == 0) {
        a.stats[it * 2] = mu;
        a.stats[it * 2 + 1] = 1.0f / sqrtf(var + BN_EPS);
    }
}
// NK > 0: group widths <= 16 NK; the row is asked for BEFORE the statistics are reduced, not after.
template <int NK>
__global__ __launch_bounds__(256) void k_bn_apply_fwd(BnApplyArgs a) {
    const int tl = threadIdx.x & 15;
    const long team = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long nteams = (long)gridDim.x * 16;
    const long items = (long)a.B * a.n * a.g.G;
    for (long it = team; it < items; it += nteams) {
        const long row = it / a.g.G;
        const int g = (int)(it % a.g.G);
        const int node = (int)(row % a.n);
        const int w = a.g.w[g];
        const float* y = a.Y + row * a.ldy + a.g.c0[g];
        constexpr int NKK = NK > 0 ? NK : 1;
        float yv[NKK];
        if (NK > 0) {
#pragma unroll
            for (int k = 0; k < NKK; ++k) yv[k] = y[min(tl + 16 * k, w - 1)];
        }
        float mu = 0.f, rstd = 1.f;
        if (a.part && a.stats_ready) {
            mu = a.stats[((long)node * a.g.G + g) * 2];
            rstd = a.stats[((long)node * a.g.G + g) * 2 + 1];
        } else if (a.part) {
            // B <= 32 here (larger batches come with finished statistics): a lane's two partial pairs are read once
            const long pstride = (long)a.n * a.g.G * 2;
            const float* p = a.part + ((long)node * a.g.G + g) * 2;
            float pm[2], pq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long o = (long)min(tl + 16 * u, a.Bs - 1) * pstride;
                pm[u] = p[o];
                pq[u] = p[o + 1];
            }
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) sm += (tl + 16 * u < a.Bs) ? pm[u] : 0.f;
            mu = team_sum(sm) / (float)a.Bs;
            float s2 = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float d = pm[u] - mu;
                s2 += (tl + 16 * u < a.Bs) ? pq[u] + (float)w * d * d : 0.f;
            }
            const float var = team_sum(s2) / ((float)a.Bs * (float)w);
            rstd = 1.0f / sqrtf(var + BN_EPS);
            if (row < a.n && tl == 0) {   // the b == 0 row of this node
                a.stats[((long)node * a.g.G + g) * 2] = mu;
                a.stats[((long)node * a.g.G + g) * 2 + 1] = rstd;
            }
        }
        float* x = a.xout.p[g] + row * a.xout.ld[g];
        if (NK > 0) {
#pragma unroll
            for (int k = 0; k < NKK; ++k) {
                const int c = tl + 16 * k;
                const float v = a.relu ? fmaxf(yv[k], 0.f) : yv[k];
                if (c < w) x[c] = (v - mu) * rstd;
            }
        } else {
            for (int c = tl; c < w; c += 16) {
                const float v = a.relu ? fmaxf(y[c], 0.f) : y[c];
                x[c] = (v - mu) * rstd;
            }
        }
    }
}
void bn_apply_fwd(Seq& q, const float* Y, int ldy, const float* part, float* stats, RowGroups g, GroupPtrs xout,
                  int B, int n, int relu, int Bs) {
    if (!q.ok()) return;
    if (Bs <= 0) Bs = B;
    BnApplyArgs a{Y, ldy, part, stats, g, xout, B, n, relu, 0, Bs};
    if (part && Bs > 32) {
        hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)(((long)n * g.G + 15) / 16)), dim3(256), 0, q.stream, a);
        q.check_launch("bn_finalize");
        a.stats_ready = 1;
    }
    const int maxw = g.G == 2 && g.w[1] > g.w[0] ? g.w[1] : g.w[0];
    const dim3 grid(team_grid((long)B * n * g.G));
    if (maxw <= 32) hipLaunchKernelGGL(k_bn_apply_fwd<2>, grid, dim3(256), 0, q.stream, a);
    else if (maxw <= 64) hipLaunchKernelGGL(k_bn_apply_fwd<4>, grid, dim3(256), 0, q.stream, a);
    else if (maxw <= 128) hipLaunchKernelGGL(k_bn_apply_fwd<8>, grid, dim3(256), 0, q.stream, a);
    else hipLaunchKernelGGL(k_bn_apply_fwd<0>, grid, dim3(256), 0, q.stream, a);
    q.check_launch("bn_apply_fwd");
}

// ------------------------------------------------------------------ bn apply + next layer's transform
// apply_bn of layer l (as k_bn_apply_fwd) AND the next GraphConv's feature transform P = x W (encoders.py:968 with the
// product re-associated, A (x W)) in one launch: the transform is row-local and tiny (20 x 20 .. 64 x 64 per group),
// so as a GEMM launch of its own it was ~6 us of launch floor and LDS staging per layer.  grid (16-row chunks, B):
// a 16-lane team normalises one row (both column groups), the rows meet the group's weights in LDS, and — when the
// aggregation that follows runs on the packed adjacency — the workgroup also writes the exact 3-plane bf16 split of
// its 16 rows of P (two k8 groups; the last chunk writes the zero groups that pad n to a multiple of 32).
struct BnTransformArgs {
    BnApplyArgs bn;
    const float* W[2];       // per group: [w_in, w_out] row-major
    RowGroups gout;          // output column groups of P
    float* P;                // [B, n, ldp]
    int ldp;
    unsigned short* vs;
    int vs_ct, vs_k8;
};
template <int NK>
__global__ __launch_bounds__(256) void k_bn_transform(BnTransformArgs t) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const BnApplyArgs& a = t.bn;
    const int tl = threadIdx.x & 15, team = threadIdx.x >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int G = a.g.G;
    const int cin = a.g.c0[G - 1] + a.g.w[G - 1];
    const int cout = t.gout.c0[G - 1] + t.gout.w[G - 1];
    const int wcnt0 = a.g.w[0] * t.gout.w[0], wcnt1 = G == 2 ? a.g.w[1] * t.gout.w[1] : 0;
    float* WL = lds;                         // both groups' weights
    float* XT = WL + wcnt0 + wcnt1;          // [16][cin]
    float* PT = XT + 16 * cin;               // [16][cout]
    // (1) weights: the first 1024 (all of them at the shapes this kernel serves most) are loaded into registers here and
    // reach LDS only after the row and statistics loads below have been issued: one round trip for all three
    constexpr int WR = 4;
    const int wtot = wcnt0 + wcnt1;
    float wreg[WR];
#pragma unroll
    for (int u = 0; u < WR; ++u) {
        const int e = min(u * 256 + (int)threadIdx.x, wtot - 1);
        wreg[u] = e < wcnt0 ? t.W[0][e] : t.W[1][e - wcnt0];
    }
    // (2) the team's row: BatchNorm of both groups (exactly k_bn_apply_fwd's arithmetic)
    const int node = chunk * 16 + team;
    const bool live = node < a.n;
    const long row = (long)b * a.n + min(node, a.n - 1);
    for (int g = 0; g < G; ++g) {
        const int w = a.g.w[g];
        const float* y = a.Y + row * a.ldy + a.g.c0[g];
        float yv[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) yv[k] = y[min(tl + 16 * k, w - 1)];
        float mu = 0.f, rstd = 1.f;
        const int nd = min(node, a.n - 1);
        if (a.part && a.stats_ready) {
            mu = a.stats[((long)nd * G + g) * 2];
            rstd = a.stats[((long)nd * G + g) * 2 + 1];
        } else if (a.part) {
            const long pstride = (long)a.n * G * 2;
            const float* p = a.part + ((long)nd * G + g) * 2;
            float pm[2], pq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long o = (long)min(tl + 16 * u, a.B - 1) * pstride;
                pm[u] = p[o];
                pq[u] = p[o + 1];
            }
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) sm += (tl + 16 * u < a.B) ? pm[u] : 0.f;
            mu = team_sum(sm) / (float)a.B;
            float s2 = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float d = pm[u] - mu;
                s2 += (tl + 16 * u < a.B) ? pq[u] + (float)w * d * d : 0.f;
            }
            const float var = team_sum(s2) / ((float)a.B * (float)w);
            rstd = 1.0f / sqrtf(var + BN_EPS);
            if (live && b == 0 && tl == 0) {
                a.stats[((long)nd * G + g) * 2] = mu;
                a.stats[((long)nd * G + g) * 2 + 1] = rstd;
            }
        }
        float* x = a.xout.p[g] + row * a.xout.ld[g];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int c = tl + 16 * k;
            const float v = a.relu ? fmaxf(yv[k], 0.f) : yv[k];
            const float xv = (v - mu) * rstd;
            if (c < w) {
                if (live) x[c] = xv;
                XT[team * cin + a.g.c0[g] + c] = live ? xv : 0.f;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < WR; ++u) {
        const int e = u * 256 + (int)threadIdx.x;
        if (e < wtot) WL[e] = wreg[u];
    }
    for (int e = WR * 256 + (int)threadIdx.x; e < wtot; e += 256) WL[e] = e < wcnt0 ? t.W[0][e] : t.W[1][e - wcnt0];
    __syncthreads();
    // (3) P = x W per group: the team keeps its row (broadcast from LDS), a lane takes every 16th output column
    for (int c = tl; c < cout; c += 16) {
        const int g = (G == 2 && c >= t.gout.c0[1]) ? 1 : 0;
        const int win = a.g.w[g], wout = t.gout.w[g];
        const float* xr = XT + team * cin + a.g.c0[g];
        const float* wc = WL + (g ? wcnt0 : 0) + (c - t.gout.c0[g]);
        float acc = 0.f;
#pragma unroll 4
        for (int k = 0; k < win; ++k) acc = fmaf(xr[k], wc[k * wout], acc);
        PT[team * cout + c] = acc;
        if (live) t.P[row * t.ldp + c] = acc;
    }
    if (!t.vs) return;
    __syncthreads();
    unsigned short* vb = t.vs + (long)b * 3 * t.vs_ct * t.vs_k8 * 128;
    const long pl = (long)t.vs_ct * t.vs_k8 * 128;
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const int kend = chunk == (int)gridDim.x - 1 ? t.vs_k8 : 2 * chunk + 2;
    for (int item = threadIdx.x; item < (kend - 2 * chunk) * t.vs_ct * 16; item += 256) {
        const int k8 = 2 * chunk + item / (t.vs_ct * 16), vc = item % (t.vs_ct * 16);
        const int lr = (k8 - 2 * chunk) * 8;
        u16x8 h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (lr < 16 && vc < cout) ? PT[(lr + j) * cout + vc] : 0.f;   // rows past n were computed from 0
            unsigned short hh, mm, ll;
            bf16_split3(v, hh, mm, ll);
            h[j] = hh; m[j] = mm; l[j] = ll;
        }
        const long o = vs_index(0, t.vs_ct, t.vs_k8, vc >> 4, k8, vc & 15, 0);
        *reinterpret_cast<u16x8*>(vb + o) = h;
        *reinterpret_cast<u16x8*>(vb + o + pl) = m;
        *reinterpret_cast<u16x8*>(vb + o + 2 * pl) = l;
    }
}
bool bn_transform_supported(RowGroups gin, RowGroups gout, int B) {
    const int G = gin.G;
    (void)B;                                  // (batches above 32 graphs: k_bn_finalize runs in front, bn_transform_fwd)
    size_t wfl = 0;
    for (int g = 0; g < G; ++g) {
        if (gin.w[g] > 64 || gout.w[g] > 128) return false;
        wfl += (size_t)gin.w[g] * gout.w[g];
    }
    const int cin = gin.c0[G - 1] + gin.w[G - 1], cout = gout.c0[G - 1] + gout.w[G - 1];
    return (wfl + 16 * (size_t)(cin + cout)) * sizeof(float) <= 60 * 1024;
}
// part == null: no BN (relu only).  W[g]: [gin.w[g], gout.w[g]].
void bn_transform_fwd(Seq& q, const float* Y, int ldy, const float* part, float* stats, RowGroups gin, GroupPtrs xout,
                      const float* const W[2], RowGroups gout, float* P, int ldp, int B, int n, int relu,
                      unsigned short* vs) {
    if (!q.ok()) return;
    BnTransformArgs t{};
    t.bn = BnApplyArgs{Y, ldy, part, stats, gin, xout, B, n, relu, 0, B};
    t.W[0] = W[0];
    t.W[1] = W[1];
    t.gout = gout;
    t.P = P;
    t.ldp = ldp;
    t.vs = vs;
    const int G = gin.G;
    const int cin = gin.c0[G - 1] + gin.w[G - 1], cout = gout.c0[G - 1] + gout.w[G - 1];
    if (part && B > 32) {     // big batches: combine the B row partials of a node once, not in each of its B rows
        hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)(((long)n * G + 15) / 16)), dim3(256), 0, q.stream, t.bn);
        q.check_launch("bn_finalize");
        t.bn.stats_ready = 1;
    }
    t.vs_ct = (cout + 15) / 16;
    t.vs_k8 = ((n + 31) / 32) * 4;
    size_t wfl = 0;
    for (int g = 0; g < G; ++g) wfl += (size_t)gin.w[g] * gout.w[g];
    const size_t lds = (wfl + 16 * (size_t)(cin + cout)) * sizeof(float);
    const int maxw = G == 2 && gin.w[1] > gin.w[0] ? gin.w[1] : gin.w[0];
    const dim3 grid((n + 15) / 16, B);
    if (maxw <= 32) hipLaunchKernelGGL(k_bn_transform<2>, grid, dim3(256), lds, q.stream, t);
    else hipLaunchKernelGGL(k_bn_transform<4>, grid, dim3(256), lds, q.stream, t);
    q.check_launch("bn_transform");
}

// ------------------------------------------------------------------ bn bwd partials
// per (row, group): sum_f dx, sum_f dx * xhat
struct BnBwdPartArgs {
    GroupCPtrs dx, xhat;
    RowGroups g;
    float* part;
    long rows;
};
__global__ __launch_bounds__(256) void k_bn_bwd_partials(BnBwdPartArgs a) {
    const int tl = threadIdx.x & 15;
    const long team = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long nteams = (long)gridDim.x * 16;
    const long items = a.rows * a.g.G;
    for (long it = team; it < items; it += nteams) {
        const long row = it / a.g.G;
        const int g = (int)(it % a.g.G);
        const float* dx = a.dx.p[g] + row * a.dx.ld[g];
        const float* xh = a.xhat.p[g] + row * a.xhat.ld[g];
        float s0 = 0.f, s1 = 0.f;
        for (int c = tl; c < a.g.w[g]; c += 16) {
            const float d = dx[c];
            s0 += d;
            s1 += d * xh[c];
        }
        s0 = team_sum(s0);
        s1 = team_sum(s1);
        if (tl == 0) {
            a.part[it * 2] = s0;
            a.part[it * 2 + 1] = s1;
        }
    }
}
void bn_bwd_partials(Seq& q, GroupCPtrs dx, GroupCPtrs xhat, RowGroups g, float* part, long rows) {
    if (!q.ok()) return;
    BnBwdPartArgs a{dx, xhat, g, part, rows};
    hipLaunchKernelGGL(k_bn_bwd_partials, dim3(team_grid(rows * g.G)), dim3(256), 0, q.stream, a);
    q.check_launch("bn_bwd_partials");
}

// ------------------------------------------------------------------ rownorm bwd
// dx -> (BN bwd) -> (ReLU bwd) -> (l2-normalise bwd) -> dU
//   dR = rstd * (dx - m0 - xhat * m1);  dY = dR * (y > 0)
//   dU = inv * (dY - y * <y, dY>)   when ||u|| >= eps, else inv * dY  (clamp_min branch of F.normalize)
struct RownormBwdArgs {
    GroupCPtrs dx, xhat, y;
    const float* invn;
    const float* stats;
    const float* part2;   // [B, n, G, 2] per-row (sum dx, sum dx*xhat)
    RowGroups g;
    float* dU;
    int ldu;
    GroupPtrs dbias;      // per group: this layer's bias-gradient slab (row of graph 0; graphs are dbias.ld apart)
                          // — column sums of dU are ADDED with float atomics (the slabs are zeroed per backward)
    int want_bias;
    unsigned short* vs;   // optional: exact 3-plane bf16 split of dU for the bf16 aggregation (dp_agg.hip layout)
    int vs_ct, vs_k8;
    int B, n;
    int rows_per_chunk;
    int has_relu, has_bn, normalize;
    int Bs;               // graphs in part2 (B, or B x ranks under sync-BN)
    int means_ready;      // big batches: k_bn_bwd_finalize left (m0, m1) of every (node, group) in part2's first block
};
// grid (chunks, B): a workgroup owns a contiguous chunk of rows of ONE graph, so the column sums of dU
// (the bias gradients, db = sum_rows dU) can be accumulated in LDS and leave as one partial per workgroup.
// NK > 0: group widths up to 16 NK, the row's operands live in registers (see the item loop); NK == 0: any width.
template <int NK, bool Q4 = false>
__global__ __launch_bounds__(256) void k_rownorm_bwd(RownormBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float colsum[];
    const int tl = threadIdx.x & 15;
    const int team = threadIdx.x >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int ct = a.g.c0[a.g.G - 1] + a.g.w[a.g.G - 1];
    // each 16-lane team owns one row of the LDS accumulator (no atomics: lane c owns columns c, c+16, ...)
    float* mysum = colsum + team * ct;
    float* tile8 = colsum + (a.want_bias ? 16 * ct : 0);     // [8 rows][ct] copy of dU for the bf16 split
    if (a.want_bias || a.vs) {
        const int tot = (a.want_bias ? 16 * ct : 0) + (a.vs ? 8 * ct : 0);
        for (int c = threadIdx.x; c < tot; c += 256) colsum[c] = 0.f;
        __syncthreads();
    }
    const int r0 = chunk * a.rows_per_chunk;
    const int r1 = min(a.n, r0 + a.rows_per_chunk);
    const int items = (r1 - r0) * a.g.G;
    for (int it = team; it < items; it += 16) {
        const int node = r0 + it / a.g.G;
        const int g = it % a.g.G;
        const long row = (long)b * a.n + node;
        const int w = a.g.w[g];
        const float* dx = a.dx.p[g] + row * a.dx.ld[g];
        const float* y = a.y.p[g] + row * a.y.ld[g];
        if constexpr (Q4) {
            // the NK > 0 form below with 16-byte quads (NK = quads per lane)
            constexpr int NQ = NK > 0 ? NK : 1;
            const int nq = w >> 2;
            const float* xh = a.has_bn ? a.xhat.p[g] + row * a.xhat.ld[g] : dx;
            f4u dxv[NQ], yv[NQ], xhv[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int qd = min(tl + 16 * k, nq - 1) * 4;
                dxv[k] = *reinterpret_cast<const f4u*>(dx + qd);
                yv[k] = *reinterpret_cast<const f4u*>(y + qd);
                xhv[k] = *reinterpret_cast<const f4u*>(xh + qd);
            }
            const float rstd_l = (a.has_bn ? a.stats : dx)[a.has_bn ? ((long)node * a.g.G + g) * 2 + 1 : 0];
            const float inv_l = (a.normalize ? a.invn : dx)[a.normalize ? row * a.g.G + g : 0];
            float s0 = 0.f, s1 = 0.f;
            {
                const long pstride = (long)a.n * a.g.G * 2;
                const float* p = (a.has_bn ? a.part2 : dx) + (a.has_bn ? ((long)node * a.g.G + g) * 2 : 0);
                const int nb = (a.has_bn && !a.means_ready) ? a.Bs : 0;
                for (int bb = tl; bb < nb; bb += 16) {
                    s0 += p[bb * pstride];
                    s1 += p[bb * pstride + 1];
                }
                if (a.means_ready) {
                    s0 = p[0];
                    s1 = p[1];
                }
            }
            const float cnt = (float)a.Bs * (float)w;
            const float rstd = a.has_bn ? rstd_l : 1.f;
            const float m0 = !a.has_bn ? 0.f : a.means_ready ? s0 : team_sum(s0) / cnt;
            const float m1 = !a.has_bn ? 0.f : a.means_ready ? s1 : team_sum(s1) / cnt;
            const float inv = a.normalize ? inv_l : 1.f;
            const bool project = a.normalize && (inv < 1.0f / L2_EPS);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < NQ; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float d = dxv[k][j];
                    if (a.has_bn) d = rstd * (d - m0 - xhv[k][j] * m1);
                    if (a.has_relu) d = yv[k][j] > 0.f ? d : 0.f;
                    if (tl + 16 * k >= nq) d = 0.f;
                    dxv[k][j] = d;
                    dot += d * yv[k][j];
                }
            dot = team_sum(dot);
            float* du = a.dU + row * a.ldu + a.g.c0[g];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int c = (tl + 16 * k) * 4;
                if (tl + 16 * k < nq) {
                    f4u v;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] = project ? inv * (dxv[k][j] - yv[k][j] * dot) : inv * dxv[k][j];
                    *reinterpret_cast<f4u*>(du + c) = v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (a.want_bias) mysum[a.g.c0[g] + c + j] += v[j];
                        if (a.vs) tile8[(node - r0) * ct + a.g.c0[g] + c + j] = v[j];
                    }
                }
            }
            continue;
        }
        if (NK > 0) {
            // ---- every global read of the item goes out here, clamped and unpredicated: the row is one memory
            // round trip instead of five dependent ones (statistics -> row -> row again)
            const float* xh = a.has_bn ? a.xhat.p[g] + row * a.xhat.ld[g] : dx;
            float dxv[NK > 0 ? NK : 1], yv[NK > 0 ? NK : 1], xhv[NK > 0 ? NK : 1];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int cc = min(tl + 16 * k, w - 1);
                dxv[k] = dx[cc];
                yv[k] = y[cc];
                xhv[k] = xh[cc];
            }
            const float rstd_l = (a.has_bn ? a.stats : dx)[a.has_bn ? ((long)node * a.g.G + g) * 2 + 1 : 0];
            const float inv_l = (a.normalize ? a.invn : dx)[a.normalize ? row * a.g.G + g : 0];
            float s0 = 0.f, s1 = 0.f;
            {
                const long pstride = (long)a.n * a.g.G * 2;
                const float* p = (a.has_bn ? a.part2 : dx) + (a.has_bn ? ((long)node * a.g.G + g) * 2 : 0);
                const int nb = (a.has_bn && !a.means_ready) ? a.Bs : 0;
                for (int bb = tl; bb < nb; bb += 16) {
                    s0 += p[bb * pstride];
                    s1 += p[bb * pstride + 1];
                }
                if (a.means_ready) {          // (m0, m1) already combined: every lane reads the same pair
                    s0 = p[0];
                    s1 = p[1];
                }
            }
            // ---- arithmetic
            const float cnt = (float)a.Bs * (float)w;
            const float rstd = a.has_bn ? rstd_l : 1.f;
            const float m0 = !a.has_bn ? 0.f : a.means_ready ? s0 : team_sum(s0) / cnt;
            const float m1 = !a.has_bn ? 0.f : a.means_ready ? s1 : team_sum(s1) / cnt;
            const float inv = a.normalize ? inv_l : 1.f;
            const bool project = a.normalize && (inv < 1.0f / L2_EPS);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                float d = dxv[k];
                if (a.has_bn) d = rstd * (d - m0 - xhv[k] * m1);
                if (a.has_relu) d = yv[k] > 0.f ? d : 0.f;
                if (tl + 16 * k >= w) d = 0.f;
                dxv[k] = d;
                dot += d * yv[k];
            }
            dot = team_sum(dot);
            float* du = a.dU + row * a.ldu + a.g.c0[g];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int c = tl + 16 * k;
                if (c < w) {
                    const float v = project ? inv * (dxv[k] - yv[k] * dot) : inv * dxv[k];
                    du[c] = v;
                    if (a.want_bias) mysum[a.g.c0[g] + c] += v;
                    if (a.vs) tile8[(node - r0) * ct + a.g.c0[g] + c] = v;
                }
            }
            continue;
        }
        const float* xh = a.has_bn ? a.xhat.p[g] + row * a.xhat.ld[g] : nullptr;
        float rstd = 1.f, m0 = 0.f, m1 = 0.f;
        if (a.has_bn) {
            rstd = a.stats[((long)node * a.g.G + g) * 2 + 1];
            const long pstride = (long)a.n * a.g.G * 2;
            const float* p = a.part2 + ((long)node * a.g.G + g) * 2;
            if (a.means_ready) {
                m0 = p[0];
                m1 = p[1];
            } else {
                float s0 = 0.f, s1 = 0.f;
                for (int bb = tl; bb < a.Bs; bb += 16) {
                    s0 += p[bb * pstride];
                    s1 += p[bb * pstride + 1];
                }
                const float cnt = (float)a.Bs * (float)w;
                m0 = team_sum(s0) / cnt;
                m1 = team_sum(s1) / cnt;
            }
        }
        const float inv = a.normalize ? a.invn[row * a.g.G + g] : 1.f;
        const bool project = a.normalize && (inv < 1.0f / L2_EPS);
        float dot = 0.f;
        for (int c = tl; c < w; c += 16) {
            float d = dx[c];
            const float yy = y[c];
            if (a.has_bn) d = rstd * (d - m0 - xh[c] * m1);
            if (a.has_relu) d = yy > 0.f ? d : 0.f;
            dot += d * yy;
        }
        dot = team_sum(dot);
        float* du = a.dU + row * a.ldu + a.g.c0[g];
        for (int c = tl; c < w; c += 16) {
            float d = dx[c];
            const float yy = y[c];
            if (a.has_bn) d = rstd * (d - m0 - xh[c] * m1);
            if (a.has_relu) d = yy > 0.f ? d : 0.f;
            const float v = project ? inv * (d - yy * dot) : inv * d;
            du[c] = v;
            if (a.want_bias) mysum[a.g.c0[g] + c] += v;
            if (a.vs) tile8[(node - r0) * ct + a.g.c0[g] + c] = v;
        }
    }
    if (a.want_bias || a.vs) __syncthreads();
    if (a.vs) {
        // the 8 rows of this workgroup are exactly one k8 group of the split operand
        unsigned short* vb = a.vs + (long)b * 3 * a.vs_ct * a.vs_k8 * 128;
        const long pl = (long)a.vs_ct * a.vs_k8 * 128;
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        const int last = gridDim.x - 1;
        const int kend = chunk == last ? a.vs_k8 : chunk + 1;     // the last workgroup also zeroes the padded groups
        for (int k8 = chunk; k8 < kend; ++k8)
            for (int vc = threadIdx.x; vc < a.vs_ct * 16; vc += 256) {
                u16x8 h, m, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = (k8 == chunk && vc < ct) ? tile8[j * ct + vc] : 0.f;
                    unsigned short hh, mm, ll;
                    bf16_split3(v, hh, mm, ll);
                    h[j] = hh; m[j] = mm; l[j] = ll;
                }
                const long o = vs_index(0, a.vs_ct, a.vs_k8, vc >> 4, k8, vc & 15, 0);
                *reinterpret_cast<u16x8*>(vb + o) = h;
                *reinterpret_cast<u16x8*>(vb + o + pl) = m;
                *reinterpret_cast<u16x8*>(vb + o + 2 * pl) = l;
            }
    }
    if (a.want_bias) {
        for (int c = threadIdx.x; c < ct; c += 256) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += colsum[k * ct + c];
            const int g = (a.g.G == 2 && c >= a.g.c0[1]) ? 1 : 0;
            float* dst = a.dbias.p[g];
            if (dst) atomicAdd(dst + (long)b * a.dbias.ld[g] + (c - a.g.c0[g]), t);
        }
    }
}
// ------------------------------------------------------------------ rownorm bwd + d(A x) = dV W^T
// The backward twin of k_widen_fwd for layers run as (A x) W: the item computes dV exactly as k_rownorm_bwd's quad
// form does (and writes it: dW = (A x)^T dV still needs it) and, with the group's weights staged in LDS, also the
// row-local product d(A x)[j] = sum_c dV[c] W[j][c] — din 16-lane sums per row — so the [n x din] x [din x 256]^T GEMM
// launch (135 us at the ER shape) goes.  64 rows per workgroup (the weights are staged once per workgroup), items
// group by group, narrow groups in the one-quad form.
struct RownormBwdMvArgs {
    RownormBwdArgs r;
    const float* W[2];       // [din_g][w_g] row-major
    int din[2], c0in[2];
    float* dUin;             // [B * n, lddu]: group g's columns at c0in[g]
    int lddu;
};
template <int NQ>
__device__ __forceinline__ void rnb_mv_item(const RownormBwdMvArgs& m, const float* Wg, float* mysum, int b, int node,
                                            int g, int tl) {
    const RownormBwdArgs& a = m.r;
    const long row = (long)b * a.n + node;
    const int w = a.g.w[g], nq = w >> 2;
    const float* dx = a.dx.p[g] + row * a.dx.ld[g];
    const float* y = a.y.p[g] + row * a.y.ld[g];
    const float* xh = a.has_bn ? a.xhat.p[g] + row * a.xhat.ld[g] : dx;
    f4u dxv[NQ], yv[NQ], xhv[NQ];
    int qoff[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        qoff[k] = min(tl + 16 * k, nq - 1) * 4;
        dxv[k] = *reinterpret_cast<const f4u*>(dx + qoff[k]);
        yv[k] = *reinterpret_cast<const f4u*>(y + qoff[k]);
        xhv[k] = *reinterpret_cast<const f4u*>(xh + qoff[k]);
    }
    const float rstd_l = (a.has_bn ? a.stats : dx)[a.has_bn ? ((long)node * a.g.G + g) * 2 + 1 : 0];
    const float inv_l = (a.normalize ? a.invn : dx)[a.normalize ? row * a.g.G + g : 0];
    float s0 = 0.f, s1 = 0.f;
    {
        const long pstride = (long)a.n * a.g.G * 2;
        const float* p = (a.has_bn ? a.part2 : dx) + (a.has_bn ? ((long)node * a.g.G + g) * 2 : 0);
        const int nb = (a.has_bn && !a.means_ready) ? a.Bs : 0;
        for (int bb = tl; bb < nb; bb += 16) {
            s0 += p[bb * pstride];
            s1 += p[bb * pstride + 1];
        }
        if (a.means_ready) {
            s0 = p[0];
            s1 = p[1];
        }
    }
    const float cnt = (float)a.Bs * (float)w;
    const float rstd = a.has_bn ? rstd_l : 1.f;
    const float m0 = !a.has_bn ? 0.f : a.means_ready ? s0 : team_sum(s0) / cnt;
    const float m1 = !a.has_bn ? 0.f : a.means_ready ? s1 : team_sum(s1) / cnt;
    const float inv = a.normalize ? inv_l : 1.f;
    const bool project = a.normalize && (inv < 1.0f / L2_EPS);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = dxv[k][j];
            if (a.has_bn) d = rstd * (d - m0 - xhv[k][j] * m1);
            if (a.has_relu) d = yv[k][j] > 0.f ? d : 0.f;
            if (tl + 16 * k >= nq) d = 0.f;
            dxv[k][j] = d;
            dot += d * yv[k][j];
        }
    dot = team_sum(dot);
    float* du = a.dU + row * a.ldu + a.g.c0[g];
    f4a dv[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const bool on = tl + 16 * k < nq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = project ? inv * (dxv[k][j] - yv[k][j] * dot) : inv * dxv[k][j];
            dv[k][j] = on ? t : 0.f;
        }
        if (on) {
            *reinterpret_cast<f4u*>(du + qoff[k]) = (f4u)dv[k];
            if (a.want_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mysum[a.g.c0[g] + qoff[k] + j] += dv[k][j];
            }
        }
    }
    // d(A x)[j] = <dV, W[j, :]>: lane j & 15 keeps sum j
    const int din = m.din[g];
    float keep0 = 0.f, keep1 = 0.f;
#pragma unroll 2
    for (int j = 0; j < din; ++j) {
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const f4a w4 = *reinterpret_cast<const f4a*>(Wg + j * w + qoff[k]);
            p += dv[k][0] * w4[0] + dv[k][1] * w4[1] + dv[k][2] * w4[2] + dv[k][3] * w4[3];
        }
        p = team_sum(p);
        if (tl == (j & 15)) {
            if (j < 16) keep0 = p;
            else keep1 = p;
        }
    }
    float* dui = m.dUin + row * m.lddu + m.c0in[g];
    if (tl < din) dui[tl] = keep0;
    if (tl + 16 < din) dui[tl + 16] = keep1;
}
template <int NQ>
__global__ __launch_bounds__(256) void k_rownorm_bwd_mv(RownormBwdMvArgs m) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const RownormBwdArgs& a = m.r;
    const int tl = threadIdx.x & 15, team = threadIdx.x >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int G = a.g.G;
    const int ct = a.g.c0[G - 1] + a.g.w[G - 1];
    float* colsum = sm;                                       // [16][ct] when bias gradients are wanted
    float* wl = sm + (a.want_bias ? 16 * ct : 0);             // both groups' weights
    const int wcnt0 = m.din[0] * a.g.w[0], wcnt1 = G == 2 ? m.din[1] * a.g.w[1] : 0;
    for (int e = threadIdx.x * 4; e < wcnt0 + wcnt1; e += 1024)
        *reinterpret_cast<f4a*>(wl + e) =
            *reinterpret_cast<const f4u*>(e < wcnt0 ? m.W[0] + e : m.W[1] + (e - wcnt0));
    if (a.want_bias)
        for (int c = threadIdx.x; c < 16 * ct; c += 256) colsum[c] = 0.f;
    __syncthreads();
    float* mysum = colsum + team * ct;
    const int r0 = chunk * a.rows_per_chunk;
    const int nrows = min(a.n, r0 + a.rows_per_chunk) - r0;
    for (int it = team; it < nrows * G; it += 16) {
        const int g = it >= nrows ? 1 : 0;
        const int node = r0 + (g ? it - nrows : it);
        const float* Wg = wl + (g ? wcnt0 : 0);
        if (a.g.w[g] <= 64) rnb_mv_item<1>(m, Wg, mysum, b, node, g, tl);
        else rnb_mv_item<NQ>(m, Wg, mysum, b, node, g, tl);
    }
    if (!a.want_bias) return;
    __syncthreads();
    for (int c = threadIdx.x; c < ct; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += colsum[k * ct + c];
        const int g = (G == 2 && c >= a.g.c0[1]) ? 1 : 0;
        float* dst = a.dbias.p[g];
        if (dst) atomicAdd(dst + (long)b * a.dbias.ld[g] + (c - a.g.c0[g]), t);
    }
}

// Big batches: the B x ranks partial pairs of a (node, group) are combined ONCE, in place (the result overwrites the
// pair of graph 0), instead of in every one of the B rows of that node — at B = 256 the on-the-fly combine was 256
// scattered 8-byte loads per row item: 325 us per launch against ~70 (the forward pass has k_bn_finalize for the same
// reason).
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(float* part2, int Bs, int n, RowGroups g) {
    const int tl = threadIdx.x & 15;
    const long it = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (it >= (long)n * g.G) return;
    const int gi = (int)(it % g.G);
    const long pstride = (long)n * g.G * 2;
    float* p = part2 + it * 2;
    float s0 = 0.f, s1 = 0.f;
    for (int b = tl; b < Bs; b += 16) {
        s0 += p[b * pstride];
        s1 += p[b * pstride + 1];
    }
    s0 = team_sum(s0);
    s1 = team_sum(s1);
    const float cnt = (float)Bs * (float)g.w[gi];
    if (tl == 0) {
        p[0] = s0 / cnt;
        p[1] = s1 / cnt;
    }
}
int rownorm_bwd_chunks(int n) { return (n + 7) / 8; }
void rownorm_bwd(Seq& q, GroupCPtrs dx, GroupCPtrs xhat, GroupCPtrs y, const float* invn, const float* stats,
                 const float* part2, RowGroups g, float* dU, int ldu, const GroupPtrs* dbias, int B, int n,
                 int has_relu, int has_bn, int normalize, unsigned short* vs, int Bs) {
    if (!q.ok()) return;
    if (Bs <= 0) Bs = B;
    GroupPtrs db{};
    int want = 0;
    if (dbias) {
        db = *dbias;
        want = (db.p[0] || db.p[1]) ? 1 : 0;
    }
    const int ct = g.c0[g.G - 1] + g.w[g.G - 1];
    RownormBwdArgs a{dx, xhat, y, invn, stats, part2, g, dU, ldu, db, want, vs, (ct + 15) / 16, ((n + 31) / 32) * 4,
                     B, n, 8, has_relu, has_bn, normalize, Bs, 0};
    if (has_bn && Bs > 32) {
        // (part2 is plan scratch; the combined means go where graph 0's pair was)
        hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((unsigned)(((long)n * g.G + 15) / 16)), dim3(256), 0, q.stream,
                           const_cast<float*>(part2), Bs, n, g);
        q.check_launch("bn_bwd_finalize");
        a.means_ready = 1;
    }
    const int maxw = g.G == 2 && g.w[1] > g.w[0] ? g.w[1] : g.w[0];
    const dim3 grid(rownorm_bwd_chunks(n), B);
    const size_t lds = ((want ? 16 : 0) + (vs ? 8 : 0)) * ct * sizeof(float);
    if (maxw > 128 && maxw <= 512 && row_quads_ok(g)) {
        if (maxw <= 256) hipLaunchKernelGGL((k_rownorm_bwd<4, true>), grid, dim3(256), lds, q.stream, a);
        else if (maxw <= 320) hipLaunchKernelGGL((k_rownorm_bwd<5, true>), grid, dim3(256), lds, q.stream, a);
        else hipLaunchKernelGGL((k_rownorm_bwd<8, true>), grid, dim3(256), lds, q.stream, a);
        q.check_launch("rownorm_bwd");
        return;
    }
    if (maxw <= 32) hipLaunchKernelGGL(k_rownorm_bwd<2>, grid, dim3(256), lds, q.stream, a);
    else if (maxw <= 64) hipLaunchKernelGGL(k_rownorm_bwd<4>, grid, dim3(256), lds, q.stream, a);
    else if (maxw <= 128) hipLaunchKernelGGL(k_rownorm_bwd<8>, grid, dim3(256), lds, q.stream, a);
    else if (maxw <= 256) hipLaunchKernelGGL(k_rownorm_bwd<16>, grid, dim3(256), lds, q.stream, a);
    else hipLaunchKernelGGL(k_rownorm_bwd<0>, grid, dim3(256), lds, q.stream, a);
    q.check_launch("rownorm_bwd");
}

bool rownorm_bwd_mv_supported(RowGroups g, const int din[2], int n) {
    if (knobs().no_widen_fusion || !row_quads_ok(g)) return false;
    const int ct = g.c0[g.G - 1] + g.w[g.G - 1];
    size_t fl = (size_t)16 * ct;
    int maxw = 0;
    for (int i = 0; i < g.G; ++i) {
        if (din[i] < 1 || din[i] > 32) return false;
        fl += (size_t)din[i] * g.w[i];
        maxw = g.w[i] > maxw ? g.w[i] : maxw;
    }
    return n >= 1 && maxw > 128 && maxw <= 320 && fl * sizeof(float) <= 60 * 1024;
}
// rownorm_bwd (no bf16 split output) + dUin[:, c0in[g] + j] = sum_c dU_g[:, c] W_g[j, c]
void rownorm_bwd_mv(Seq& q, GroupCPtrs dx, GroupCPtrs xhat, GroupCPtrs y, const float* invn, const float* stats,
                    const float* part2, RowGroups g, float* dU, int ldu, const GroupPtrs* dbias, int B, int n,
                    int has_relu, int has_bn, int normalize, int Bs, const float* const W[2], const int din[2],
                    const int c0in[2], float* dUin, int lddu) {
    if (!q.ok()) return;
    if (Bs <= 0) Bs = B;
    GroupPtrs db{};
    int want = 0;
    if (dbias) {
        db = *dbias;
        want = (db.p[0] || db.p[1]) ? 1 : 0;
    }
    const int ct = g.c0[g.G - 1] + g.w[g.G - 1];
    RownormBwdMvArgs m{};
    m.r = RownormBwdArgs{dx, xhat, y, invn, stats, part2, g, dU, ldu, db, want, nullptr, 0, 0,
                         B, n, 64, has_relu, has_bn, normalize, Bs, 0};
    if (has_bn && Bs > 32) {
        hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((unsigned)(((long)n * g.G + 15) / 16)), dim3(256), 0, q.stream,
                           const_cast<float*>(part2), Bs, n, g);
        q.check_launch("bn_bwd_finalize");
        m.r.means_ready = 1;
    }
    size_t fl = want ? (size_t)16 * ct : 0;
    int maxw = 0;
    for (int i = 0; i < 2; ++i) {
        m.W[i] = W[i];
        m.din[i] = i < g.G ? din[i] : 0;
        m.c0in[i] = c0in[i];
        if (i < g.G) {
            fl += (size_t)din[i] * g.w[i];
            maxw = g.w[i] > maxw ? g.w[i] : maxw;
        }
    }
    m.dUin = dUin;
    m.lddu = lddu;
    const dim3 grid((n + 63) / 64, B);
    if (maxw <= 256) hipLaunchKernelGGL(k_rownorm_bwd_mv<4>, grid, dim3(256), fl * sizeof(float), q.stream, m);
    else hipLaunchKernelGGL(k_rownorm_bwd_mv<5>, grid, dim3(256), fl * sizeof(float), q.stream, m);
    q.check_launch("rownorm_bwd_mv");
}

// ------------------------------------------------------------------ column sums
// out[b, c] = sum_r X[b, r, c]    (bias gradients; deterministic)
__global__ __launch_bounds__(1024) void k_colsum_batched(const float* X, int ldx, long strideX, int rows, int cols,
                                                         float* out, long strideOut, int atomic) {
    __shared__ float red[16][64];
    const int b = blockIdx.y;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    // blockIdx.z splits the rows (atomic mode only)
    const int per = (rows + gridDim.z - 1) / gridDim.z;
    const int rbeg = blockIdx.z * per, rend = min(rows, rbeg + per);
    float s = 0.f;
    if (c < cols) {
        const float* x = X + (long)b * strideX + c;
        for (int r = rbeg + rl; r < rend; r += 16) s += x[(long)r * ldx];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        if (atomic) atomicAdd(out + (long)b * strideOut + c, t);
        else out[(long)b * strideOut + c] = t;
    }
}
// rowsplit > 1: the rows are cut into `rowsplit` ranges that ADD into `out` with float atomics (out pre-zeroed)
void colsum_batched(Seq& q, const float* X, int ldx, long strideX, int rows, int cols, float* out, long strideOut,
                    int batch, int rowsplit) {
    if (!q.ok() || cols <= 0 || batch <= 0) return;
    if (rowsplit < 1) rowsplit = 1;
    hipLaunchKernelGGL(k_colsum_batched, dim3((cols + 63) / 64, batch, rowsplit), dim3(1024), 0, q.stream, X, ldx,
                       strideX, rows, cols, out, strideOut, rowsplit > 1 ? 1 : 0);
    q.check_launch("colsum_batched");
}

// ------------------------------------------------------------------ softmax * mask
// S = softmax_K(logits) for n < num_nodes[b], 0 otherwise  (encoders.py:1273-1275)
__global__ __launch_bounds__(256) void k_softmax_mask_fwd(const float* logits, int ldl, float* S, int lds,
                                                          float* S2, const int* num_nodes, long rows, int n, int K) {
    const int tl = threadIdx.x & 15;
    const long team = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long nteams = (long)gridDim.x * 16;
    for (long row = team; row < rows; row += nteams) {
        const int b = (int)(row / n), node = (int)(row % n);
        const bool valid = num_nodes ? node < num_nodes[b] : true;
        const float* l = logits + row * ldl;
        float* s = S + row * lds;
        float* s2 = S2 ? S2 + row * lds : nullptr;   // second copy (the caller-visible assign_tensor)
        if (!valid) {
            for (int c = tl; c < K; c += 16) {
                s[c] = 0.f;
                if (s2) s2[c] = 0.f;
            }
            continue;
        }
        float m = -INFINITY;
        for (int c = tl; c < K; c += 16) m = fmaxf(m, l[c]);
        m = team_max(m);
        float sum = 0.f;
        for (int c = tl; c < K; c += 16) sum += expf(l[c] - m);
        sum = team_sum(sum);
        const float r = 1.f / sum;
        for (int c = tl; c < K; c += 16) {
            const float v = expf(l[c] - m) * r;
            s[c] = v;
            if (s2) s2[c] = v;
        }
    }
}
// Encoder-plan variant: grid (16-row chunks, B).  Besides S (and the caller-visible copy S2) the workgroup writes the
// exact 3-plane bf16 split of its 16 rows of S in the layout the packed aggregation reads (two k8 groups; the last
// chunk also writes the zero k8 groups that pad n to a multiple of 32), and the grid zero-fills `zero_p`
// (the split-K accumulators X', A' of the pooling products) — two small launches folded into this one.
struct SoftmaxFwdArgs {
    const float* logits;
    int ldl;
    float* S;
    int lds;
    float* S2;
    const int* num_nodes;
    int n, K;
    unsigned short* vs;
    int vs_ct, vs_k8;
    uint4* zero_p;
    long zero_n16;
};
// NK > 0: K <= 16 NK, the row's logits are read once and its exponentials computed once, both kept in registers.
template <int NK, bool Q4 = false>
__global__ __launch_bounds__(256) void k_softmax_mask_fwd_plan(SoftmaxFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [16][K]
    const int tl = threadIdx.x & 15, team = threadIdx.x >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x;
    if (a.zero_p) {
        const long wg = (long)b * gridDim.x + chunk, nwg = (long)gridDim.x * gridDim.y;
        for (long i = wg * 256 + threadIdx.x; i < a.zero_n16; i += nwg * 256) a.zero_p[i] = make_uint4(0, 0, 0, 0);
    }
    const int node = chunk * 16 + team;
    const int K = a.K;
    if (node < a.n) {
        const long row = (long)b * a.n + node;
        const bool valid = a.num_nodes ? node < a.num_nodes[b] : true;
        const float* l = a.logits + row * a.ldl;
        float* s = a.S + row * a.lds;
        float* s2 = a.S2 ? a.S2 + row * a.lds : nullptr;
        if constexpr (Q4) {
            constexpr int NQ = NK > 0 ? NK : 1;       // quads per lane, K % 4 == 0
            const int nq = K >> 2;
            f4u lv[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) lv[k] = *reinterpret_cast<const f4u*>(l + min(tl + 16 * k, nq - 1) * 4);
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < NQ; ++k) m = fmaxf(fmaxf(m, fmaxf(lv[k][0], lv[k][1])), fmaxf(lv[k][2], lv[k][3]));
            m = team_max(m);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < NQ; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lv[k][j] = expf(lv[k][j] - m);
                    sum += (tl + 16 * k < nq) ? lv[k][j] : 0.f;
                }
            sum = team_sum(sum);
            const float r = 1.f / sum;
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int c = (tl + 16 * k) * 4;
                if (tl + 16 * k < nq) {
                    f4u v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = valid ? lv[k][j] * r : 0.f;
                    *reinterpret_cast<f4u*>(s + c) = v;
                    if (s2) *reinterpret_cast<f4u*>(s2 + c) = v;
                    *reinterpret_cast<f4u*>(tile + team * K + c) = v;
                }
            }
        } else if (NK > 0) {
            constexpr int NKK = NK > 0 ? NK : 1;
            float lv[NKK];
#pragma unroll
            for (int k = 0; k < NKK; ++k) lv[k] = l[min(tl + 16 * k, K - 1)];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < NKK; ++k) m = fmaxf(m, lv[k]);        // clamped duplicates do not change a max
            m = team_max(m);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < NKK; ++k) {
                lv[k] = expf(lv[k] - m);
                sum += (tl + 16 * k < K) ? lv[k] : 0.f;
            }
            sum = team_sum(sum);
            const float r = 1.f / sum;
#pragma unroll
            for (int k = 0; k < NKK; ++k) {
                const int c = tl + 16 * k;
                if (c < K) {
                    const float v = valid ? lv[k] * r : 0.f;
                    s[c] = v;
                    if (s2) s2[c] = v;
                    tile[team * K + c] = v;
                }
            }
        } else {
            float m = -INFINITY;
            for (int c = tl; c < K; c += 16) m = fmaxf(m, l[c]);
            m = team_max(m);
            float sum = 0.f;
            for (int c = tl; c < K; c += 16) sum += expf(l[c] - m);
            sum = team_sum(sum);
            const float r = 1.f / sum;
            for (int c = tl; c < K; c += 16) {
                const float v = valid ? expf(l[c] - m) * r : 0.f;
                s[c] = v;
                if (s2) s2[c] = v;
                tile[team * K + c] = v;
            }
        }
    } else {
        for (int c = tl; c < K; c += 16) tile[team * K + c] = 0.f;
    }
    if (!a.vs) return;
    __syncthreads();
    unsigned short* vb = a.vs + (long)b * 3 * a.vs_ct * a.vs_k8 * 128;
    const long pl = (long)a.vs_ct * a.vs_k8 * 128;
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const int kend = chunk == (int)gridDim.x - 1 ? a.vs_k8 : 2 * chunk + 2;   // last chunk: zero groups up to k8 pad
    for (int item = threadIdx.x; item < (kend - 2 * chunk) * a.vs_ct * 16; item += 256) {
        const int k8 = 2 * chunk + item / (a.vs_ct * 16), vc = item % (a.vs_ct * 16);
        const int lr = (k8 - 2 * chunk) * 8;                     // first local row of this k8 group (>= 16: padding)
        u16x8 h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (lr < 16 && vc < K) ? tile[(lr + j) * K + vc] : 0.f;
            unsigned short hh, mm, ll;
            bf16_split3(v, hh, mm, ll);
            h[j] = hh; m[j] = mm; l[j] = ll;
        }
        const long o = vs_index(0, a.vs_ct, a.vs_k8, vc >> 4, k8, vc & 15, 0);
        *reinterpret_cast<u16x8*>(vb + o) = h;
        *reinterpret_cast<u16x8*>(vb + o + pl) = m;
        *reinterpret_cast<u16x8*>(vb + o + 2 * pl) = l;
    }
}
void softmax_mask_fwd(Seq& q, const float* logits, int ldl, float* S, int lds, const int* num_nodes, int B, int n,
                      int K, float* S2, unsigned short* vs, void* zero_p, size_t zero_bytes) {
    if (!q.ok()) return;
    const long rows = (long)B * n;
    const bool fits = (size_t)16 * K * sizeof(float) <= 48 * 1024;   // (a split is only ever asked for K <= 320)
    if (!fits && zero_p) {
        zero_fill(q, zero_p, zero_bytes);
        zero_p = nullptr;
    }
    if (fits && (vs || zero_p)) {
        if (zero_p && ((reinterpret_cast<uintptr_t>(zero_p) & 15) != 0 || (zero_bytes & 15) != 0)) {
            zero_small(q, zero_p, zero_bytes);
            zero_p = nullptr;
        }
        SoftmaxFwdArgs a{logits, ldl, S, lds, S2, num_nodes, n, K, vs, (K + 15) / 16, ((n + 31) / 32) * 4,
                         (uint4*)zero_p, (long)(zero_bytes / 16)};
        const dim3 grid((n + 15) / 16, B);
        const size_t sm = (size_t)16 * K * sizeof(float);
        if (K > 128 && K <= 512 && (K & 3) == 0 && !knobs().no_row_quads) {
            if (K <= 256) hipLaunchKernelGGL((k_softmax_mask_fwd_plan<4, true>), grid, dim3(256), sm, q.stream, a);
            else if (K <= 320) hipLaunchKernelGGL((k_softmax_mask_fwd_plan<5, true>), grid, dim3(256), sm, q.stream, a);
            else hipLaunchKernelGGL((k_softmax_mask_fwd_plan<8, true>), grid, dim3(256), sm, q.stream, a);
        } else
        if (K <= 64) hipLaunchKernelGGL(k_softmax_mask_fwd_plan<4>, grid, dim3(256), sm, q.stream, a);
        else if (K <= 128) hipLaunchKernelGGL(k_softmax_mask_fwd_plan<8>, grid, dim3(256), sm, q.stream, a);
        else if (K <= 256) hipLaunchKernelGGL(k_softmax_mask_fwd_plan<16>, grid, dim3(256), sm, q.stream, a);
        else hipLaunchKernelGGL(k_softmax_mask_fwd_plan<0>, grid, dim3(256), sm, q.stream, a);
        q.check_launch("softmax_mask_fwd_plan");
        return;
    }
    hipLaunchKernelGGL(k_softmax_mask_fwd, dim3(team_grid(rows)), dim3(256), 0, q.stream, logits, ldl, S, lds, S2,
                       num_nodes, rows, n, K);
    q.check_launch("softmax_mask_fwd");
}

// dlogits = S * (dS - <dS, S>)   (rows with n >= num_nodes: S = 0 -> dlogits = 0)
__global__ __launch_bounds__(256) void k_softmax_mask_bwd(const float* S, int lds, const float* dS, int ldds,
                                                          const int* num_nodes, float* dl, int ldl, long rows, int n,
                                                          int K) {
    const int tl = threadIdx.x & 15;
    const long team = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long nteams = (long)gridDim.x * 16;
    for (long row = team; row < rows; row += nteams) {
        const float* s = S + row * lds;
        const float* d = dS + row * ldds;
        float dot = 0.f;
        for (int c = tl; c < K; c += 16) dot += s[c] * d[c];
        dot = team_sum(dot);
        float* o = dl + row * ldl;
        for (int c = tl; c < K; c += 16) o[c] = s[c] * (d[c] - dot);
    }
}
// Encoder-plan variant: grid (64-row chunks, B); also adds the column sums of dlogits (the assign_pred bias
// gradient) of its rows to the graph's slab row with one float atomic per column and workgroup.
// NK > 0: K <= 16 NK and RW of a team's four rows have every operand in flight at once and kept in registers (one
// memory round trip per RW rows; row after row with two passes each it was eight).  NK == 0: any K.
template <int NK, int RW, bool Q4 = false>
__global__ __launch_bounds__(256) void k_softmax_mask_bwd_plan(const float* S, int lds, const float* dS, int ldds,
                                                               float* dl, int ldl, int n, int K, float* dbias,
                                                               long dbias_stride, const float* dS2) {
    extern __shared__ float colsum[];                 // [16 teams][K]
    const int tl = threadIdx.x & 15, team = threadIdx.x >> 4;
    const int b = blockIdx.y;
    float* mysum = colsum + team * K;
    for (int c = tl; c < K; c += 16) mysum[c] = 0.f;
    const int r0 = blockIdx.x * 64, r1 = min(n, r0 + 64);
    if constexpr (Q4) {
        constexpr int NQ = NK > 0 ? NK : 1;           // quads per lane, K % 4 == 0; one row per team and pass
        const int nq = K >> 2;
        const float* d2base = dS2 ? dS2 : dS;
        for (int node = r0 + team; node < r1; node += 16) {
            const long row = (long)b * n + node;
            f4u sv[NQ], dv[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int qd = min(tl + 16 * k, nq - 1) * 4;
                sv[k] = *reinterpret_cast<const f4u*>(S + row * lds + qd);
                const f4u x = *reinterpret_cast<const f4u*>(dS + row * ldds + qd);
                const f4u y = *reinterpret_cast<const f4u*>(d2base + row * ldds + qd);
                dv[k] = dS2 ? x + y : x;
            }
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < NQ; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) dot += (tl + 16 * k < nq) ? sv[k][j] * dv[k][j] : 0.f;
            dot = team_sum(dot);
            float* o = dl + row * ldl;
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int c = (tl + 16 * k) * 4;
                if (tl + 16 * k < nq) {
                    f4u v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = sv[k][j] * (dv[k][j] - dot);
                    *reinterpret_cast<f4u*>(o + c) = v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) mysum[c + j] += v[j];
                }
            }
        }
    } else if (NK > 0) {
        constexpr int NKK = NK > 0 ? NK : 1;
        for (int base = r0 + team; base < r1; base += 16 * RW) {
            float sv[RW][NKK], dv[RW][NKK];
            const float* d2base = dS2 ? dS2 : dS;            // a valid address either way: no branch around loads
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const long row = (long)b * n + min(base + 16 * j, n - 1);
#pragma unroll
                for (int k = 0; k < NKK; ++k) {
                    const int c = min(tl + 16 * k, K - 1);
                    sv[j][k] = S[row * lds + c];
                    const float x = dS[row * ldds + c], y = d2base[row * ldds + c];
                    dv[j][k] = dS2 ? x + y : x;
                }
            }
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const int node = base + 16 * j;
                float dot = 0.f;
#pragma unroll
                for (int k = 0; k < NKK; ++k) dot += (tl + 16 * k < K) ? sv[j][k] * dv[j][k] : 0.f;
                dot = team_sum(dot);
                if (node < r1) {
                    float* o = dl + ((long)b * n + node) * ldl;
#pragma unroll
                    for (int k = 0; k < NKK; ++k) {
                        const int c = tl + 16 * k;
                        if (c < K) {
                            const float v = sv[j][k] * (dv[j][k] - dot);
                            o[c] = v;
                            mysum[c] += v;
                        }
                    }
                }
            }
        }
    } else {
        for (int node = r0 + team; node < r1; node += 16) {
            const long row = (long)b * n + node;
            const float* s = S + row * lds;
            const float* d = dS + row * ldds;
            const float* d2 = dS2 ? dS2 + row * ldds : nullptr;     // second addend of the incoming gradient (same ld)
            float dot = 0.f;
            for (int c = tl; c < K; c += 16) dot += s[c] * (d[c] + (d2 ? d2[c] : 0.f));
            dot = team_sum(dot);
            float* o = dl + row * ldl;
            for (int c = tl; c < K; c += 16) {
                const float v = s[c] * (d[c] + (d2 ? d2[c] : 0.f) - dot);
                o[c] = v;
                mysum[c] += v;
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += colsum[k * K + c];
        atomicAdd(dbias + (long)b * dbias_stride + c, t);
    }
}
void softmax_mask_bwd(Seq& q, const float* S, int lds, const float* dS, int ldds, const int* num_nodes,
                      float* dlogits, int ldl, int B, int n, int K, float* dbias, long dbias_stride,
                      const float* dS2) {
    if (!q.ok()) return;
    const long rows = (long)B * n;
    if (dbias && (size_t)16 * K * sizeof(float) <= 64 * 1024) {
        const dim3 grid((n + 63) / 64, B);
        const size_t sm = (size_t)16 * K * sizeof(float);
#define DP_SMB(NK, RW)                                                                                            \
    hipLaunchKernelGGL((k_softmax_mask_bwd_plan<NK, RW>), grid, dim3(256), sm, q.stream, S, lds, dS, ldds, dlogits, \
                       ldl, n, K, dbias, dbias_stride, dS2)
        if (K > 128 && K <= 512 && (K & 3) == 0 && !knobs().no_row_quads) {
            if (K <= 256) hipLaunchKernelGGL((k_softmax_mask_bwd_plan<4, 1, true>), grid, dim3(256), sm, q.stream, S, lds,
                                             dS, ldds, dlogits, ldl, n, K, dbias, dbias_stride, dS2);
            else if (K <= 320) hipLaunchKernelGGL((k_softmax_mask_bwd_plan<5, 1, true>), grid, dim3(256), sm, q.stream, S,
                                                  lds, dS, ldds, dlogits, ldl, n, K, dbias, dbias_stride, dS2);
            else hipLaunchKernelGGL((k_softmax_mask_bwd_plan<8, 1, true>), grid, dim3(256), sm, q.stream, S, lds, dS, ldds,
                                    dlogits, ldl, n, K, dbias, dbias_stride, dS2);
        } else
        if (K <= 64) DP_SMB(4, 4);
        else if (K <= 128) DP_SMB(8, 2);
        else if (K <= 256) DP_SMB(16, 1);
        else DP_SMB(0, 1);
#undef DP_SMB
        q.check_launch("softmax_mask_bwd_plan");
        return;
    }
    if (dS2) axpy(q, const_cast<float*>(dS), dS2, 1.f, rows * ldds);      // generic path: fold the addend first
    hipLaunchKernelGGL(k_softmax_mask_bwd, dim3(team_grid(rows)), dim3(256), 0, q.stream, S, lds, dS, ldds,
                       num_nodes, dlogits, ldl, rows, n, K);
    q.check_launch("softmax_mask_bwd");
    if (dbias) colsum_batched(q, dlogits, ldl, (long)n * ldl, n, K, dbias, dbias_stride, B, n >= 256 ? 8 : 1);
}

// ------------------------------------------------------------------ masked max readout
// out[b, f] = max_n (n < n_b ? Z[b, n, f] : 0)   (max over Z * mask, encoders.py:1079-1080,1257).
// Ties -> lowest row index (torch CPU max).  argmax = -1 when the winner is a masked (zero) row.
__global__ __launch_bounds__(1024) void k_masked_max_fwd(const float* Z, int ldz, const int* num_nodes, float* out,
                                                         int ldo, int* argmax, int lda, int n, int F) {
    __shared__ float sv[16][64];
    __shared__ int si[16][64];
    const int b = blockIdx.y;
    const int fl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + fl;
    const int nb = num_nodes ? min(num_nodes[b], n) : n;
    float best = -INFINITY;
    int bi = -1;
    const float* z = Z + (long)b * n * ldz + min(f, F - 1);
    if (n <= 16 * 32) {
        // the thread's (up to) 32 rows are requested in one batch from clamped addresses, compared afterwards: one
        // memory round trip instead of eight (a compare-and-branch behind each group of four loads)
        float zv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) zv[u] = z[(long)min(rl + 16 * u, n - 1) * ldz];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int r = rl + 16 * u;
            if (r < nb && f < F && zv[u] > best) {
                best = zv[u];
                bi = r;
            }
        }
    } else if (f < F) {
#pragma unroll 4
        for (int r = rl; r < nb; r += 16) {
            const float v = z[(long)r * ldz];
            if (v > best) {
                best = v;
                bi = r;
            }
        }
    }
    sv[rl][fl] = best;
    si[rl][fl] = bi;
    __syncthreads();
    if (rl == 0 && f < F) {
        for (int k = 1; k < 16; ++k) {
            const float v = sv[k][fl];
            const int i = si[k][fl];
            if (i >= 0 && (v > best || (v == best && i < bi))) {
                best = v;
                bi = i;
            }
        }
        if (nb < n && !(best > 0.f)) {   // a masked zero row wins; it precedes no valid row only if 0 > best,
            if (best < 0.f || bi < 0) {  // on an exact tie (best == 0) the valid row has the lower index
                best = 0.f;
                bi = -1;
            }
        }
        out[(long)b * ldo + f] = best;
        argmax[(long)b * lda + f] = bi;
    }
}
void masked_max_fwd(Seq& q, const float* Z, int ldz, const int* num_nodes, float* out, int ldo, int* argmax,
                    int lda, int B, int n, int F) {
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_masked_max_fwd, dim3((F + 63) / 64, B), dim3(1024), 0, q.stream, Z, ldz, num_nodes, out,
                       ldo, argmax, lda, n, F);
    q.check_launch("masked_max_fwd");
}

__global__ void k_masked_max_bwd(const float* dout, int ldo, const int* argmax, int lda, float* dZ, int ldz, int B,
                                 int n, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * F) return;
    const int b = i / F, f = i % F;
    const int r = argmax[(long)b * lda + f];
    if (r >= 0) dZ[((long)b * n + r) * ldz + f] += dout[(long)b * ldo + f];
}
void masked_max_bwd(Seq& q, const float* dout, int ldo, const int* argmax, int lda, float* dZ, int ldz, int B,
                    int n, int F) {
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_masked_max_bwd, dim3((B * F + 255) / 256), dim3(256), 0, q.stream, dout, ldo, argmax, lda,
                       dZ, ldz, B, n, F);
    q.check_launch("masked_max_bwd");
}

// dst[b, r, :] = r < num_nodes[b] ? src[b, r, :] : 0     (x_tensor * embedding_mask, encoders.py:1079-1080)
__global__ void k_mask_rows(const float* src, int lds, float* dst, int ldd, const int* num_nodes, int B, int n,
                            int F) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * n * F;
    if (i >= total) return;
    const int f = (int)(i % F);
    const long row = i / F;
    const int b = (int)(row / n), r = (int)(row % n);
    const bool valid = num_nodes ? r < num_nodes[b] : true;
    dst[row * ldd + f] = valid ? src[row * lds + f] : 0.f;
}
void mask_rows(Seq& q, const float* src, int lds, float* dst, int ldd, const int* num_nodes, int B, int n, int F) {
    if (!q.ok()) return;
    const long total = (long)B * n * F;
    hipLaunchKernelGGL(k_mask_rows, dim3((total + 255) / 256), dim3(256), 0, q.stream, src, lds, dst, ldd, num_nodes,
                       B, n, F);
    q.check_launch("mask_rows");
}

// ------------------------------------------------------------------ layer-input dropout
// out[row, :w] = x[row, :w] * m[row, :w]   (m holds 0 or 1/(1-p): nn.Dropout on a GraphConv input, encoders.py:962-964)
__global__ __launch_bounds__(256) void k_mask_mul(const float* x, int ldx, const float* m, float* out, long rows,
                                                  int w) {
    const long total = rows * w;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / w;
        const int c = (int)(e % w);
        out[e] = x[row * ldx + c] * m[e];
    }
}
// dst[row, :w] += src[row, :w] * m[row, :w]   (gradient through the same mask)
__global__ __launch_bounds__(256) void k_mask_axpy(float* dst, int ldd, const float* src, const float* m, long rows,
                                                   int w) {
    const long total = rows * w;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / w;
        const int c = (int)(e % w);
        dst[row * ldd + c] += src[e] * m[e];
    }
}
void mask_mul(Seq& q, const float* x, int ldx, const float* m, float* out, long rows, int w) {
    if (!q.ok() || rows <= 0) return;
    long blocks = (rows * w + 255) / 256;
    hipLaunchKernelGGL(k_mask_mul, dim3((int)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, q.stream, x, ldx, m, out,
                       rows, w);
    q.check_launch("mask_mul");
}
void mask_axpy(Seq& q, float* dst, int ldd, const float* src, const float* m, long rows, int w) {
    if (!q.ok() || rows <= 0) return;
    long blocks = (rows * w + 255) / 256;
    hipLaunchKernelGGL(k_mask_axpy, dim3((int)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, q.stream, dst, ldd, src, m,
                       rows, w);
    q.check_launch("mask_axpy");
}

// ------------------------------------------------------------------ column gather / scatter-add of the two stacks
// out[row] = [x0[row, :w0] | x1[row, :w1]]  (the inputs of an aggregate-first GraphConv layer side by side), and the
// reverse with accumulation: d0[row, :w0] += src[row, :w0], d1[row, :w1] += src[row, w0:]
__global__ __launch_bounds__(256) void k_gather_cols(const float* x0, int ld0, int w0, const float* x1, int ld1, int w1,
                                                     float* out, long rows) {
    const int w = w0 + w1;
    const long total = rows * w;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / w;
        const int c = (int)(e - row * w);
        out[e] = c < w0 ? x0[row * ld0 + c] : x1[row * ld1 + (c - w0)];
    }
}
__global__ __launch_bounds__(256) void k_scatter_add_cols(const float* src, float* d0, int ld0, int w0, float* d1, int ld1,
                                                          int w1, long rows) {
    const int w = w0 + w1;
    const long total = rows * w;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / w;
        const int c = (int)(e - row * w);
        float* d = c < w0 ? d0 + row * ld0 + c : d1 + row * ld1 + (c - w0);
        *d += src[e];
    }
}
// scatter_add_cols + the BatchNorm-backward partials of the rows it completes: one 16-lane team per (row, group),
// widths <= 32.  d_g[row, :w_g] += src[row, c0_g : c0_g + w_g];  part[(row * G + g) * 2 + {0, 1}] = (sum v, sum v * xhat).
__global__ __launch_bounds__(256) void k_scatter_add_cols_part(const float* src, GroupPtrs d, GroupCPtrs xhat, int w0,
                                                               int w1, int G, float* part, long rows) {
    const int tl = threadIdx.x & 15;
    const long team = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long nteams = (long)gridDim.x * 16;
    const int ws = w0 + w1;
    for (long it = team; it < rows * G; it += nteams) {
        const long row = it / G;
        const int g = (int)(it % G);
        const int w = g ? w1 : w0;
        const float* s = src + row * ws + (g ? w0 : 0);
        float* dd = d.p[g] + row * d.ld[g];
        const float* xh = xhat.p[g] + row * xhat.ld[g];
        const int c0 = min(tl, w - 1), c1 = min(tl + 16, w - 1);
        const float a0 = s[c0], a1 = s[c1], b0 = dd[c0], b1 = dd[c1], x0 = xh[c0], x1 = xh[c1];
        const float v0 = tl < w ? a0 + b0 : 0.f, v1 = tl + 16 < w ? a1 + b1 : 0.f;
        if (tl < w) dd[tl] = v0;
        if (tl + 16 < w) dd[tl + 16] = v1;
        const float s0 = team_sum(v0 + v1), s1 = team_sum(v0 * x0 + v1 * x1);
        if (tl == 0) {
            part[it * 2] = s0;
            part[it * 2 + 1] = s1;
        }
    }
}
bool scatter_add_cols_part_supported(int w0, int w1) { return w0 >= 1 && w0 <= 32 && w1 <= 32; }
void scatter_add_cols_part(Seq& q, const float* src, GroupPtrs d, GroupCPtrs xhat, int w0, int w1, int G, float* part,
                           long rows) {
    if (!q.ok() || rows <= 0) return;
    hipLaunchKernelGGL(k_scatter_add_cols_part, dim3(team_grid(rows * G)), dim3(256), 0, q.stream, src, d, xhat, w0, w1, G,
                       part, rows);
    q.check_launch("scatter_add_cols_part");
}
void gather_cols(Seq& q, const float* x0, int ld0, int w0, const float* x1, int ld1, int w1, float* out, long rows) {
    if (!q.ok() || rows <= 0) return;
    const long blocks = (rows * (w0 + w1) + 255) / 256;
    hipLaunchKernelGGL(k_gather_cols, dim3((int)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, q.stream, x0, ld0, w0, x1,
                       ld1, w1, out, rows);
    q.check_launch("gather_cols");
}
void scatter_add_cols(Seq& q, const float* src, float* d0, int ld0, int w0, float* d1, int ld1, int w1, long rows) {
    if (!q.ok() || rows <= 0) return;
    const long blocks = (rows * (w0 + w1) + 255) / 256;
    hipLaunchKernelGGL(k_scatter_add_cols, dim3((int)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, q.stream, src, d0,
                       ld0, w0, d1, ld1, w1, rows);
    q.check_launch("scatter_add_cols");
}

// ------------------------------------------------------------------ zero fill
// Never hipMemsetAsync on this path.  (1) It runs its fill kernel on 256 workgroups whatever the size (17 us for the
// 4 MB of gradient slabs); a plain wide-store kernel is 3-5x faster.  (2) A memset node captured into a hipGraph
// writes garbage from the second replay on with this runtime (ROCm 7.2: the 256-byte pack flag read back
// 0x633f5c00... after replay 1 — tools/graph_memset_probe.py), which silently sent every captured step down the
// fp32 fallback of the aggregation kernels.  Kernels only.
__global__ __launch_bounds__(256) void k_zero16(uint4* p, long n16) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (long k = i; k < n16; k += stride) p[k] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_zero_bytes(unsigned char* p, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (long k = i; k < n; k += stride) p[k] = 0;
}
void zero_small(Seq& q, void* p, size_t bytes) {
    if (!q.ok() || bytes == 0) return;
    long blocks = ((long)bytes + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_zero_bytes, dim3((int)blocks), dim3(256), 0, q.stream, (unsigned char*)p, (long)bytes);
    q.check_launch("zero_small");
}
void zero_fill(Seq& q, void* p, size_t bytes) {
    if (!q.ok() || bytes == 0) return;
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0 || (bytes & 15) != 0) {
        zero_small(q, p, bytes);
        return;
    }
    const long n16 = (long)(bytes / 16);
    long blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_zero16, dim3((int)blocks), dim3(256), 0, q.stream, (uint4*)p, n16);
    q.check_launch("zero_fill");
}

// ------------------------------------------------------------------ small elementwise
__global__ void k_relu_bwd(float* d, const float* h, long count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && !(h[i] > 0.f)) d[i] = 0.f;
}
void relu_bwd_inplace(Seq& q, float* d, const float* h, long count) {
    if (!q.ok() || count <= 0) return;
    hipLaunchKernelGGL(k_relu_bwd, dim3((count + 255) / 256), dim3(256), 0, q.stream, d, h, count);
    q.check_launch("relu_bwd");
}

__global__ void k_axpy(float* y, const float* x, float a, long count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) y[i] += a * x[i];
}
void axpy(Seq& q, float* y, const float* x, float a, long count) {
    if (!q.ok() || count <= 0) return;
    hipLaunchKernelGGL(k_axpy, dim3((count + 255) / 256), dim3(256), 0, q.stream, y, x, a, count);
    q.check_launch("axpy");
}

// ------------------------------------------------------------------ cross entropy
// loss = mean_b (logsumexp(logits_b) - logits_b[label_b])   (F.cross_entropy, encoders.py:1127)
// dunit (optional): d loss / d logits for a unit upstream gradient, (softmax - onehot) / B — the gradient the
// reference's `loss.backward()` (train.py:208) sends down; writing it here lets the backward pass start at the
// prediction head without a launch of its own.
__global__ __launch_bounds__(256) void k_ce_fwd(const float* logits, const long long* label, float* loss,
                                                float* prob, int B, int C, float* also_zero, float* dunit) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* l = logits + (long)b * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, l[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(l[c] - m);
        const float lse = m + logf(s);
        const long long y = label[b];
        acc += lse - l[y];
        if (prob)
            for (int c = 0; c < C; ++c) prob[(long)b * C + c] = expf(l[c] - lse);
        if (dunit)
            for (int c = 0; c < C; ++c)
                dunit[(long)b * C + c] = (expf(l[c] - lse) - (y == c ? 1.f : 0.f)) / (float)B;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss[0] = red[0] / (float)B;
        if (also_zero) also_zero[0] = 0.f;
    }
}
void ce_fwd(Seq& q, const float* logits, const long long* label, float* loss, float* prob, int B, int C,
            float* also_zero, float* dunit) {
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_ce_fwd, dim3(1), dim3(256), 0, q.stream, logits, label, loss, prob, B, C, also_zero, dunit);
    q.check_launch("ce_fwd");
}
__global__ void k_ce_bwd(const float* prob, const long long* label, const float* dloss, float scale, float* dl,
                         int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    const float g = (dloss ? dloss[0] : 1.f) * scale / (float)B;
    dl[i] = g * (prob[i] - (label[b] == c ? 1.f : 0.f));
}
void ce_bwd(Seq& q, const float* prob, const long long* label, const float* dloss, float scale, float* dlogits,
            int B, int C) {
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_ce_bwd, dim3((B * C + 255) / 256), dim3(256), 0, q.stream, prob, label, dloss, scale,
                       dlogits, B, C);
    q.check_launch("ce_bwd");
}

// out[r] = argmax_c X[r, c] (first index on ties) — evaluate()'s torch.max(ypred, 1), train.py:43, for the plans whose
// prediction head is not the fused kernel
__global__ void k_argmax_rows(const float* X, int ldx, long long* out, int rows, int cols) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* x = X + (long)r * ldx;
    int best = 0;
    for (int c = 1; c < cols; ++c)
        if (x[c] > x[best]) best = c;
    out[r] = best;
}
void argmax_rows(Seq& q, const float* X, int ldx, long long* out, int rows, int cols) {
    if (!q.ok() || rows <= 0) return;
    hipLaunchKernelGGL(k_argmax_rows, dim3((rows + 255) / 256), dim3(256), 0, q.stream, X, ldx, out, rows, cols);
    q.check_launch("argmax_rows");
}

// ------------------------------------------------------------------ slab reduce
// out[p] (+)= sum_b slabs[b, p]   — per-graph parameter-gradient slabs -> the flat gradient buffer
__global__ __launch_bounds__(1024) void k_reduce_slabs(const float* slabs, long stride, int B, float* out,
                                                       long count, int accumulate) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (i < count)
        for (int b = rl; b < B; b += 16) s += slabs[(long)b * stride + i];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[i] = accumulate ? out[i] + t : t;
    }
}
void reduce_slabs(Seq& q, const float* slabs, long stride, int B, float* out, long count, int accumulate) {
    if (!q.ok() || count <= 0) return;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((count + 63) / 64), dim3(1024), 0, q.stream, slabs, stride, B, out, count,
                       accumulate);
    q.check_launch("reduce_slabs");
}

}  // namespace dp
