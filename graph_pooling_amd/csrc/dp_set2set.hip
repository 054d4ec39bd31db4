// Set2Set readout (Set2Set.forward, set2set.py:32-57): n sequential steps of
//     q, (h, c) = LSTM(q*, (h, c));  e = emb·q;  a = softmax over ALL n rows;  r = sum a·emb;  q* = [q, r]
// then out = ReLU(Linear(q*)).  The reference runs 6 torch ops per step (n = N_pad steps: 305 ms per step of
// pure launch latency at the ENZYMES shape).  Here the whole recurrence of a graph runs inside
// ONE persistent workgroup: the LSTM weights sit in LDS for all n steps (transposed, W_ih[:, :d] + W_hh
// pre-combined because q = h: gates = (W_ih[:, :d] + W_hh) h + W_ih[:, d:] r + b), the state never leaves
// the CU, and the per-step state needed by backward streams to a save buffer.
// Backward walks the steps in reverse inside one persistent workgroup per graph and only emits the per-step
// vectors (d gates, d r, d e); every weight / embedding gradient is then a plain contraction over (graph, step)
// done by the MFMA GEMM:  dW_ih = DG^T QP,  dW_hh = DG^T QP[:, :d],  demb = A^T DR + DE^T H.
#include "dp_common.h"

namespace dp {

#ifdef DP_STAMP
// diagnostic build only: cycles per phase of workgroup 0, summed over the recurrence steps (tools/s2s_stamps.py)
__device__ unsigned long long g_s2s_stamps[2][16];
#define S2S_T0() unsigned long long s2s_t_ = __builtin_amdgcn_s_memtime()
#define S2S_ACC(k, i)                                                    \
    do {                                                                 \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();      \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_s2s_stamps[k][i] += n_ - s2s_t_; \
        s2s_t_ = n_;                                                     \
    } while (0)
#define S2S_ZERO(k)                                                                        \
    do {                                                                                   \
        if (blockIdx.x == 0 && threadIdx.x < 16) g_s2s_stamps[k][threadIdx.x] = 0;         \
    } while (0)
#else
#define S2S_T0() \
    do {         \
    } while (0)
#define S2S_ACC(k, i) \
    do {              \
    } while (0)
#define S2S_ZERO(k) \
    do {            \
    } while (0)
#endif

namespace {

struct S2SLayout {   // offsets in floats
    size_t wt, qp, h, c, g, a, qn, total;
    int GS;
};
S2SLayout s2s_layout(int B, int n, int d) {
    S2SLayout L{};
    L.GS = 4 * d + 1;
    size_t off = 0;
    auto take = [&](size_t cnt) {
        size_t o = off;
        off += (cnt + 63) & ~size_t(63);
        return o;
    };
    L.wt = take((size_t)2 * d * L.GS);
    L.qp = take((size_t)B * n * 2 * d);
    L.h = take((size_t)B * n * d);
    L.c = take((size_t)B * n * d);
    L.g = take((size_t)B * n * 4 * d);
    L.a = take((size_t)B * n * n);
    L.qn = take((size_t)B * 2 * d);
    L.total = off;
    return L;
}

__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ inline float team16_sum(float v) {
    return row16_sum(v);
}

// block-wide reductions over 256 threads (red: 8 floats of LDS)
__device__ inline float block_max(float v, float* red) {
    v = wave64_max(v);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ inline float block_sum(float v, float* red) {
    v = wave64_sum(v);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    return red[0] + red[1] + red[2] + red[3];
}

// Wt[k][g] (row stride GS = 4d+1): k < d: W_ih[g][k] + W_hh[g][k];  d <= k < 2d: W_ih[g][k]
__global__ void k_s2s_prep(const float* w_ih, const float* w_hh, float* wt, int d, int GS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * d * 4 * d) return;
    const int k = i / (4 * d), g = i % (4 * d);
    float v = w_ih[(long)g * 2 * d + k];
    if (k < d) v += w_hh[(long)g * d + k];
    wt[(long)k * GS + g] = v;
}

struct S2SFwdArgs {
    const float* emb;
    int lde;
    const float* wt;       // [2d][GS] global copy
    const float* b_ih;
    const float* b_hh;
    const float* Wp;       // [d][2d]
    const float* bp;
    float* out;            // [B, d]
    float *QP, *H, *Cs, *G, *Aw, *QN;   // save arrays (may be null: inference)
    int n, d, GS;
    int w_in_lds;
    int emb_in_lds;        // the graph's embedding [n][d] is staged once and every step reads LDS (it fits beside W)
};

// Round 3: 1024 threads per graph.  The in-kernel stamps (tools/s2s_stamps.py, profiles/r03_s2s_stamps.txt) put a
// recurrence step at 18.3k cycles forward / 25.4k backward with 256 threads: 7.0k for the gates (240 threads each
// walking a 120-term dot product out of LDS), 14.1k for [dh, dr] = Wt dg (120 threads, 240 terms each), the rest in
// row passes over the embedding and two block reductions of two barriers each.  Now a dot product is spread over the
// lanes of a 4- or 8-lane team (DPP sums), the softmax and the <a, da> sum run inside ONE wave (no block reductions),
// and the row passes use 64 teams / 16 row parts.
constexpr int S2S_NT = 1024;
constexpr int S2S_GPT = 4;         // gate teams per thread slot: 4d <= 1024 gates, 256 four-lane teams

// WL / EL: the weights / the graph's embedding are staged in LDS.  Compile-time, so that every access has a known address
// space: with `W = a.w_in_lds ? lw : a.wt` the pointer was GENERIC, every read a flat_load — out of order with both
// counters, so each one is waited for in full (the stamps had a 30-term dot product at ~1000 cycles per two terms).
template <bool WL, bool EL, bool WF>
__global__ __launch_bounds__(S2S_NT) void k_set2set_fwd(S2SFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, d = a.d, GS = a.GS;
    float* lw = lds;                                   // [2d][GS] when WL
    float* vec = lds + (WL ? 2 * d * GS : 0);
    float* h = vec;                                    // [d]   } contiguous: the LSTM input q* = [h, r]
    float* r = h + d;                                  // [d]   }
    float* c = r + d;                                  // [d]
    float* gates = c + d;                              // [4d]  ACTIVATED gates i, f, g, o
    float* rpart = gates + 4 * d + 8;                  // [16][64]
    float* al = rpart + 1024;                          // [n]
    float* le = al + ((n + 15) & ~15);                 // [n][d] when emb_in_lds
    if constexpr (WL)
        for (int i = tid; i < 2 * d * GS; i += S2S_NT) lw[i] = a.wt[i];
    for (int i = tid; i < 3 * d; i += S2S_NT) vec[i] = 0.f;    // h, r, c = 0 (set2set.py:42-45)
    const float* embg = a.emb + (long)b * n * a.lde;
    if constexpr (EL)
        for (int i = tid; i < n * d; i += S2S_NT) le[i] = embg[(long)(i / d) * a.lde + i % d];
    __syncthreads();
    // (two differently typed pointers per operand: the branches below pick one at compile time)
    auto Wat = [&](int i) -> float {
        if constexpr (WL) return lw[i];
        else return a.wt[i];
    };
    const int lde = EL ? d : a.lde;
    auto Eat = [&](long i) -> float {
        if constexpr (EL) return le[i];
        else return embg[i];
    };
    const int lane = tid & 63, wave = tid >> 6;
    const int tl = tid & 15, team = tid >> 4;          // 64 sixteen-lane teams
    const int q4 = tid & 3, gteam = tid >> 2;          // 256 four-lane gate teams
    float gbias[S2S_GPT];
#pragma unroll
    for (int i = 0; i < S2S_GPT; ++i) {
        const int g = min(gteam + 256 * i, 4 * d - 1);
        gbias[i] = a.b_ih[g] + a.b_hh[g];
    }
    const int kq = (2 * d + 3) >> 2;                   // terms per lane of a gate team
    const int k0 = q4 * kq, k1 = min(2 * d, k0 + kq);
    // The weights do not change over the n steps: when a gate team owns ONE gate (4d <= 256) its lanes keep their
    // <= 32 weights in registers for the whole recurrence — a step's gate phase is then ~8 LDS reads of [h, r] and
    // 30 FMAs per lane instead of 60 LDS reads with their address arithmetic (the phase was VALU-issue-bound:
    // 16 waves x ~450 instructions for 450 wave-FMAs of useful work)
    constexpr int WREG = 32;                           // WF (launcher): 4d <= 256 and ceil(2d / 4) <= 32
    float wreg[WF ? WREG : 1];
    if constexpr (WF) {
        const int gg = min(gteam, 4 * d - 1);
#pragma unroll
        for (int i = 0; i < WREG; ++i) wreg[i] = k0 + i < k1 ? a.wt[(long)(k0 + i) * GS + gg] : 0.f;
    }

    S2S_ZERO(0);
    S2S_T0();
    for (int t = 0; t < n; ++t) {
        // ---- q*_{t-1} = [h, r] is the LSTM input of this step
        if (a.QP && tid < 2 * d) a.QP[((long)b * n + t) * 2 * d + tid] = h[tid];
        // ---- gates = act(b + Wc h + Wr r): four lanes per gate, 2d / 4 terms each
        if constexpr (WF) {
            float acc = 0.f, acc2 = 0.f;
#pragma unroll
            for (int i = 0; i < WREG; i += 2) {
                acc += wreg[i] * h[min(k0 + i, 2 * d - 1)];
                acc2 += wreg[i + 1] * h[min(k0 + i + 1, 2 * d - 1)];
            }
            acc = quad4_sum(acc + acc2);
            if (q4 == 0 && gteam < 4 * d) {
                const float pre = acc + gbias[0];
                const float act = (gteam >= 2 * d && gteam < 3 * d) ? tanhf(pre) : sigmoidf_(pre);
                gates[gteam] = act;
                if (a.G) a.G[((long)b * n + t) * 4 * d + gteam] = act;
            }
        } else
#pragma unroll
        for (int i = 0; i < S2S_GPT; ++i) {
            if (256 * i >= 4 * d) break;               // (uniform)
            const int g = gteam + 256 * i, gg = min(g, 4 * d - 1);
            float acc = 0.f, acc2 = 0.f;
            int k = k0;
            for (; k + 1 < k1; k += 2) {
                acc += Wat(k * GS + gg) * h[k];
                acc2 += Wat((k + 1) * GS + gg) * h[k + 1];
            }
            if (k < k1) acc += Wat(k * GS + gg) * h[k];
            acc = quad4_sum(acc + acc2);
            if (q4 == 0 && g < 4 * d) {
                const float pre = acc + gbias[i];
                const float act = (gg >= 2 * d && gg < 3 * d) ? tanhf(pre) : sigmoidf_(pre);
                gates[gg] = act;
                if (a.G) a.G[((long)b * n + t) * 4 * d + gg] = act;
            }
        }
        lds_barrier();
        S2S_ACC(0, 0);
        // ---- LSTM cell (gate order i, f, g, o)
        if (tid < d) {
            const int j = tid;
            const float cn = gates[d + j] * c[j] + gates[j] * gates[2 * d + j];
            const float hn = gates[3 * d + j] * tanhf(cn);
            c[j] = cn;
            h[j] = hn;
            if (a.G) {
                a.Cs[((long)b * n + t) * d + j] = cn;
                a.H[((long)b * n + t) * d + j] = hn;
            }
        }
        lds_barrier();
        S2S_ACC(0, 1);
        // ---- e = emb . h  (all n rows, padded rows included — set2set.py:50-51)
        for (int row = team; row < n; row += 64) {
            float s = 0.f;
            for (int k = tl; k < d; k += 16) s += Eat((long)row * lde + k) * h[k];
            s = team16_sum(s);
            if (tl == 0) al[row] = s;
        }
        lds_barrier();
        S2S_ACC(0, 2);
        // ---- a = softmax(e) over all n rows: inside one wave (n <= 1024: 16 rows per lane)
        if (wave == 0) {
            float m = -INFINITY;
            for (int row = lane; row < n; row += 64) m = fmaxf(m, al[row]);
            m = wave64_max(m);
            float sum = 0.f;
            for (int row = lane; row < n; row += 64) {
                const float p = expf(al[row] - m);
                al[row] = p;
                sum += p;
            }
            const float inv = 1.f / wave64_sum(sum);
            for (int row = lane; row < n; row += 64) {
                const float p = al[row] * inv;
                al[row] = p;
                if (a.Aw) a.Aw[((long)b * n + t) * n + row] = p;
            }
        }
        lds_barrier();
        S2S_ACC(0, 3);
        // ---- r = sum_n a[n] emb[n]: 16 row parts (one per wave) x 64 columns
        for (int j0 = 0; j0 < d; j0 += 64) {
            const int j = j0 + lane;
            float s = 0.f;
            if (j < d) {
#pragma unroll 4
                for (int row = wave; row < n; row += 16) s += al[row] * Eat((long)row * lde + j);
            }
            rpart[wave * 64 + lane] = s;
            lds_barrier();
            if (wave == 0 && j < d) {
                float tsum = 0.f;
#pragma unroll
                for (int p = 0; p < 16; ++p) tsum += rpart[p * 64 + lane];
                r[j] = tsum;
            }
            lds_barrier();
        }
        S2S_ACC(0, 4);
    }
    // ---- out = relu(Wp [h, r] + bp)
    if (a.QN && tid < 2 * d) a.QN[(long)b * 2 * d + tid] = h[tid];
    for (int j = tid; j < d; j += S2S_NT) {
        float acc = a.bp[j];
        const float* wr = a.Wp + (long)j * 2 * d;
        for (int k = 0; k < 2 * d; ++k) acc += wr[k] * h[k];
        a.out[(long)b * d + j] = fmaxf(acc, 0.f);
    }
}

struct S2SBwdArgs {
    const float* emb;
    int lde;
    const float* wt;
    const float* Wp;
    const float* out;
    const float* dout;
    const float *H, *Cs, *G, *Aw;
    float *DG, *DR, *DE, *DPRE;    // workspace outputs
    int n, d, GS;
    int w_in_lds;
    int emb_in_lds;
};

template <bool WL, bool EL, bool WF>
__global__ __launch_bounds__(S2S_NT) void k_set2set_bwd(S2SBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, d = a.d, GS = a.GS;
    float* lw = lds;
    float* vec = lds + (WL ? 2 * d * GS : 0);
    float* dh = vec;                 // [d]  } contiguous: d q* = [dh, dr]
    float* dr = dh + d;              // [d]  }
    float* dc = dr + d;              // [d]
    float* dg = dc + d;              // [4d]
    float* rpart = dg + 4 * d + 8;   // [16][64]
    float* de = rpart + 1024;        // [n]
    float* lat = de + ((n + 15) & ~15);  // [n] attention weights a_t of the step being processed
    float* le = lat + ((n + 15) & ~15);  // [n][d] when emb_in_lds
    if constexpr (WL)
        for (int i = tid; i < 2 * d * GS; i += S2S_NT) lw[i] = a.wt[i];
    const float* embg = a.emb + (long)b * n * a.lde;
    if constexpr (EL)
        for (int i = tid; i < n * d; i += S2S_NT) le[i] = embg[(long)(i / d) * a.lde + i % d];
    auto Wat = [&](long i) -> float {
        if constexpr (WL) return lw[i];
        else return a.wt[i];
    };
    auto Eat = [&](long i) -> float {
        if constexpr (EL) return le[i];
        else return embg[i];
    };
    // ---- output layer: dpre = dout * (out > 0);  [dh, dr] = Wp^T dpre;  dc = 0
    for (int j = tid; j < d; j += S2S_NT) {
        const float o = a.out[(long)b * d + j];
        const float v = o > 0.f ? a.dout[(long)b * d + j] : 0.f;
        dg[j] = v;                                   // borrow dg[0:d] for dpre
        a.DPRE[(long)b * d + j] = v;
        dc[j] = 0.f;
    }
    __syncthreads();
    for (int k = tid; k < 2 * d; k += S2S_NT) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) s += a.Wp[(long)j * 2 * d + k] * dg[j];
        dh[k] = s;                                   // (k >= d lands in dr)
    }
    __syncthreads();
    const int lde = EL ? d : a.lde;
    const int lane = tid & 63, wave = tid >> 6;
    const int tl = tid & 15, team = tid >> 4;
    const int q8 = tid & 7, oteam = tid >> 3;         // 128 eight-lane teams for [dh, dr] = Wt dg
    const int gq = (4 * d + 7) >> 3;                   // terms per lane there
    const int g0 = q8 * gq, g1 = min(4 * d, g0 + gq);
    constexpr int WREG = 32;                           // (as the forward kernel: one output per team, weights in registers)
    float wreg[WF ? WREG : 1];
    if constexpr (WF) {
        const long wr = (long)min(oteam, 2 * d - 1) * GS;
#pragma unroll
        for (int i = 0; i < WREG; ++i) wreg[i] = g0 + i < g1 ? a.wt[wr + g0 + i] : 0.f;
    }

    // The saved per-step state (gates, c_t, c_{t-1}) is read from global memory: fetched one step AHEAD into registers
    // so the loads fly under the previous step; a_t goes the same way for the lanes of wave 0 (16 rows each).
    constexpr int APT = 4;                            // prefetched a_t rows per lane of wave 0 (rows >= 256: read in place)
    float at_n[APT], gn[6];
    auto prefetch = [&](int t) {
        const int tc = t < 0 ? 0 : t;                 // (the last prefetch is unused; keep the address valid)
        if (wave == 0) {
            const float* at = a.Aw + ((long)b * n + tc) * n;
#pragma unroll
            for (int i = 0; i < APT; ++i) at_n[i] = at[min(lane + 64 * i, n - 1)];
        }
        const int j = min(tid, d - 1);
        const float* gs = a.G + ((long)b * n + tc) * 4 * d;
        gn[0] = gs[j]; gn[1] = gs[d + j]; gn[2] = gs[2 * d + j]; gn[3] = gs[3 * d + j];
        gn[4] = a.Cs[((long)b * n + tc) * d + j];
        gn[5] = tc > 0 ? a.Cs[((long)b * n + tc - 1) * d + j] : 0.f;
    };
    prefetch(n - 1);
    S2S_ZERO(1);
    S2S_T0();
    for (int t = n - 1; t >= 0; --t) {
        // (this step's saved state sits in at_n / gn: requested at the end of the previous step)
        // ---- r_t = sum a emb:  da = emb . dr ;  de = a * (da - sum a da)
        if (tid < d) a.DR[((long)b * n + t) * d + tid] = dr[tid];
        for (int row = team; row < n; row += 64) {
            float s = 0.f;
            for (int k = tl; k < d; k += 16) s += Eat((long)row * lde + k) * dr[k];
            s = team16_sum(s);
            if (tl == 0) de[row] = s;
        }
        lds_barrier();
        S2S_ACC(1, 0);
        if (wave == 0) {
            const float* at = a.Aw + ((long)b * n + t) * n;
            float lsum = 0.f;
#pragma unroll
            for (int i = 0; i < APT; ++i) {
                const int row = lane + 64 * i;
                lsum += row < n ? at_n[i] * de[row] : 0.f;
            }
            for (int row = lane + 64 * APT; row < n; row += 64) lsum += at[row] * de[row];
            const float sdot = wave64_sum(lsum);
#pragma unroll
            for (int i = 0; i < APT; ++i) {
                const int row = lane + 64 * i;
                if (row < n) {
                    const float v = at_n[i] * (de[row] - sdot);
                    de[row] = v;
                    a.DE[((long)b * n + t) * n + row] = v;
                }
            }
            for (int row = lane + 64 * APT; row < n; row += 64) {
                const float v = at[row] * (de[row] - sdot);
                de[row] = v;
                a.DE[((long)b * n + t) * n + row] = v;
            }
        }
        lds_barrier();
        S2S_ACC(1, 1);
        // ---- e = emb . h_t:  dh += sum_n de[n] emb[n]
        for (int j0 = 0; j0 < d; j0 += 64) {
            const int j = j0 + lane;
            float s = 0.f;
            if (j < d) {
#pragma unroll 4
                for (int row = wave; row < n; row += 16) s += de[row] * Eat((long)row * lde + j);
            }
            rpart[wave * 64 + lane] = s;
            lds_barrier();
            if (wave == 0 && j < d) {
                float tsum = 0.f;
#pragma unroll
                for (int p = 0; p < 16; ++p) tsum += rpart[p * 64 + lane];
                dh[j] += tsum;
            }
            lds_barrier();
        }
        S2S_ACC(1, 2);
        // ---- LSTM cell backward
        if (tid < d) {
            const int j = tid;
            const float ig = gn[0], fg = gn[1], gg = gn[2], og = gn[3];
            const float ct = gn[4];
            const float cp = gn[5];
            const float tc = tanhf(ct);
            const float dhj = dh[j];
            const float dct = dc[j] + dhj * og * (1.f - tc * tc);
            const float d_i = dct * gg * ig * (1.f - ig);
            const float d_f = dct * cp * fg * (1.f - fg);
            const float d_g = dct * ig * (1.f - gg * gg);
            const float d_o = dhj * tc * og * (1.f - og);
            dg[j] = d_i; dg[d + j] = d_f; dg[2 * d + j] = d_g; dg[3 * d + j] = d_o;
            float* o = a.DG + ((long)b * n + t) * 4 * d;
            o[j] = d_i; o[d + j] = d_f; o[2 * d + j] = d_g; o[3 * d + j] = d_o;
            dc[j] = dct * fg;
        }
        prefetch(t - 1);                               // the next step's saved state flies under the product below
        lds_barrier();
        S2S_ACC(1, 3);
        // ---- [dh_{t-1}, dr_{t-1}] = Wt dg   (h_{t-1} and r_{t-1} feed only this step's LSTM): eight lanes per output
        if constexpr (WF) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < WREG; i += 2) {
                s0 += wreg[i] * dg[min(g0 + i, 4 * d - 1)];
                s1 += wreg[i + 1] * dg[min(g0 + i + 1, 4 * d - 1)];
            }
            s0 = oct8_sum(s0 + s1);
            if (q8 == 0 && oteam < 2 * d) dh[oteam] = s0;      // (k >= d lands in dr)
        } else
        for (int k = oteam; k < 2 * d; k += 128) {
            const long wr = (long)k * GS;
            float s = 0.f, s2 = 0.f;
            int g = g0;
            for (; g + 1 < g1; g += 2) {
                s += Wat(wr + g) * dg[g];
                s2 += Wat(wr + g + 1) * dg[g + 1];
            }
            if (g < g1) s += Wat(wr + g) * dg[g];
            s = oct8_sum(s + s2);
            if (q8 == 0) dh[k] = s;                    // (k >= d lands in dr)
        }
        lds_barrier();
        S2S_ACC(1, 4);
    }
}

size_t s2s_dyn_lds(int n, int d, bool w_in_lds, bool emb_in_lds = false) {
    const size_t GS = 4 * d + 1;
    return ((w_in_lds ? (size_t)2 * d * GS : 0) + 7 * d + 8 + 1024 + 2 * ((n + 15) & ~15) + 16 +
            (emb_in_lds ? (size_t)n * d : 0)) * sizeof(float);
}
bool s2s_w_fits(int n, int d) { return s2s_dyn_lds(n, d, true) <= 158 * 1024; }
// the lanes of a gate / output team keep their weights in registers for the whole recurrence (k_set2set_*: WF)
bool s2s_w_in_regs(int d) { return 4 * d <= 256 && ((2 * d + 3) >> 2) <= 32 && ((4 * d + 7) >> 3) <= 32; }
bool s2s_emb_fits(int n, int d) { return s2s_dyn_lds(n, d, s2s_w_fits(n, d), true) <= 158 * 1024; }

}  // namespace

size_t set2set_save_bytes(int B, int n, int d) { return s2s_layout(B, n, d).total * sizeof(float) + 256; }

void set2set_fwd(Seq& q, const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                 const float* b_hh, const float* Wp, const float* bp, float* out, int B, int n, int d, void* save) {
    if (q.err) return;
    const S2SLayout L = s2s_layout(B, n, d);
    if (!q.ok()) return;
    if (s2s_dyn_lds(n, d, false) > 158 * 1024 || d > 256 || n > 1024) {
        set_error("Set2Set: n=%d / d=%d outside the persistent kernel's limits (n <= 1024, d <= 256)", n, d);
        q.err = DP_ERR_UNSUPPORTED;
        return;
    }
    float* sv = (float*)save;
    float* wt = sv + L.wt;
    hipLaunchKernelGGL(k_s2s_prep, dim3((8 * d * d + 255) / 256), dim3(256), 0, q.stream, w_ih, w_hh, wt, d, L.GS);
    q.check_launch("s2s_prep");
    S2SFwdArgs a{};
    a.emb = emb; a.lde = lde; a.wt = wt; a.b_ih = b_ih; a.b_hh = b_hh; a.Wp = Wp; a.bp = bp; a.out = out;
    a.QP = sv + L.qp; a.H = sv + L.h; a.Cs = sv + L.c; a.G = sv + L.g; a.Aw = sv + L.a; a.QN = sv + L.qn;
    a.n = n; a.d = d; a.GS = L.GS;
    const bool wf = s2s_w_in_regs(d);
    a.w_in_lds = !wf && s2s_w_fits(n, d) ? 1 : 0;
    a.emb_in_lds = s2s_dyn_lds(n, d, a.w_in_lds, true) <= 158 * 1024 ? 1 : 0;
    const size_t ldsb = s2s_dyn_lds(n, d, a.w_in_lds, a.emb_in_lds);
    auto go = [&](auto kern, DynLdsOnce& at) {
        ensure_dyn_lds(q, at, reinterpret_cast<const void*>(kern), 160 * 1024, "k_set2set_fwd");
        if (!q.ok()) return;
        hipLaunchKernelGGL(kern, dim3(B), dim3(S2S_NT), ldsb, q.stream, a);
    };
    static DynLdsOnce at00, at10, at11, af0, af1;
    if (wf && a.emb_in_lds) go(&k_set2set_fwd<false, true, true>, af1);
    else if (wf) go(&k_set2set_fwd<false, false, true>, af0);
    else if (a.w_in_lds && a.emb_in_lds) go(&k_set2set_fwd<true, true, false>, at11);
    else if (a.w_in_lds) go(&k_set2set_fwd<true, false, false>, at10);
    else go(&k_set2set_fwd<false, false, false>, at00);
    q.check_launch("set2set_fwd");
}

void set2set_bwd(Seq& q, const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                 const float* b_hh, const float* Wp, const float* bp, const float* out, const float* dout,
                 float* demb, int ldde, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dWp,
                 float* dbp, int B, int n, int d, const void* save) {
    if (q.err) return;
    const S2SLayout L = s2s_layout(B, n, d);
    float* DG = q.alloc<float>((size_t)B * n * 4 * d);
    float* DR = q.alloc<float>((size_t)B * n * d);
    float* DE = q.alloc<float>((size_t)B * n * n);
    float* DPRE = q.alloc<float>((size_t)B * d);
    if (!q.ok()) return;
    const float* sv = (const float*)save;
    S2SBwdArgs a{};
    a.emb = emb; a.lde = lde; a.wt = sv + L.wt; a.Wp = Wp; a.out = out; a.dout = dout;
    a.H = sv + L.h; a.Cs = sv + L.c; a.G = sv + L.g; a.Aw = sv + L.a;
    a.DG = DG; a.DR = DR; a.DE = DE; a.DPRE = DPRE;
    a.n = n; a.d = d; a.GS = L.GS;
    const bool wf = s2s_w_in_regs(d);
    a.w_in_lds = !wf && s2s_w_fits(n, d) ? 1 : 0;
    a.emb_in_lds = s2s_dyn_lds(n, d, a.w_in_lds, true) <= 158 * 1024 ? 1 : 0;
    const size_t ldsb = s2s_dyn_lds(n, d, a.w_in_lds, a.emb_in_lds);
    auto go = [&](auto kern, DynLdsOnce& at) {
        ensure_dyn_lds(q, at, reinterpret_cast<const void*>(kern), 160 * 1024, "k_set2set_bwd");
        if (!q.ok()) return;
        hipLaunchKernelGGL(kern, dim3(B), dim3(S2S_NT), ldsb, q.stream, a);
    };
    static DynLdsOnce at00, at10, at11, af0, af1;
    if (wf && a.emb_in_lds) go(&k_set2set_bwd<false, true, true>, af1);
    else if (wf) go(&k_set2set_bwd<false, false, true>, af0);
    else if (a.w_in_lds && a.emb_in_lds) go(&k_set2set_bwd<true, true, false>, at11);
    else if (a.w_in_lds) go(&k_set2set_bwd<true, false, false>, at10);
    else go(&k_set2set_bwd<false, false, false>, at00);
    q.check_launch("set2set_bwd");
    const float* QP = sv + L.qp;
    const float* QN = sv + L.qn;
    const float* H = sv + L.h;
    const float* Aw = sv + L.a;
    const int T = n;
    // output layer: dWp = dpre^T [h_n, r_n];  dbp = colsum(dpre)
    bgemm(q, DPRE, QN, dWp, nullptr, 1, d, 2 * d, B, d, 2 * d, 2 * d, 0, 0, 0, true, false, 1.f, 0.f, 0);
    colsum_batched(q, DPRE, d, 0, B, d, dbp, 0, 1);
    // LSTM weights: contractions over all (graph, step) rows
    {
        GemmDesc g[2] = {
            {DG, QP, dw_ih, nullptr, 4 * d, 2 * d, B * T, 4 * d, 2 * d, 2 * d, 0, 0, 0, true, false, 1.f, 0.f, 0},
            {DG, QP, dw_hh, nullptr, 4 * d, d, B * T, 4 * d, 2 * d, d, 0, 0, 0, true, false, 1.f, 0.f, 0}};
        bgemm_group(q, g, 2, 1);
    }
    colsum_batched(q, DG, 4 * d, 0, B * T, 4 * d, db_ih, 0, 1);
    // b_ih and b_hh enter the gates as a sum: the same gradient, written by the same kernel (no memcpy node on this path)
    colsum_batched(q, DG, 4 * d, 0, B * T, 4 * d, db_hh, 0, 1);
    // demb[b] = A_b^T DR_b + DE_b^T H_b
    bgemm(q, Aw, DR, demb, nullptr, B, n, d, T, n, d, ldde, (long)T * n, (long)T * d, (long)n * ldde, true, false, 1.f,
          0.f, 0);
    bgemm(q, DE, H, demb, nullptr, B, n, d, T, n, d, ldde, (long)T * n, (long)T * d, (long)n * ldde, true, false, 1.f,
          1.f, 0);
}

#ifdef DP_STAMP
extern "C" __attribute__((visibility("default"))) int dp_debug_s2s_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s2s_stamps), sizeof(unsigned long long) * 32);
}
#endif

}  // namespace dp
