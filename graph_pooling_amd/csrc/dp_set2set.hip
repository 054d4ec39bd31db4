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

__global__ __launch_bounds__(256) void k_set2set_fwd(S2SFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, d = a.d, GS = a.GS;
    float* lw = lds;                                   // [2d][GS] when w_in_lds
    float* vec = lds + (a.w_in_lds ? 2 * d * GS : 0);
    float* h = vec;                                    // [d]
    float* c = h + d;                                  // [d]
    float* r = c + d;                                  // [d]
    float* gates = r + d;                              // [4d]
    float* red = gates + 4 * d;                        // [8]
    float* rpart = red + 8;                            // [4][64]
    float* al = rpart + 256;                           // [n]
    float* le = al + ((n + 15) & ~15);                 // [n][d] when emb_in_lds
    const float* W = a.w_in_lds ? lw : a.wt;
    if (a.w_in_lds)
        for (int i = tid; i < 2 * d * GS; i += 256) lw[i] = a.wt[i];
    for (int i = tid; i < 3 * d; i += 256) vec[i] = 0.f;       // h, c, r = 0 (set2set.py:42-45)
    const float* emb = a.emb + (long)b * n * a.lde;
    int lde = a.lde;
    if (a.emb_in_lds) {
        // the 2 n passes over the embedding (e = emb h, r = a^T emb, every step) were global reads: ~7 dependent
        // L2 round trips per step, most of the step's time at n = 100
        for (int i = tid; i < n * d; i += 256) le[i] = emb[(long)(i / d) * a.lde + i % d];
        emb = le;
        lde = d;
    }
    lds_barrier();
    const int tl = tid & 15, team = tid >> 4;
    // per-thread gate biases (were two global loads per gate per step, at the head of every step's critical path)
    constexpr int GPT = 4;                            // gates per thread: 4d <= 1024
    float gbias[GPT];
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int g = tid + 256 * i;
        gbias[i] = g < 4 * d ? a.b_ih[g] + a.b_hh[g] : 0.f;
    }

    for (int t = 0; t < n; ++t) {
        // ---- q*_{t-1} = [h, r] is the LSTM input of this step
        if (a.QP)
            for (int i = tid; i < 2 * d; i += 256) a.QP[((long)b * n + t) * 2 * d + i] = i < d ? h[i] : r[i - d];
        // ---- gates = b + Wc h + Wr r
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int g = tid + 256 * i;
            if (g >= 4 * d) break;
            float acc = gbias[i], acc2 = 0.f;
#pragma unroll 8
            for (int k = 0; k < d; ++k) acc += W[k * GS + g] * h[k];
#pragma unroll 8
            for (int k = 0; k < d; ++k) acc2 += W[(d + k) * GS + g] * r[k];
            gates[g] = acc + acc2;
        }
        lds_barrier();
        // ---- LSTM cell (gate order i, f, g, o)
        for (int j = tid; j < d; j += 256) {
            const float ig = sigmoidf_(gates[j]), fg = sigmoidf_(gates[d + j]);
            const float gg = tanhf(gates[2 * d + j]), og = sigmoidf_(gates[3 * d + j]);
            const float cn = fg * c[j] + ig * gg;
            const float hn = og * tanhf(cn);
            c[j] = cn;
            h[j] = hn;
            if (a.G) {
                float* gs = a.G + ((long)b * n + t) * 4 * d;
                gs[j] = ig; gs[d + j] = fg; gs[2 * d + j] = gg; gs[3 * d + j] = og;
                a.Cs[((long)b * n + t) * d + j] = cn;
                a.H[((long)b * n + t) * d + j] = hn;
            }
        }
        lds_barrier();
        // ---- e = emb . h  (all n rows, padded rows included — set2set.py:50-51)
        float lmax = -INFINITY;
        for (int row = team; row < n; row += 16) {
            const float* er = emb + (long)row * lde;
            float s = 0.f;
            for (int k = tl; k < d; k += 16) s += er[k] * h[k];
            s = team16_sum(s);
            if (tl == 0) al[row] = s;
            lmax = fmaxf(lmax, s);
        }
        const float m = block_max(lmax, red);
        float lsum = 0.f;
        for (int row = tid; row < n; row += 256) {
            const float p = expf(al[row] - m);
            al[row] = p;
            lsum += p;
        }
        const float inv = 1.f / block_sum(lsum, red);
        for (int row = tid; row < n; row += 256) {
            const float v = al[row] * inv;
            al[row] = v;
            if (a.Aw) a.Aw[((long)b * n + t) * n + row] = v;
        }
        lds_barrier();
        // ---- r = sum_n a[n] emb[n]
        for (int j0 = 0; j0 < d; j0 += 64) {
            const int j = j0 + (tid & 63), part = tid >> 6;
            float s = 0.f;
            if (j < d) {
#pragma unroll 8
                for (int row = part; row < n; row += 4) s += al[row] * emb[(long)row * lde + j];
            }
            rpart[part * 64 + (tid & 63)] = s;
            lds_barrier();
            if (part == 0 && j < d) r[j] = rpart[tid] + rpart[64 + tid] + rpart[128 + tid] + rpart[192 + tid];
            lds_barrier();
        }
    }
    // ---- out = relu(Wp [h, r] + bp)
    if (a.QN)
        for (int i = tid; i < 2 * d; i += 256) a.QN[(long)b * 2 * d + i] = i < d ? h[i] : r[i - d];
    for (int j = tid; j < d; j += 256) {
        float acc = a.bp[j];
        const float* wr = a.Wp + (long)j * 2 * d;
        for (int k = 0; k < d; ++k) acc += wr[k] * h[k];
        for (int k = 0; k < d; ++k) acc += wr[d + k] * r[k];
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

__global__ __launch_bounds__(256) void k_set2set_bwd(S2SBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, d = a.d, GS = a.GS;
    float* lw = lds;
    float* vec = lds + (a.w_in_lds ? 2 * d * GS : 0);
    float* dh = vec;                 // [d]
    float* dc = dh + d;              // [d]
    float* dr = dc + d;              // [d]
    float* dg = dr + d;              // [4d]
    float* red = dg + 4 * d;         // [8]
    float* rpart = red + 8;          // [256]
    float* de = rpart + 256;         // [n]
    float* lat = de + ((n + 15) & ~15);  // [n] attention weights a_t of the step being processed
    float* le = lat + ((n + 15) & ~15);  // [n][d] when emb_in_lds
    const float* W = a.w_in_lds ? lw : a.wt;
    if (a.w_in_lds)
        for (int i = tid; i < 2 * d * GS; i += 256) lw[i] = a.wt[i];
    if (a.emb_in_lds) {
        const float* eg = a.emb + (long)b * n * a.lde;
        for (int i = tid; i < n * d; i += 256) le[i] = eg[(long)(i / d) * a.lde + i % d];
    }
    // ---- output layer: dpre = dout * (out > 0);  [dh, dr] = Wp^T dpre;  dc = 0
    for (int j = tid; j < d; j += 256) {
        const float o = a.out[(long)b * d + j];
        const float v = o > 0.f ? a.dout[(long)b * d + j] : 0.f;
        dg[j] = v;                                   // borrow dg[0:d] for dpre
        a.DPRE[(long)b * d + j] = v;
        dc[j] = 0.f;
    }
    lds_barrier();
    for (int k = tid; k < 2 * d; k += 256) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) s += a.Wp[(long)j * 2 * d + k] * dg[j];
        if (k < d) dh[k] = s; else dr[k - d] = s;
    }
    lds_barrier();
    const float* emb = a.emb_in_lds ? le : a.emb + (long)b * n * a.lde;
    const int lde = a.emb_in_lds ? d : a.lde;
    const int tl = tid & 15, team = tid >> 4;

    // The saved per-step state (a_t, gates, c_t, c_{t-1}) is read from global memory: fetched one step AHEAD into
    // registers so the loads fly under the previous step instead of heading each step's critical path.
    constexpr int APT = 4;                            // a_t rows per thread: n <= 1024 (checked by the launcher)
    float at_n[APT], gn[6];
    auto prefetch = [&](int t) {
        const int tc = t < 0 ? 0 : t;                 // (the last prefetch is unused; keep the address valid)
        const float* at = a.Aw + ((long)b * n + tc) * n;
#pragma unroll
        for (int i = 0; i < APT; ++i) at_n[i] = at[min(tid + 256 * i, n - 1)];
        const int j = min(tid, d - 1);
        const float* gs = a.G + ((long)b * n + tc) * 4 * d;
        gn[0] = gs[j]; gn[1] = gs[d + j]; gn[2] = gs[2 * d + j]; gn[3] = gs[3 * d + j];
        gn[4] = a.Cs[((long)b * n + tc) * d + j];
        gn[5] = tc > 0 ? a.Cs[((long)b * n + tc - 1) * d + j] : 0.f;
    };
    prefetch(n - 1);
    for (int t = n - 1; t >= 0; --t) {
        float at_c[APT], gc[6];
#pragma unroll
        for (int i = 0; i < APT; ++i) at_c[i] = at_n[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) gc[i] = gn[i];
#pragma unroll
        for (int i = 0; i < APT; ++i)
            if (tid + 256 * i < n) lat[tid + 256 * i] = at_c[i];
        // ---- r_t = sum a emb:  da = emb . dr ;  de = a * (da - sum a da)
        for (int j = tid; j < d; j += 256) a.DR[((long)b * n + t) * d + j] = dr[j];
        lds_barrier();
        prefetch(t - 1);
        float lsum = 0.f;
        for (int row = team; row < n; row += 16) {
            const float* er = emb + (long)row * lde;
            float s = 0.f;
            for (int k = tl; k < d; k += 16) s += er[k] * dr[k];
            s = team16_sum(s);
            if (tl == 0) {
                de[row] = s;
                lsum += lat[row] * s;
            }
        }
        const float sdot = block_sum(lsum, red);
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int row = tid + 256 * i;
            if (row < n) {
                const float v = at_c[i] * (de[row] - sdot);
                de[row] = v;
                a.DE[((long)b * n + t) * n + row] = v;
            }
        }
        lds_barrier();
        // ---- e = emb . h_t:  dh += sum_n de[n] emb[n]
        for (int j0 = 0; j0 < d; j0 += 64) {
            const int j = j0 + (tid & 63), part = tid >> 6;
            float s = 0.f;
            if (j < d) {
#pragma unroll 8
                for (int row = part; row < n; row += 4) s += de[row] * emb[(long)row * lde + j];
            }
            rpart[part * 64 + (tid & 63)] = s;
            lds_barrier();
            if (part == 0 && j < d) dh[j] += rpart[tid] + rpart[64 + tid] + rpart[128 + tid] + rpart[192 + tid];
            lds_barrier();
        }
        // ---- LSTM cell backward
        if (tid < d) {
            const int j = tid;
            const float ig = gc[0], fg = gc[1], gg = gc[2], og = gc[3];
            const float ct = gc[4];
            const float cp = gc[5];
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
        lds_barrier();
        // ---- [dh_{t-1}, dr_{t-1}] = Wt dg   (h_{t-1} and r_{t-1} feed only this step's LSTM)
        for (int k = tid; k < 2 * d; k += 256) {
            float s = 0.f, s2 = 0.f;
            const float* wr = W + (long)k * GS;
#pragma unroll 4
            for (int g = 0; g < 2 * d; ++g) {
                s += wr[g] * dg[g];
                s2 += wr[2 * d + g] * dg[2 * d + g];
            }
            s += s2;
            if (k < d) dh[k] = s; else dr[k - d] = s;
        }
        lds_barrier();
    }
}

size_t s2s_dyn_lds(int n, int d, bool w_in_lds, bool emb_in_lds = false) {
    const size_t GS = 4 * d + 1;
    return ((w_in_lds ? (size_t)2 * d * GS : 0) + 7 * d + 8 + 256 + 2 * ((n + 15) & ~15) + 16 +
            (emb_in_lds ? (size_t)n * d : 0)) * sizeof(float);
}
bool s2s_w_fits(int n, int d) { return s2s_dyn_lds(n, d, true) <= 158 * 1024; }
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
    a.w_in_lds = s2s_w_fits(n, d) ? 1 : 0;
    a.emb_in_lds = s2s_emb_fits(n, d) ? 1 : 0;
    const size_t ldsb = s2s_dyn_lds(n, d, a.w_in_lds, a.emb_in_lds);
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_set2set_fwd), 160 * 1024, "k_set2set_fwd");
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_set2set_fwd, dim3(B), dim3(256), ldsb, q.stream, a);
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
    a.w_in_lds = s2s_w_fits(n, d) ? 1 : 0;
    a.emb_in_lds = s2s_emb_fits(n, d) ? 1 : 0;
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_set2set_bwd), 160 * 1024, "k_set2set_bwd");
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_set2set_bwd, dim3(B), dim3(256), s2s_dyn_lds(n, d, a.w_in_lds, a.emb_in_lds), q.stream, a);
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

}  // namespace dp
