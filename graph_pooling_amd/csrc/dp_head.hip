// The prediction head of the encoders in two launches instead of ten:
//   forward : last-level max readout (encoders.py:1069-1071 / 1296-1298) + pred_model, the Linear/ReLU stack of
//             build_pred_layers (encoders.py:1027-1041) — per graph independent, one workgroup per graph;
//   backward: the whole pred_model backward (weight, bias and input gradients) + the max-readout scatter of every
//             level — one workgroup for the batch, so the weight gradients are plain deterministic sums over b.
// At B = 20 these were 3 + 7 kernels of 4-6 us each, all of them at the launch floor.
#include "dp_common.h"

namespace dp {

#ifdef DP_STAMP
// diagnostic build only: shader-clock stamps of workgroup 0 ([0,8) forward, [8,16) backward), tools/head_stamps.py
__device__ unsigned long long g_head_stamps[16];
#define HEAD_STAMP(i)                                                                                 \
    do {                                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_head_stamps[i] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define HEAD_STAMP(i) \
    do {              \
    } while (0)
#endif

constexpr int HEAD_FWD_LDS_FLOATS = 30 * 1024;    // weights + biases + two activation vectors + readout scratch
constexpr int HEAD_BWD_LDS_FLOATS = 38 * 1024;    // weights + saved activations of the batch + two gradient blocks

__device__ inline float head_wave_sum(float v) {
    return wave64_sum(v);
}

// dst[i] = src[i] for i < cnt, 16 loads per thread in flight (clamped addresses, no branch around the loads)
template <int NT>
__device__ inline void head_stage(float* dst, const float* src, int cnt) {
    for (int e0 = 0; e0 < cnt; e0 += NT * 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = src[min(e0 + u * NT + (int)threadIdx.x, cnt - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = e0 + u * NT + (int)threadIdx.x;
            if (i < cnt) dst[i] = t[u];
        }
    }
}

// Split form of head_stage: `issue` asks for round `e0` of a segment (16 elements per thread, clamped addresses, gather
// through idx), `commit` writes it to LDS.  Several segments are issued before the first commit, so their loads share
// one memory round trip; segments longer than one round finish with stage_rest.
template <int NT, typename Idx>
__device__ __forceinline__ void stage_issue(float (&t)[16], const float* src, int cnt, int e0, Idx idx) {
    // only the slots the segment has (uniform early exit): most segments are a few hundred floats, and sixteen clamped
    // loads per thread for each of them made the staging instruction-bound (13 k cycles for ~2 300 floats)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (e0 + u * NT >= cnt) break;
        t[u] = src[idx(min(e0 + u * NT + (int)threadIdx.x, cnt - 1))];
    }
}
template <int NT>
__device__ __forceinline__ void stage_commit(float* dst, const float (&t)[16], int cnt, int e0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (e0 + u * NT >= cnt) break;
        const int i = e0 + u * NT + (int)threadIdx.x;
        if (i < cnt) dst[i] = t[u];
    }
}
template <int NT, typename Idx>
__device__ inline void stage_rest(float* dst, const float* src, int cnt, Idx idx) {
    for (int e0 = NT * 16; e0 < cnt; e0 += NT * 16) {
        float t[16];
        stage_issue<NT>(t, src, cnt, e0, idx);
        stage_commit<NT>(dst, t, cnt, e0);
    }
}

struct HeadSizes {
    int wtot, btot, mx, hsum;      // weight floats, bias floats, widest layer, sum of dims[0..L]
};
__host__ __device__ inline HeadSizes head_sizes(const HeadArgs& a) {
    HeadSizes z{0, 0, 0, 0};
    for (int i = 0; i < a.n_pred; ++i) {
        z.wtot += a.dims[i] * a.dims[i + 1];
        z.btot += a.dims[i + 1];
    }
    for (int i = 0; i <= a.n_pred; ++i) {
        z.mx = a.dims[i] > z.mx ? a.dims[i] : z.mx;
        z.hsum += a.dims[i];
    }
    return z;
}

// Both kernels are latency problems, not throughput problems (a 120-50-2 MLP on 20 graphs): every global operand is
// requested up front in bulk, coalesced loops into LDS, and all arithmetic then runs out of LDS.

// ------------------------------------------------------------------ forward  (one workgroup per graph)
__global__ __launch_bounds__(256) void k_head_fwd(HeadArgs a) {
    extern __shared__ float lds[];
    const HeadSizes z = head_sizes(a);
    float* Wl = lds;                       // all layers' weights, layer after layer
    float* bl = Wl + z.wtot;               // all biases (zeros where a layer has none)
    float* h0 = bl + z.btot;               // activations, ping
    float* h1 = h0 + z.mx;                 // pong
    float* sv = h1 + z.mx;                 // readout partials [4][64] values
    int* si = reinterpret_cast<int*>(sv + 256);
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* feat = a.hid[0] + (long)b * a.dims[0];
    HEAD_STAMP(0);
    // (1) readout loads go out first (16 rows per thread in flight: one pass for a level of <= 64 nodes, clamped addresses, no branches) ...
    constexpr int RU = 16;
    float zv[RU];
    const bool one_pass = a.Z && a.rw <= 64 && a.n <= 4 * RU;
    if (one_pass) {
        const float* zc = a.Z + (long)b * a.n * a.ldz + min(lane, a.rw - 1);
#pragma unroll
        for (int u = 0; u < RU; ++u) zv[u] = zc[(long)min(wave + 4 * u, a.n - 1) * a.ldz];
    }
    // (2) ... then the weights, biases and lower-level features stream into LDS behind them
    //     The first round of layer i+1 (and of the features) is in flight while layer i's is committed, so the
    //     staging is about as many round trips as the longest segment has rounds, not their sum.
    {
        const auto ident = [](int e) { return e; };
        float tf[16], tw[16], tb[16];
        stage_issue<256>(tf, feat, a.dims[0], 0, ident);
        int wo = 0, bo = 0;
        {
            const int cnt0 = a.dims[0] * a.dims[1];
            stage_issue<256>(tw, a.params + a.w_off[0], cnt0, 0, ident);
        }
        for (int i = 0; i < a.n_pred; ++i) {
            const int cnt = a.dims[i] * a.dims[i + 1], bc = a.dims[i + 1];
            const float* W = a.params + a.w_off[i];
            const float* bsrc = a.b_off[i] >= 0 ? a.params + a.b_off[i] : W;      // a valid address either way
            stage_issue<256>(tb, bsrc, bc, 0, ident);
            float tn[16];                                                          // next layer's first round
            const bool more = i + 1 < a.n_pred;
            if (more) stage_issue<256>(tn, a.params + a.w_off[i + 1], a.dims[i + 1] * a.dims[i + 2], 0, ident);
            stage_commit<256>(Wl + wo, tw, cnt, 0);
            if (a.b_off[i] < 0) {
#pragma unroll
                for (int u = 0; u < 16; ++u) tb[u] = 0.f;
            }
            stage_commit<256>(bl + bo, tb, bc, 0);
            stage_rest<256>(Wl + wo, W, cnt, ident);
            if (a.b_off[i] >= 0) stage_rest<256>(bl + bo, bsrc, bc, ident);
            else
                for (int e = 4096 + threadIdx.x; e < bc; e += 256) bl[bo + e] = 0.f;
            if (more) {
#pragma unroll
                for (int u = 0; u < 16; ++u) tw[u] = tn[u];
            }
            wo += cnt;
            bo += bc;
        }
        // features of the lower levels (the readout columns of this level are produced below)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = u * 256 + (int)threadIdx.x;
            if (k < a.dims[0] && (!a.Z || k < a.featoff || k >= a.featoff + a.rw)) h0[k] = tf[u];
        }
        for (int k = 4096 + threadIdx.x; k < a.dims[0]; k += 256)
            if (!a.Z || k < a.featoff || k >= a.featoff + a.rw) h0[k] = feat[k];
    }
    HEAD_STAMP(1);
    if (a.Z) {
        // unmasked max over the nodes of the last level; ties -> lowest row (torch CPU max).  Four row groups x 64
        // columns per pass, combined through LDS.
        for (int c0 = 0; c0 < a.rw; c0 += 64) {
            const int c = c0 + lane;
            float best = -INFINITY;
            int bi = -1;
            if (one_pass) {
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int r = wave + 4 * u;
                    if (r < a.n && zv[u] > best) {
                        best = zv[u];
                        bi = r;
                    }
                }
            } else if (c < a.rw) {
                const float* zc = a.Z + (long)b * a.n * a.ldz + c;
#pragma unroll 4
                for (int r = wave; r < a.n; r += 4) {
                    const float v = zc[(long)r * a.ldz];
                    if (v > best) {
                        best = v;
                        bi = r;
                    }
                }
            }
            sv[wave * 64 + lane] = best;
            si[wave * 64 + lane] = bi;
            __syncthreads();
            if (wave == 0 && c < a.rw) {
                for (int k = 1; k < 4; ++k) {
                    const float v = sv[k * 64 + lane];
                    const int i2 = si[k * 64 + lane];
                    if (i2 >= 0 && (v > best || (v == best && i2 < bi))) {
                        best = v;
                        bi = i2;
                    }
                }
                h0[a.featoff + c] = best;
                feat[a.featoff + c] = best;
                a.argmax[(long)b * a.lda + c] = bi;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    HEAD_STAMP(2);
    float* cur = h0;
    float* nxt = h1;
    int wo = 0, bo = 0;
    bool poisoned = false;
    for (int i = 0; i < a.n_poison; ++i)
        poisoned |= __hip_atomic_load(a.poison[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (int i = 0; i < a.n_pred; ++i) {
        const int din = a.dims[i], dout = a.dims[i + 1];
        const bool last = i == a.n_pred - 1;
        float* out = a.hid[i + 1] + (long)b * dout;
        // lane = output (64 per pass), the four waves split k; partial sums meet in LDS.  (A wave-per-output dot
        // product costs a 6-step cross-lane reduction per output: 13 serial ones for the 50-wide hidden layer.)
        for (int j0 = 0; j0 < dout; j0 += 64) {
            const int j = j0 + lane;
            const int kq = (din + 3) / 4, kb = wave * kq, ke = min(din, kb + kq);
            float s = 0.f;
            if (j < dout) {
                const float* w = Wl + wo + j * din;
#pragma unroll 8
                for (int k = kb; k < ke; ++k) s += cur[k] * w[k];
            }
            sv[wave * 64 + lane] = s;
            __syncthreads();
            if (wave == 0 && j < dout) {
                s = sv[lane] + sv[64 + lane] + sv[128 + lane] + sv[192 + lane] + bl[bo + j];
                if (!last) s = fmaxf(s, 0.f);
                else if (poisoned) s = __builtin_nanf("");
                nxt[j] = s;
                out[j] = s;
            }
            __syncthreads();
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
        wo += din * dout;
        bo += dout;
    }
    // evaluation: the predicted class leaves the device as ONE integer per graph (`torch.max(ypred, 1)` +
    // `.cpu()` of B x C logits in the reference's evaluate(), train.py:42-44); first index on ties
    if (a.labels && threadIdx.x == 0) {
        const int C = a.dims[a.n_pred];
        int best = 0;
        for (int c = 1; c < C; ++c)
            if (cur[c] > cur[best]) best = c;
        a.labels[b] = best;
    }
    HEAD_STAMP(3);
}

static size_t head_fwd_lds_floats(const HeadArgs& a) {
    const HeadSizes z = head_sizes(a);
    return (size_t)z.wtot + z.btot + 2 * z.mx + 512;
}
void head_fwd(Seq& q, const HeadArgs& a) {
    if (!q.ok()) return;
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_head_fwd), HEAD_FWD_LDS_FLOATS * (int)sizeof(float),
                   "k_head_fwd");
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_head_fwd, dim3(a.B), dim3(256), head_fwd_lds_floats(a) * sizeof(float), q.stream, a);
    q.check_launch("head_fwd");
}

// ------------------------------------------------------------------ backward
// Workgroup w owns a slice of HEAD_SLICE input columns of the FIRST Linear (the wide one: 120 readout features against
// 50 and 2 above it): its weight-gradient columns, its share of d(features) and the max-readout scatter of those
// features.  The layers above are a few hundred MACs per graph; every workgroup recomputes them from LDS instead of
// waiting for a neighbour (no inter-workgroup dependency), and workgroup 0 stores their gradients.  All sums over the
// batch are plain loops in b: deterministic.
constexpr int HEAD_SLICE = 16;
__global__ __launch_bounds__(256) void k_head_bwd(HeadBwdArgs a) {
    extern __shared__ float lds[];
    const int B = a.h.B, L = a.h.n_pred;
    const int d0 = a.h.dims[0], d1 = a.h.dims[1];
    const int k0 = blockIdx.x * HEAD_SLICE, ks = min(HEAD_SLICE, d0 - k0);
    int wup = 0, hup = 0, mx = 0;                 // floats of the upper layers' weights / saved inputs; widest upper dim
    for (int i = 1; i < L; ++i) {
        wup += a.h.dims[i] * a.h.dims[i + 1];
        hup += a.h.dims[i];
    }
    for (int i = 1; i <= L; ++i) mx = max(mx, a.h.dims[i]);
    float* Wu = lds;                              // upper weights, layer 1 first
    float* Hu = Wu + wup;                         // saved inputs of layers 1..L-1, [B x dims[i]] each
    float* W0 = Hu + (long)B * hup;               // [d1][ks] slice of the first layer's weight
    float* F0 = W0 + d1 * HEAD_SLICE;             // [B][ks] slice of the features
    float* g0 = F0 + B * HEAD_SLICE;              // gradient ping / pong [B x mx]
    float* g1 = g0 + (long)B * mx;
    HEAD_STAMP(8);
    int win[2] = {-1, -1};                        // max-readout winner rows of this thread's first two features
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = min((int)threadIdx.x + 256 * it, B * ks - 1);
        const int b = e / ks, k = k0 + e % ks;
        if (a.n_levels > 0) {                     // uniform; the load itself is unconditional (no branch, no select:
            const int* src = a.lv[0].argmax;      // a feature outside every level reads a valid dummy it never uses)
            for (int lv = 0; lv < a.n_levels; ++lv) {
                const int f = k - a.lv[lv].featoff;
                if (f >= 0 && f < a.lv[lv].rw) src = a.lv[lv].argmax + (long)b * a.lv[lv].lda + f;
            }
            win[it] = *src;
        }
    }
    {
        // first-layer slice, feature slice and incoming gradient: asked for first, committed after the upper layers'
        // operands, so the whole staging is about two memory round trips (it was eight: one per loop)
        const auto ident = [](int e) { return e; };
        const auto slice = [=](int e) { return (long)(e >> 4) * d0 + k0 + min(e & 15, ks - 1); };   // [rows][HEAD_SLICE]
        static_assert(HEAD_SLICE == 16, "slice index math");
        const float* W0g = a.h.params + a.h.w_off[0];
        const float* F0g = a.h.hid[0];
        const int cW0 = d1 * HEAD_SLICE, cF0 = B * HEAD_SLICE, cG = B * a.h.dims[L];
        float tw0[16], tf0[16], tg[16];
        stage_issue<256>(tw0, W0g, cW0, 0, slice);
        stage_issue<256>(tf0, F0g, cF0, 0, slice);
        stage_issue<256>(tg, a.d_ypred, cG, 0, ident);
        int wo = 0, ho = 0;
        for (int i = 1; i < L; ++i) {
            const int cnt = a.h.dims[i] * a.h.dims[i + 1], hc = B * a.h.dims[i];
            const float* W = a.h.params + a.h.w_off[i];
            float tw[16], th[16];
            stage_issue<256>(tw, W, cnt, 0, ident);
            stage_issue<256>(th, a.h.hid[i], hc, 0, ident);
            stage_commit<256>(Wu + wo, tw, cnt, 0);
            stage_commit<256>(Hu + ho, th, hc, 0);
            stage_rest<256>(Wu + wo, W, cnt, ident);
            stage_rest<256>(Hu + ho, a.h.hid[i], hc, ident);
            wo += cnt;
            ho += hc;
        }
        stage_commit<256>(W0, tw0, cW0, 0);
        stage_commit<256>(F0, tf0, cF0, 0);
        stage_commit<256>(g0, tg, cG, 0);
        stage_rest<256>(W0, W0g, cW0, slice);
        stage_rest<256>(F0, F0g, cF0, slice);
        stage_rest<256>(g0, a.d_ypred, cG, ident);
    }
    __syncthreads();
    HEAD_STAMP(9);
    float* go = g0;
    float* gi = g1;
    int wo = wup, ho = B * hup;
    for (int i = L - 1; i >= 1; --i) {
        const int din = a.h.dims[i], dout = a.h.dims[i + 1];
        wo -= din * dout;
        ho -= B * din;
        const float* W = Wu + wo;
        const float* hin = Hu + ho;
        if (blockIdx.x == 0) {
            float* dW = a.grads + a.h.w_off[i];
            for (int e = threadIdx.x; e < dout * din; e += 256) {
                const int j = e / din, k = e % din;
                float s = 0.f;
#pragma unroll 8
                for (int b = 0; b < B; ++b) s += go[b * dout + j] * hin[b * din + k];
                dW[e] = s;
            }
            if (a.h.b_off[i] >= 0)
                for (int j = threadIdx.x; j < dout; j += 256) {
                    float s = 0.f;
#pragma unroll 8
                    for (int b = 0; b < B; ++b) s += go[b * dout + j];
                    a.grads[a.h.b_off[i] + j] = s;
                }
        }
        // gi[b][k] = relu'(hin) * sum_j go[b][j] W[j][k]   (hin = relu(.), so hin > 0 decides)
        for (int e = threadIdx.x; e < B * din; e += 256) {
            const int b = e / din, k = e % din;
            float s = 0.f;
#pragma unroll 8
            for (int j = 0; j < dout; ++j) s += go[b * dout + j] * W[j * din + k];
            gi[e] = hin[e] > 0.f ? s : 0.f;
        }
        __syncthreads();
        float* t = go; go = gi; gi = t;
    }
    HEAD_STAMP(10);
    // first layer, this workgroup's columns: go is [B x d1]
    float* dW = a.grads + a.h.w_off[0];
    for (int e = threadIdx.x; e < d1 * ks; e += 256) {
        const int j = e / ks, kk = e % ks;
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s += go[b * d1 + j] * F0[b * HEAD_SLICE + kk];
        dW[(long)j * d0 + k0 + kk] = s;
    }
    if (blockIdx.x == 0 && a.h.b_off[0] >= 0)
        for (int j = threadIdx.x; j < d1; j += 256) {
            float s = 0.f;
#pragma unroll 8
            for (int b = 0; b < B; ++b) s += go[b * d1 + j];
            a.grads[a.h.b_off[0] + j] = s;
        }
    // d(features) of the slice and its max-readout scatter: dZ[b, argmax, f] += dfeat  (dZ zero-initialised by the caller)
    HEAD_STAMP(11);
    // (a feature belongs to one level; its winner row was asked for at kernel start; dZ is zero on entry -- the caller's
    // contract -- and every (graph, row, feature) is written by one thread, so the scatter is a plain store)
    for (int e = threadIdx.x, it = 0; e < B * ks; e += 256, ++it) {
        const int b = e / ks, kk = e % ks, k = k0 + kk;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < d1; ++j) s += go[b * d1 + j] * W0[j * HEAD_SLICE + kk];
        for (int lv = 0; lv < a.n_levels; ++lv) {
            const HeadBwdArgs::Level& t = a.lv[lv];
            const int f = k - t.featoff;
            if (f >= 0 && f < t.rw) {
                const int r = it == 0 ? win[0] : it == 1 ? win[1] : t.argmax[(long)b * t.lda + f];
                if (r >= 0) t.dZ[((long)b * t.n + r) * t.ldz + f] = s;
            }
        }
    }
    HEAD_STAMP(12);
}

static size_t head_bwd_lds_floats(const HeadArgs& a) {
    size_t wup = 0, hup = 0, mx = 0;
    for (int i = 1; i < a.n_pred; ++i) {
        wup += (size_t)a.dims[i] * a.dims[i + 1];
        hup += a.dims[i];
    }
    for (int i = 1; i <= a.n_pred; ++i) mx = (size_t)a.dims[i] > mx ? a.dims[i] : mx;
    return wup + (size_t)a.B * hup + (size_t)a.dims[1] * HEAD_SLICE + (size_t)a.B * HEAD_SLICE + 2 * (size_t)a.B * mx;
}

bool head_supported(const HeadArgs& a) {
    if (a.n_pred < 1 || a.B < 1 || a.B > 1024) return false;
    // everything the kernels touch repeatedly must fit LDS (they are latency-, not throughput-shaped)
    return head_fwd_lds_floats(a) <= HEAD_FWD_LDS_FLOATS && head_bwd_lds_floats(a) <= HEAD_BWD_LDS_FLOATS;
}

void head_bwd(Seq& q, const HeadBwdArgs& a) {
    if (!q.ok()) return;
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_head_bwd), HEAD_BWD_LDS_FLOATS * (int)sizeof(float),
                   "k_head_bwd");
    if (!q.ok()) return;
    const int slices = (a.h.dims[0] + HEAD_SLICE - 1) / HEAD_SLICE;
    hipLaunchKernelGGL(k_head_bwd, dim3(slices), dim3(256), head_bwd_lds_floats(a.h) * sizeof(float), q.stream, a);
    q.check_launch("head_bwd");
}

#ifdef DP_STAMP
extern "C" __attribute__((visibility("default"))) int dp_debug_head_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_head_stamps), sizeof(unsigned long long) * 16);
}
#endif

}  // namespace dp
