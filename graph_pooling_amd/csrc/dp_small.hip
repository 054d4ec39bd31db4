// GCN layers of a POOLED level (n = K_j <= 64 nodes per graph, e.g. 50 at the DD shape, 10 at ENZYMES):
// gcn_forward over (X', A') after the pooling step (encoders.py:1282-1284, 1054-1081) and its backward.
// The generic path spends ~13 forward + ~25 backward launches of 4-8 us on a level whose whole per-graph
// state (A' 50x50, X' 50x60, W 60x20) is a few tens of KB.  Here ONE workgroup per graph runs a complete
// layer out of LDS — transform, aggregation, bias, l2-normalise, BN statistics forward;  BN/ReLU/normalise
// backward, bias sums, A'^T dU, dW, dX, dA' backward — so a layer is one launch each way (the cross-graph
// BatchNorm statistics still force one grid-wide boundary per layer; they travel as per-row partials and each
// workgroup combines the B partials of its node indices itself).
// Plain fp32 FMA loops: ~0.1 MFLOP per graph per layer, latency-bound whatever the ALU.
#include "dp_common.h"

namespace dp {

#define SM_L2_EPS 1e-12f
#define SM_BN_EPS 1e-5f

#ifdef DP_STAMP
// Diagnostic build only (csrc/build.sh with DP_STAMP=1 writes a separate library): shader-clock stamps of the phases of
// workgroup 0, read back with dp_debug_stamps().  No stamp executes in the product build.
__device__ unsigned long long g_small_stamps[2][32];
#define SM_STAMP(k, i)                                                                       \
    do {                                                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_small_stamps[k][i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SM_STAMP(k, i) \
    do {               \
    } while (0)
#endif

namespace {

__device__ inline float sm_team_sum(float v) {
    return row16_sum(v);
}

typedef float sm_f32x4 __attribute__((ext_vector_type(4)));

// Bulk staging by LDS-DMA: dst[e] <- *src_of(e) for e < count, 64 dwords per wave instruction, no register round
// trip; every request of every region is in flight before the single barrier that follows (whose fence drains
// vmcnt).  The LDS destination is wave-uniform (M0) + 4 * lane.
// In-kernel stamps (DP_STAMP build, tools/small_kernel_stamps.py) put the opening burst at ~10k of the backward
// kernel's ~29k cycles and ~7.5k of the forward's ~15k whichever way it is written — plain load/store loops, this
// DMA form, or registers with all loads issued first: one workgroup pulls ~180 KB through ONE CU's vector-memory
// path, and 16 waves issuing DMAs at once pay ~400 cycles per instruction.  It is the structural floor of the
// one-workgroup-per-graph design, not a scheduling accident.
// e / d for e < 2^16, d <= 2^15 with a host-made reciprocal (a runtime integer division is ~40 VALU instructions, and
// the staging issues one per element: it was the larger half of the staging time)
struct SmDiv {
    unsigned magic;
    int d;
    __device__ inline int quot(int e) const {
        return d == 1 ? e : (int)(((unsigned long long)(unsigned)e * magic) >> 32);    // (2^32 / 1 + 1 does not fit)
    }
    __device__ inline int rem(int e) const { return e - quot(e) * d; }
};
static SmDiv sm_div(int d) { return SmDiv{(unsigned)(0x100000000ull / (unsigned)d + 1), d}; }

template <typename SrcFn>
__device__ inline void sm_dma(float* dst, int count, SrcFn src_of) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    for (int c0 = wave * 64; c0 < count; c0 += nw * 64) {
        const int e = c0 + lane;
        if (e < count)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_of(e),
                                             (__attribute__((address_space(3))) void*)(dst + c0), 4, 0, 0);
    }
}

// C[M x N] = op(A)[M x K] · op(B)[K x N] with both operands in LDS, on the fp32 MFMA (16x16x4), tiles dealt
// round-robin to the workgroup's waves.  Reads outside the logical matrices return 0, so nothing needs
// padding.  `store(i, j, v)` receives every in-range result element.
template <bool TA, bool TB, typename Store>
__device__ inline void lds_mma(const float* A, int lda, const float* B, int ldb, int M, int N, int K, Store store,
                               int wave_shift = 0) {
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    // `wave_shift` rotates the tile -> wave assignment so that two independent products issued back to back (8 tiles
    // each at the DD shape, 16 waves) run side by side instead of both landing on waves 0-7
    const int wave = ((threadIdx.x >> 6) + nwaves - wave_shift % nwaves) % nwaves;
    const int l15 = lane & 15, kq = lane >> 4;
    const int tm = (M + 15) / 16, tn = (N + 15) / 16;
    for (int t = wave; t < tm * tn; t += nwaves) {
        const int i = (t / tn) * 16 + l15, j = (t % tn) * 16 + l15;
        sm_f32x4 acc = (sm_f32x4){0.f, 0.f, 0.f, 0.f};
        // unconditional LDS reads on clamped indices + a select: a branch around each read makes the compiler wait
        // for every one of them in turn.  (Reading 16 steps' fragments into registers first and then issuing the
        // MFMAs back to back was tried: slower, 3.1k against 2.1k cycles per tile.)
        const int ic = min(i, M - 1), jc = min(j, N - 1);
        const bool iok = i < M, jok = j < N;
#pragma unroll 8
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + kq, kc = min(k, K - 1);
            const float ar = TA ? A[kc * lda + ic] : A[ic * lda + kc];
            const float br = TB ? B[jc * ldb + kc] : B[kc * ldb + jc];
            const float av = (k < K && iok) ? ar : 0.f;
            const float bv = (k < K && jok) ? br : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (t / tn) * 16 + kq * 4 + r;
            if (row < M && j < N) store(row, j, acc[r]);
        }
    }
}


// Same contract as lds_mma, shaped for instruction ISSUE: with 16 waves on four SIMDs (only the waves that own a
// tile are busy) a product is bound by the ~200 vector instructions of address arithmetic per tile — clamp, multiply,
// shift-add and a select per fragment element, ~15 cycles each in dependent chains — not by LDS or MFMA time
// (in-kernel stamps: 3.0k cycles for ONE 16x16 tile with K = 50).  Here a fragment address is a running pointer: one
// add per operand and k-step; full k-steps carry no clamp and no select (rows / columns past M / N are clamped once
// per tile: their products land in output elements nobody stores); only the one partial k-step at the end of K is
// masked.  Batches of four k-steps have their eight LDS reads in flight before the first MFMA; two accumulators.
template <bool TA, bool TB, typename Store>
__device__ inline void lds_mma2(const float* A, int lda, const float* B, int ldb, int M, int N, int K, Store store,
                                int wave_shift = 0) {
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = ((threadIdx.x >> 6) + nwaves - wave_shift % nwaves) % nwaves;
    const int l15 = lane & 15, kq = lane >> 4;
    const int tm = (M + 15) / 16, tn = (N + 15) / 16;
    const int sa4 = 4 * (TA ? lda : 1), sb4 = 4 * (TB ? 1 : ldb);     // element stride of one k-step
    int tr = wave / tn, tc = wave - tr * tn;                           // (tile row, tile column), advanced by nwaves
    const int dr = nwaves / tn, dc = nwaves - dr * tn;
    for (; tr < tm; ) {
        const int i = tr * 16 + l15, j = tc * 16 + l15;
        const float* ap = (TA ? A + min(i, M - 1) : A + min(i, M - 1) * lda) + kq * (sa4 >> 2);
        const float* bp = (TB ? B + min(j, N - 1) * ldb : B + min(j, N - 1)) + kq * (sb4 >> 2);
        sm_f32x4 acc0 = (sm_f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        int k = 0;
        for (; k + 16 <= K; k += 16) {
            const float a0 = ap[0], a1 = ap[sa4], a2 = ap[2 * sa4], a3 = ap[3 * sa4];
            const float b0 = bp[0], b1 = bp[sb4], b2 = bp[2 * sb4], b3 = bp[3 * sb4];
            ap += 4 * sa4;
            bp += 4 * sb4;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
        }
        for (; k + 4 <= K; k += 4) {
            const float a0 = ap[0], b0 = bp[0];
            ap += sa4;
            bp += sb4;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        }
        if (k < K) {                        // partial k-step: lanes with k >= K contribute 0 (the load is discarded)
            const bool kin = k + kq < K;
            const float a0 = ap[0], b0 = bp[0];
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kin ? a0 : 0.f, kin ? b0 : 0.f, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tr * 16 + kq * 4 + r;
            if (row < M && j < N) store(row, j, acc0[r] + acc1[r]);
        }
        tr += dr;
        tc += dc;
        if (tc >= tn) {
            tc -= tn;
            ++tr;
        }
    }
}

struct SmallFwdArgs {
    const float* adj;      // [B, n, n]
    const float* x0;       // layer 0: raw input [B, n, din] (ld = ldx0); else null
    int ldx0;
    const float* yprev;    // layer > 0: previous layer's normalised pre-ReLU output [B, n, din] (ld = ldyp)
    int ldyp;
    const float* part_prev;  // [B, n, 2] (row mean, row M2) of relu(yprev), or null (no BN)
    float* stats_prev;       // [n, 2] out (mu, rstd) of the previous layer
    float* xout;           // layer > 0: BN output of the previous layer -> concat buffer slice (ld = ldxo)
    int ldxo;
    const float* W;        // [din, dout]
    const float* bias;     // [dout] or null
    float* y;              // [B, n, dout] out (ld = ldy)
    int ldy;
    float* invn;           // [B, n]
    float* part;           // [B, n, 2] out or null
    int B, n, din, dout;
    int add_self, stats;
    SmDiv qdin;
};

// Every global input is staged into LDS in ONE burst (one round of global latency); all phases after the
// first barrier read LDS only.
__global__ __launch_bounds__(1024) void k_small_gcn_fwd(SmallFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, din = a.din, dout = a.dout;
    float* A = lds;                    // [n][n]
    float* X = A + n * n;              // [n][din]
    float* W = X + n * din;            // [din][dout]
    float* P = W + din * dout;         // [n][dout]
    float* U = P + n * dout;           // [n][dout]
    float* mu = U + n * dout;          // [n]
    float* rs = mu + n;                // [n]
    float* BI = rs + n;                // [dout] bias (zeros without one)
    float* PP = BI + dout;             // [B][n][2] staged BN partials of the previous layer
    const int tl = tid & 15, team = tid >> 4;
    const int NT = blockDim.x, NTEAMS = blockDim.x >> 4;
    const bool bnprev = !a.x0 && a.part_prev;

    SM_STAMP(0, 0);
    sm_dma(A, n * n, [&](int e) { return a.adj + (long)b * n * n + e; });
    sm_dma(W, din * dout, [&](int e) { return a.W + e; });
    if (a.bias) sm_dma(BI, dout, [&](int e) { return a.bias + e; });
    else
        for (int i = tid; i < dout; i += NT) BI[i] = 0.f;
    {
        const float* xsrc = a.x0 ? a.x0 : a.yprev;
        const int ldx = a.x0 ? a.ldx0 : a.ldyp;
        sm_dma(X, n * din, [&](int e) { return xsrc + ((long)b * n + a.qdin.quot(e)) * ldx + a.qdin.rem(e); });
        if (bnprev) sm_dma(PP, a.B * n * 2, [&](int e) { return a.part_prev + e; });
    }
    __syncthreads();
    if (!a.x0) {
        // BN statistics of the previous layer per node index (Chan-combine of the B row partials)
        for (int r = tid; r < n; r += NT) {
            float m = 0.f, rstd = 1.f;
            if (bnprev) {
                float sm = 0.f;
#pragma unroll 4
                for (int bb = 0; bb < a.B; ++bb) sm += PP[(bb * n + r) * 2];
                m = sm / (float)a.B;
                float s2 = 0.f;
#pragma unroll 4
                for (int bb = 0; bb < a.B; ++bb) {
                    const float d = PP[(bb * n + r) * 2] - m;
                    s2 += PP[(bb * n + r) * 2 + 1] + (float)din * d * d;
                }
                rstd = 1.0f / sqrtf(s2 / ((float)a.B * (float)din) + SM_BN_EPS);
                if (b == 0) {
                    a.stats_prev[r * 2] = m;
                    a.stats_prev[r * 2 + 1] = rstd;
                }
            }
            mu[r] = m;
            rs[r] = rstd;
        }
        __syncthreads();
        for (int r = team; r < n; r += NTEAMS)
            for (int k = tl; k < din; k += 16) {
                const float v = (fmaxf(X[r * din + k], 0.f) - mu[r]) * rs[r];
                X[r * din + k] = v;
                a.xout[((long)b * n + r) * a.ldxo + k] = v;
            }
        __syncthreads();
    }
    SM_STAMP(0, 1);
    // P = X W
    lds_mma<false, false>(X, din, W, dout, n, dout, din, [&](int r, int c, float v) { P[r * dout + c] = v; });
    __syncthreads();
    SM_STAMP(0, 2);
    // U = A P (+ P) + bias
    lds_mma<false, false>(A, n, P, dout, n, dout, n, [&](int r, int c, float v) {
        if (a.add_self) v += P[r * dout + c];
        U[r * dout + c] = v + BI[c];
    });
    __syncthreads();
    SM_STAMP(0, 3);
    // l2-normalise rows, BN partials of relu(y)
    for (int r = team; r < n; r += NTEAMS) {
        const long row = (long)b * n + r;
        float ss = 0.f;
        for (int c = tl; c < dout; c += 16) ss += U[r * dout + c] * U[r * dout + c];
        ss = sm_team_sum(ss);
        const float inv = 1.f / fmaxf(sqrtf(ss), SM_L2_EPS);
        float s1 = 0.f;
        for (int c = tl; c < dout; c += 16) {
            const float v = U[r * dout + c] * inv;
            a.y[row * a.ldy + c] = v;
            s1 += fmaxf(v, 0.f);
        }
        if (tl == 0) a.invn[row] = inv;
        if (a.stats) {
            s1 = sm_team_sum(s1);
            const float mean = s1 / (float)dout;
            float m2 = 0.f;
            for (int c = tl; c < dout; c += 16) {
                const float v = fmaxf(U[r * dout + c] * inv, 0.f) - mean;
                m2 += v * v;
            }
            m2 = sm_team_sum(m2);
            if (tl == 0) {
                a.part[row * 2] = mean;
                a.part[row * 2 + 1] = m2;
            }
        }
    }
    SM_STAMP(0, 4);
}

struct SmallBwdArgs {
    const float* adj;      // [B, n, n]
    const float* xin;      // layer input [B, n, din] (ld = ldxin): raw x0 or the BN output slice of layer l-1
    int ldxin;
    const float* W;        // [din, dout]
    const float* y;        // this layer's normalised output [B, n, dout] (ld = ldy)
    int ldy;
    const float* xhat;     // BN output of this layer (ld = ldxh) — null for the last layer
    int ldxh;
    const float* invn;     // [B, n]
    const float* stats;    // [n, 2] of this layer
    const float* part2;    // [B, n, 2] (sum dx, sum dx*xhat) of this layer — null for the last layer
    const float* dx;       // gradient w.r.t. this layer's output [B, n, dout] (ld = lddx)
    int lddx;
    float* dxin;           // gradient w.r.t. the layer input, ACCUMULATED [B, n, din] (ld = lddxin), or null
    int lddxin;
    float* part2_prev;     // [B, n, 2] out: BN-backward partials of layer l-1 (needs dxin), or null
    float* dadj;           // [B, n, n] accumulated, or null
    float* dW;             // slab of this graph: [din, dout]   (slab_stride between graphs)
    float* db;             // slab of this graph: [dout] or null
    long slab_stride;
    int B, n, din, dout;
    int add_self, has_bn, has_relu;
    SmDiv qdin, qdout;
};

__global__ __launch_bounds__(1024) void k_small_gcn_bwd(SmallBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, din = a.din, dout = a.dout;
    float* A = lds;                    // [n][n]
    float* X = A + n * n;              // [n][din]   layer input
    float* W = X + n * din;            // [din][dout]
    float* dU = W + din * dout;        // [n][dout]  staged dx, then dU in place
    float* G = dU + n * dout;          // [n][dout]
    float* P = G + n * dout;           // [n][dout]  staged y, then P = X W
    float* XH = P + n * dout;          // [n][dout]  staged xhat (BN layers)
    float* DX = XH + n * dout;         // [n][din]   gradient w.r.t. the layer input (running total)
    float* IV = DX + n * din;          // [n] 1/||u||
    float* M0 = IV + n;                // [n]
    float* M1 = M0 + n;                // [n]
    float* DB = M1 + n;                // [dout] column sums of dU (bias gradient)
    float* DBW = DB + dout;            // [16 waves][dout] their per-wave partials
    float* RS = DBW + 16 * dout;       // [n] BN rstd of this layer
    float* DA = RS + n;                // [n][n] running dA' (when the level's adjacency gradient is wanted)
    float* PP = DA + (a.dadj ? n * n : 0);   // [B][n][2] staged BN-backward partials
    const int tl = tid & 15, team = tid >> 4;
    const int NT = blockDim.x, NTEAMS = blockDim.x >> 4;
    float* dWb = a.dW + (long)b * a.slab_stride;
    float* dbb = a.db ? a.db + (long)b * a.slab_stride : nullptr;

    SM_STAMP(1, 0);
    // ---- one burst: everything this layer reads from global, all by LDS-DMA (see sm_dma)
    sm_dma(A, n * n, [&](int e) { return a.adj + (long)b * n * n + e; });
    if (a.dadj) sm_dma(DA, n * n, [&](int e) { return a.dadj + (long)b * n * n + e; });
    if (a.has_bn) sm_dma(RS, n, [&](int e) { return a.stats + e * 2 + 1; });
    sm_dma(W, din * dout, [&](int e) { return a.W + e; });
    sm_dma(X, n * din, [&](int e) { return a.xin + ((long)b * n + a.qdin.quot(e)) * a.ldxin + a.qdin.rem(e); });
    if (a.dxin)
        sm_dma(DX, n * din, [&](int e) { return a.dxin + ((long)b * n + a.qdin.quot(e)) * a.lddxin + a.qdin.rem(e); });
    sm_dma(dU, n * dout, [&](int e) { return a.dx + ((long)b * n + a.qdout.quot(e)) * a.lddx + a.qdout.rem(e); });
    sm_dma(P, n * dout, [&](int e) { return a.y + ((long)b * n + a.qdout.quot(e)) * a.ldy + a.qdout.rem(e); });
    if (a.has_bn)
        sm_dma(XH, n * dout, [&](int e) { return a.xhat + ((long)b * n + a.qdout.quot(e)) * a.ldxh + a.qdout.rem(e); });
    sm_dma(IV, n, [&](int e) { return a.invn + (long)b * n + e; });
    if (a.has_bn) sm_dma(PP, a.B * n * 2, [&](int e) { return a.part2 + e; });
    for (int i = tid; i < 17 * dout; i += NT) DB[i] = 0.f;     // DB and DBW
    SM_STAMP(1, 8);
    __syncthreads();
    SM_STAMP(1, 9);
    if (a.has_bn) {
        for (int r = tid; r < n; r += NT) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
            for (int bb = 0; bb < a.B; ++bb) {
                s0 += PP[(bb * n + r) * 2];
                s1 += PP[(bb * n + r) * 2 + 1];
            }
            const float cnt = (float)a.B * (float)dout;
            M0[r] = s0 / cnt;
            M1[r] = s1 / cnt;
        }
        __syncthreads();
    }
    SM_STAMP(1, 1);
    // ---- dU = normalise^T relu^T bn^T dx (in place over the staged dx), one team per row
    // (uniform trip count: the bias-gradient column sums below reduce ACROSS the four row teams of a wave)
    for (int r0 = 0; r0 < n; r0 += NTEAMS) {
        const int r = min(r0 + team, n - 1);
        const bool valid = r0 + team < n;
        const float rstd = a.has_bn ? RS[r] : 1.f;
        const float m0 = a.has_bn ? M0[r] : 0.f, m1 = a.has_bn ? M1[r] : 0.f;
        const float inv = IV[r];
        const bool project = inv < 1.0f / SM_L2_EPS;
        float dot = 0.f;
        for (int c = tl; c < dout; c += 16) {
            float d = dU[r * dout + c];
            const float yy = P[r * dout + c];
            if (a.has_bn) d = rstd * (d - m0 - XH[r * dout + c] * m1);
            if (a.has_relu) d = yy > 0.f ? d : 0.f;
            if (valid) dU[r * dout + c] = d;
            dot += d * yy;
        }
        dot = sm_team_sum(dot);
        for (int c0 = 0; c0 < dout; c0 += 16) {
            const int c = c0 + tl;
            float v = 0.f;
            if (c < dout && valid) {
                const float d = dU[r * dout + c];
                v = project ? inv * (d - P[r * dout + c] * dot) : inv * d;
                dU[r * dout + c] = v;
            }
            // bias gradient = column sums of dU: the four rows of this wave first, then one partial per wave and
            // column, summed in wave order below (LDS float atomics here made the bias gradients differ in the last
            // place from run to run; a serial column walk held a whole wave back for ~4.5k cycles)
            if (dbb) {
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if ((tid & 63) < 16 && c < dout) DBW[(tid >> 6) * dout + c] += v;
            }
        }
    }
    __syncthreads();
    SM_STAMP(1, 2);
    if (dbb)
        for (int c = tid; c < dout; c += NT) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) t += DBW[w * dout + c];     // wave order: bit-reproducible
            dbb[c] = t;
        }
    SM_STAMP(1, 10);
    lds_mma<true, false>(A, n, dU, dout, n, dout, n, [&](int m, int c, float v) {
        G[m * dout + c] = a.add_self ? v + dU[m * dout + c] : v;
    });
    SM_STAMP(1, 11);
    if (a.dadj)
        lds_mma<false, false>(X, din, W, dout, n, dout, din, [&](int m, int c, float v) { P[m * dout + c] = v; },
                              ((n + 15) / 16) * ((dout + 15) / 16));
    SM_STAMP(1, 12);
    __syncthreads();
    SM_STAMP(1, 3);
    // ---- dW = X^T G  -> this graph's slab
    lds_mma<true, false>(X, din, G, dout, din, dout, n, [&](int k, int c, float v) { dWb[k * dout + c] = v; });
    SM_STAMP(1, 4);
    // ---- dXin = G W^T, accumulated into the gradient of the layer input
    if (a.dxin)
        lds_mma<false, true>(G, dout, W, dout, n, din, dout, [&](int r, int k, float v) {
            const float tot = DX[r * din + k] + v;
            DX[r * din + k] = tot;
            a.dxin[((long)b * n + r) * a.lddxin + k] = tot;
        });
    SM_STAMP(1, 5);
    // ---- dA += dU P^T
    if (a.dadj)
        lds_mma<false, true>(dU, dout, P, dout, n, n, dout, [&](int r, int m, float v) {
            a.dadj[(long)b * n * n + r * n + m] = DA[r * n + m] + v;
        });
    SM_STAMP(1, 6);
    // ---- BN-backward partials of layer l-1: (sum_k dX[r][k], sum_k dX[r][k] xhat[r][k]); xhat_{l-1} = X
    if (a.part2_prev) {
        __syncthreads();
        for (int r = team; r < n; r += NTEAMS) {
            float s0 = 0.f, s1 = 0.f;
            for (int k = tl; k < din; k += 16) {
                const float d = DX[r * din + k];
                s0 += d;
                s1 += d * X[r * din + k];
            }
            s0 = sm_team_sum(s0);
            s1 = sm_team_sum(s1);
            if (tl == 0) {
                a.part2_prev[((long)b * n + r) * 2] = s0;
                a.part2_prev[((long)b * n + r) * 2 + 1] = s1;
            }
        }
    }
    SM_STAMP(1, 7);
}

// =========================================================================================================
// Whole-level kernels: ALL layers of a pooled level's GCN stack in one launch per direction.
//
// One workgroup per graph keeps the level's adjacency, every layer's weights and the running activations in LDS
// across the layers; the only thing that crosses workgroups is apply_bn (encoders.py:1048-1052: statistics per node
// index over batch x features), which needs every graph's row partials between two layers.  They travel through
// global memory as TAGGED ENTRIES, without a barrier (round 3; rounds 1-2 used a ticket barrier + reload here): a row's
// pair is one write-through 16-byte store {v0, tag, v1, tag}, tag = (launch sequence number, layer), and the team
// that needs node index r polls the B entries of r with 16-byte sc1 loads until every tag is this launch's
// (sm_poll_row).  The data is its own ready flag: no arrival counter, no acknowledgement wait on the producer's side,
// one memory round trip on the consumer's.  B <= (CUs / 2) workgroups of 1024 threads are co-resident by construction
// (one per CU; small_level_fused_ok also asks the occupancy calculator), so the wait cannot deadlock — and it is bounded
// anyway: a team that polls longer than ~1 s gives up, raises the error words and poisons its statistics with NaN,
// so a broken co-residency assumption shows up as NaN logits and DP_ERR_DEVICE on the next call, never as a hung GPU.
//
// Memory model (each XCD has its own L2; plain stores / loads of different XCDs are not coherent inside one kernel):
// entries are stored sc1 (write-through to memory) and polled sc1 (served past the XCD-local L2); each 8-byte half of
// an entry carries its own tag, so an entry torn at 8 bytes is never mistaken for a whole one; the sequence number
// lives in the workspace's first block (zero once), is read by every workgroup at kernel start and counted up by the
// last workgroup to finish (sm_finish), so tags never repeat across launches or hipGraph replays.
// The finish ticket (bar[2]) is zeroed in stream order by an earlier launch of the same sequence (never by a memset node).
constexpr int SM_SPIN_LIMIT = 1 << 22;
constexpr int SM_BMAX16 = 8;      // whole-level kernels take B <= 128 graphs (16 lanes x 8 entries per node)

// The cross-graph BatchNorm exchange of the whole-level kernels, barrier-free (the scheme of dp_level0.hip "tagged
// entries"): a row's pair travels as one write-through 16-byte entry {v0, tag, v1, tag}, tag = (launch sequence number,
// layer); a 16-lane team polls the B entries of its node index until every tag is this launch's.  Returns false when the
// wait gave up (and raises DP_DEVERR_BARRIER in the device's error word, which the next model-level entry reports:
// diffpool_hip.h "Device-side failures"; bar[1] is the word the prediction head turns into NaN logits).
__device__ inline bool sm_poll_row(ScBuf buf, unsigned off, int B, unsigned want, float (&v0)[SM_BMAX16],
                                   float (&v1)[SM_BMAX16], int spin_limit, int* bar, int* dev_err) {
    const int tl = threadIdx.x & 15;
    int it = 0;
    for (;;) {
        bool ready = true;
#pragma unroll
        for (int u = 0; u < SM_BMAX16; ++u) {
            v0[u] = 0.f;
            v1[u] = 0.f;
            if (16 * u < B) {                                              // (uniform)
                const u32x4 q = sc_ld16(buf, off + 16u * (unsigned)min(tl + 16 * u, B - 1));
                v0[u] = __uint_as_float(q[0]);
                v1[u] = __uint_as_float(q[2]);
                ready = ready && q[1] == want && q[3] == want;
            }
        }
        if (__all(ready)) return true;
        if (++it > spin_limit) {
            __hip_atomic_store(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dev_err_raise(dev_err, DP_DEVERR_BARRIER);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// the last workgroup to finish counts the sequence number up (every workgroup read it at kernel start)
__device__ inline void sm_finish(int* bar, int* seq, unsigned mine, int B) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(bar + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == B - 1)
            __hip_atomic_store(seq, (int)((mine + 1u) & 0x03ffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct SmallLevelFwdArgs {
    const float* adj;      // [B, n, n]
    const float* x0;       // [B, n, dims[0]] (ld = ldx0)
    int ldx0;
    const float* params;
    long w_off[DP_MAX_LAYERS], b_off[DP_MAX_LAYERS];
    int dims[DP_MAX_LAYERS + 1];
    int L;
    float* Y[DP_MAX_LAYERS];       // non-last layers: normalised pre-ReLU output [B, n, dims[l+1]] (ld = ldY[l])
    int ldY[DP_MAX_LAYERS];
    float* invn[DP_MAX_LAYERS];    // [B, n]
    float* stats[DP_MAX_LAYERS];   // non-last layers with BN: [n, 2] (mu, rstd)
    float* Ze;                     // concat buffer [B, n, ldz]; layer l's slice starts at column coff[l]
    int ldz;
    int coff[DP_MAX_LAYERS];
    float* part;                   // exchange: [L-1][n][B] tagged 16-byte entries (row mean, row M2) of relu(y)
    int* bar;                      // bar[1] error word, bar[2] finish ticket (zero at launch)
    int* seq;                      // launch sequence number (tags of the exchange entries), counted up by the last workgroup
    int* dev_err;                  // the device's host-visible error word (or null)
    int spin_limit, target_bias;   // SM_SPIN_LIMIT, 0 (the test knob DP_TEST_BARRIER_FAIL: 64, 1)
    int B, n, add_self, bn;
    int dmax, omax, wtot, btot;    // max layer input width, max output width, total weight / bias floats
    SmDiv qd0;
};

__global__ __launch_bounds__(1024) void k_small_level_fwd(SmallLevelFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, L = a.L;
    const unsigned seq = (unsigned)a.seq[0];               // launch sequence number: the tags of this launch's exchange entries
    float* A = lds;                        // [n][n]
    float* X = A + n * n;                  // [n][dmax]  current layer input
    float* P = X + n * a.dmax;             // [n][omax]
    float* U = P + n * a.omax;             // [n][omax]  pre-normalisation, then y
    float* mu = U + n * a.omax;            // [n]
    float* rs = mu + n;                    // [n]
    float* WL = rs + n;                    // all layers' weights, back to back
    float* BI = WL + a.wtot;               // all layers' biases (zeros where a layer has none)
    const int tl = tid & 15, team = tid >> 4;
    const int NT = blockDim.x, NTEAMS = blockDim.x >> 4;
    SM_STAMP(0, 7);

    // ---- one burst: adjacency, level input, every layer's weights and biases
    sm_dma(A, n * n, [&](int e) { return a.adj + (long)b * n * n + e; });
    {
        const int d0 = a.dims[0];
        sm_dma(X, n * d0, [&](int e) { return a.x0 + ((long)b * n + a.qd0.quot(e)) * a.ldx0 + a.qd0.rem(e); });
    }
    {
        int wo = 0, bo = 0;
        for (int l = 0; l < L; ++l) {
            const int din = a.dims[l], dout = a.dims[l + 1];
            const float* w = a.params + a.w_off[l];
            sm_dma(WL + wo, din * dout, [&](int e) { return w + e; });
            if (a.b_off[l] >= 0) {
                const float* bs = a.params + a.b_off[l];
                sm_dma(BI + bo, dout, [&](int e) { return bs + e; });
            } else {
                for (int i = tid; i < dout; i += NT) BI[bo + i] = 0.f;
            }
            wo += din * dout;
            bo += dout;
        }
    }
    SM_STAMP(0, 8);
    __syncthreads();                       // (drains the LDS-DMA burst: vmcnt)
    SM_STAMP(0, 9);

    int wo = 0, bo = 0;
    for (int l = 0; l < L; ++l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        const bool last = l == L - 1;
        const float* W = WL + wo;
        const float* BL = BI + bo;
        // P = X W
        lds_mma2<false, false>(X, din, W, dout, n, dout, din, [&](int r, int c, float v) { P[r * dout + c] = v; });
        lds_barrier();
        SM_STAMP(0, 10 + 6 * l);
        // U = A P (+ P) + bias
        lds_mma2<false, false>(A, n, P, dout, n, dout, n, [&](int r, int c, float v) {
            if (a.add_self) v += P[r * dout + c];
            U[r * dout + c] = v + BL[c];
        });
        if (l == 2) SM_STAMP(0, 30);
        lds_barrier();
        SM_STAMP(0, 11 + 6 * l);
        // l2-normalise rows -> y (kept in U), saved output, BN partials of relu(y)
        const bool stats = !last && a.bn;
        const ScBuf part_l = sc_buf(a.part + (long)l * a.B * n * 4, (size_t)a.B * n * 16);
        const unsigned bn_tag = ((seq << 4) | (unsigned)(l + 1)) ^ (a.target_bias ? 0x40000000u : 0u);
        for (int r = team; r < n; r += NTEAMS) {
            const long row = (long)b * n + r;
            float ss = 0.f;
            for (int c = tl; c < dout; c += 16) ss += U[r * dout + c] * U[r * dout + c];
            ss = sm_team_sum(ss);
            const float inv = 1.f / fmaxf(sqrtf(ss), SM_L2_EPS);
            float s1 = 0.f;
            float* yg = last ? a.Ze + row * a.ldz + a.coff[l] : a.Y[l] + row * a.ldY[l];
            for (int c = tl; c < dout; c += 16) {
                const float v = U[r * dout + c] * inv;
                U[r * dout + c] = v;
                yg[c] = v;
                s1 += fmaxf(v, 0.f);
            }
            if (tl == 0) a.invn[l][row] = inv;
            if (stats) {
                s1 = sm_team_sum(s1);
                const float mean = s1 / (float)dout;
                float m2 = 0.f;
                for (int c = tl; c < dout; c += 16) {
                    const float v = fmaxf(U[r * dout + c], 0.f) - mean;
                    m2 += v * v;
                }
                m2 = sm_team_sum(m2);
                if (tl == 0) sc_st16(part_l, (unsigned)(((long)r * a.B + b) * 16), sc_tagged(mean, m2, (seq << 4) | (unsigned)(l + 1)));
            }
        }
        SM_STAMP(0, 12 + 6 * l);
        if (last) break;
        if (stats) {
            // every graph's partials of this layer, then the statistics per node index (Chan combine)
            SM_STAMP(0, 13 + 6 * l);
            // one 16-lane team per node index: its B (mean, M2) entries, polled until all are this launch's
            for (int r = team; r < n; r += NTEAMS) {
                float pm[SM_BMAX16], pq[SM_BMAX16];
                const bool ok = sm_poll_row(part_l, (unsigned)((long)r * a.B * 16), a.B, bn_tag, pm, pq, a.spin_limit, a.bar,
                                            a.dev_err);
                float sm = 0.f;
#pragma unroll
                for (int u = 0; u < SM_BMAX16; ++u) sm += (tl + 16 * u < a.B) ? pm[u] : 0.f;
                float m = sm_team_sum(sm) / (float)a.B;
                float s2 = 0.f;
#pragma unroll
                for (int u = 0; u < SM_BMAX16; ++u) {
                    const float d = pm[u] - m;
                    s2 += (tl + 16 * u < a.B) ? pq[u] + (float)dout * d * d : 0.f;
                }
                float rstd = 1.0f / sqrtf(sm_team_sum(s2) / ((float)a.B * (float)dout) + SM_BN_EPS);
                if (!ok) m = rstd = __builtin_nanf("");        // a barrier that gave up must not go unnoticed
                if (tl == 0) {
                    if (b == 0) {
                        a.stats[l][r * 2] = m;
                        a.stats[l][r * 2 + 1] = rstd;
                    }
                    mu[r] = m;
                    rs[r] = rstd;
                }
            }
            SM_STAMP(0, 14 + 6 * l);
        } else {
            for (int r = tid; r < n; r += NT) {
                mu[r] = 0.f;
                rs[r] = 1.f;
            }
        }
        lds_barrier();
        // next layer's input: x = (relu(y) - mu) * rstd, also this layer's slice of the concat buffer
        for (int r = team; r < n; r += NTEAMS)
            for (int k = tl; k < dout; k += 16) {
                const float v = (fmaxf(U[r * dout + k], 0.f) - mu[r]) * rs[r];
                X[r * dout + k] = v;
                a.Ze[((long)b * n + r) * a.ldz + a.coff[l] + k] = v;
            }
        lds_barrier();
        SM_STAMP(0, 15 + 6 * l);
        wo += din * dout;
        bo += dout;
    }
    sm_finish(a.bar, a.seq, seq, a.B);
}

struct SmallLevelBwdArgs {
    const float* adj;      // [B, n, n]
    const float* x0;       // level input [B, n, dims[0]] (ld = ldx0)
    int ldx0;
    const float* params;
    long w_off[DP_MAX_LAYERS], b_off[DP_MAX_LAYERS];
    int dims[DP_MAX_LAYERS + 1];
    int L;
    const float* Y[DP_MAX_LAYERS];     // non-last layers: normalised pre-ReLU output (ld = ldY[l])
    int ldY[DP_MAX_LAYERS];
    const float* invn[DP_MAX_LAYERS];
    const float* stats[DP_MAX_LAYERS];
    const float* Ze;                   // concat buffer: BN outputs of the non-last layers, y of the last
    int ldz;
    int coff[DP_MAX_LAYERS];
    const float* dZe;                  // gradient of the concat buffer [B, n, ldz] (readout scatter etc.), read only
    float* dX0;                        // [B, n, dims[0]] out (overwritten), or null
    float* dadj;                       // [B, n, n] out (overwritten), or null
    float* slabs;                      // parameter-gradient slab of graph 0 (graphs slab_stride apart)
    long slab_stride;
    float* part;                       // exchange: [L-1][B][n][2] (sum dx, sum dx * xhat)
    int* bar;
    int* seq;
    int* dev_err;
    int spin_limit, target_bias;
    int B, n, add_self, bn;
    int omax, wtot, D;                 // max layer output width, total weight floats, concat width
    SmDiv qd0, qD;
    SmDiv qdout[DP_MAX_LAYERS];
};

__global__ __launch_bounds__(1024) void k_small_level_bwd(SmallLevelBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n, L = a.L, D = a.D, d0 = a.dims[0];
    const unsigned seq = (unsigned)a.seq[0];
    float* A = lds;                        // [n][n]
    float* DA = A + n * n;                 // [n][n] running dA (only when wanted)
    float* X0 = DA + (a.dadj ? n * n : 0); // [n][d0]
    float* ZE = X0 + n * d0;               // [n][D]  concat buffer slices (xhat_l for l < L-1, y_{L-1})
    float* DZ = ZE + n * D;                // [n][D]  gradient w.r.t. the concat buffer, running totals
    float* YS = DZ + n * D;                // [L-1][n][omax] normalised pre-ReLU outputs of the non-last layers
    float* WL = YS + (L - 1) * n * a.omax; // all weights
    float* IV = WL + a.wtot;               // [L][n]
    float* RS = IV + L * n;                // [L][n]
    float* dU = RS + L * n;                // [n][omax]
    float* G = dU + n * a.omax;            // [n][omax]
    float* P = G + n * a.omax;             // [n][omax]
    float* M0 = P + n * a.omax;            // [n]
    float* M1 = M0 + n;                    // [n]
    float* DB = M1 + n;                    // [omax]
    float* DBW = DB + a.omax;              // [16 waves][omax] per-wave partials of the bias sums
    const int tl = tid & 15, team = tid >> 4;
    const int NT = blockDim.x, NTEAMS = blockDim.x >> 4;

    // ---- one burst
    sm_dma(A, n * n, [&](int e) { return a.adj + (long)b * n * n + e; });
    sm_dma(X0, n * d0, [&](int e) { return a.x0 + ((long)b * n + a.qd0.quot(e)) * a.ldx0 + a.qd0.rem(e); });
    sm_dma(ZE, n * D, [&](int e) { return a.Ze + ((long)b * n + a.qD.quot(e)) * a.ldz + a.qD.rem(e); });
    sm_dma(DZ, n * D, [&](int e) { return a.dZe + ((long)b * n + a.qD.quot(e)) * a.ldz + a.qD.rem(e); });
    {
        int wo = 0;
        for (int l = 0; l < L; ++l) {
            const int din = a.dims[l], dout = a.dims[l + 1];
            const float* w = a.params + a.w_off[l];
            sm_dma(WL + wo, din * dout, [&](int e) { return w + e; });
            wo += din * dout;
            const float* iv = a.invn[l] + (long)b * n;
            sm_dma(IV + l * n, n, [&](int e) { return iv + e; });
            if (l < L - 1) {
                const float* y = a.Y[l];
                const int ldy = a.ldY[l];
                const SmDiv q = a.qdout[l];
                sm_dma(YS + l * n * a.omax, n * dout, [&](int e) { return y + ((long)b * n + q.quot(e)) * ldy + q.rem(e); });
                if (a.bn) {
                    const float* st = a.stats[l];
                    sm_dma(RS + l * n, n, [&](int e) { return st + e * 2 + 1; });
                }
            }
        }
    }
    if (a.dadj)
        for (int i = tid; i < n * n; i += NT) DA[i] = 0.f;
    __syncthreads();                       // (drains the LDS-DMA burst: vmcnt)

    int wo_end = a.wtot;
    for (int l = L - 1; l >= 0; --l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        const bool last = l == L - 1;
        const bool has_bn = !last && a.bn;
        const bool has_relu = !last;
        wo_end -= din * dout;
        const float* W = WL + wo_end;
        const float* Xl = l == 0 ? X0 : ZE + a.coff[l - 1];     // layer input (row stride ldx)
        const int ldx = l == 0 ? d0 : D;
        const float* Yl = last ? ZE + a.coff[l] : YS + l * n * a.omax;   // normalised output
        const int ldy = last ? D : dout;
        float* dx = DZ + a.coff[l];                               // gradient w.r.t. this layer's output (ld D)
        for (int i = tid; i < 16 * dout; i += NT) DBW[i] = 0.f;
        if (has_bn) {
            // BN backward needs, per node index, the sums over ALL graphs of (dx, dx * xhat)
            const ScBuf part_l = sc_buf(a.part + (long)l * a.B * n * 4, (size_t)a.B * n * 16);
            const unsigned bn_tag = (seq << 4) | (unsigned)(l + 1);
            const float* xh = ZE + a.coff[l];
            for (int r = team; r < n; r += NTEAMS) {
                float s0 = 0.f, s1 = 0.f;
                for (int c = tl; c < dout; c += 16) {
                    const float d = dx[r * D + c];
                    s0 += d;
                    s1 += d * xh[r * D + c];
                }
                s0 = sm_team_sum(s0);
                s1 = sm_team_sum(s1);
                if (tl == 0) sc_st16(part_l, (unsigned)(((long)r * a.B + b) * 16), sc_tagged(s0, s1, bn_tag));
            }
            for (int r = team; r < n; r += NTEAMS) {
                float p0[SM_BMAX16], p1[SM_BMAX16];
                const bool ok = sm_poll_row(part_l, (unsigned)((long)r * a.B * 16), a.B,
                                            bn_tag ^ (a.target_bias ? 0x40000000u : 0u), p0, p1, a.spin_limit, a.bar, a.dev_err);
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int u = 0; u < SM_BMAX16; ++u) {
                    s0 += (tl + 16 * u < a.B) ? p0[u] : 0.f;
                    s1 += (tl + 16 * u < a.B) ? p1[u] : 0.f;
                }
                s0 = sm_team_sum(s0);
                s1 = sm_team_sum(s1);
                const float cnt = (float)a.B * (float)dout;
                if (tl == 0) {
                    M0[r] = ok ? s0 / cnt : __builtin_nanf("");
                    M1[r] = s1 / cnt;
                }
            }
        }
        lds_barrier();
        // ---- dU = normalise^T relu^T bn^T dx, one team per row (uniform trip count: the bias sums reduce across
        // the four row teams of a wave)
        float* dbb = a.b_off[l] >= 0 ? a.slabs + (long)b * a.slab_stride + a.b_off[l] : nullptr;
        for (int r0 = 0; r0 < n; r0 += NTEAMS) {
            const int r = min(r0 + team, n - 1);
            const bool valid = r0 + team < n;
            const float rstd = has_bn ? RS[l * n + r] : 1.f;
            const float m0 = has_bn ? M0[r] : 0.f, m1 = has_bn ? M1[r] : 0.f;
            const float inv = IV[l * n + r];
            const bool project = inv < 1.0f / SM_L2_EPS;
            float dot = 0.f;
            float dv[4];                                   // dout <= 64 on this path (see small_level_fused_ok)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = tl + 16 * k;
                float d = 0.f;
                if (c < dout) {
                    d = dx[r * D + c];
                    const float yy = Yl[r * ldy + c];
                    if (has_bn) d = rstd * (d - m0 - ZE[r * D + a.coff[l] + c] * m1);
                    if (has_relu) d = yy > 0.f ? d : 0.f;
                    dot += d * yy;
                }
                dv[k] = d;
            }
            dot = sm_team_sum(dot);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = tl + 16 * k;
                float v = 0.f;
                if (c < dout && valid) {
                    v = project ? inv * (dv[k] - Yl[r * ldy + c] * dot) : inv * dv[k];
                    dU[r * dout + c] = v;
                }
                if (dbb && 16 * k < dout) {
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if ((tid & 63) < 16 && c < dout) DBW[(tid >> 6) * dout + c] += v;   // one partial per wave, no atomics
                }
            }
        }
        lds_barrier();
        if (dbb)
            for (int c = tid; c < dout; c += NT) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 16; ++w) t += DBW[w * dout + c];   // wave order: bit-reproducible
                dbb[c] = t;
            }
        // G = A^T dU (+ dU);  P = X W (for dA)
        lds_mma2<true, false>(A, n, dU, dout, n, dout, n, [&](int m, int c, float v) {
            G[m * dout + c] = a.add_self ? v + dU[m * dout + c] : v;
        });
        if (a.dadj)
            lds_mma2<false, false>(Xl, ldx, W, dout, n, dout, din, [&](int m, int c, float v) { P[m * dout + c] = v; },
                                  ((n + 15) / 16) * ((dout + 15) / 16));
        lds_barrier();
        // dW = X^T G -> this graph's slab
        {
            float* dWb = a.slabs + (long)b * a.slab_stride + a.w_off[l];
            lds_mma2<true, false>(Xl, ldx, G, dout, din, dout, n, [&](int k, int c, float v) { dWb[k * dout + c] = v; });
        }
        // gradient w.r.t. the layer input: G W^T
        if (l > 0) {
            float* dxin = DZ + a.coff[l - 1];
            lds_mma2<false, true>(G, dout, W, dout, n, din, dout, [&](int r, int k, float v) { dxin[r * D + k] += v; },
                                 1);
        } else if (a.dX0) {
            lds_mma2<false, true>(G, dout, W, dout, n, din, dout, [&](int r, int k, float v) {
                a.dX0[((long)b * n + r) * d0 + k] = v;
            }, 1);
        }
        // dA += dU P^T
        if (a.dadj)
            lds_mma2<false, true>(dU, dout, P, dout, n, n, dout, [&](int r, int m, float v) { DA[r * n + m] += v; }, 2);
        lds_barrier();
    }
    if (a.dadj)
        for (int i = tid; i < n * n; i += NT) a.dadj[(long)b * n * n + i] = DA[i];
    sm_finish(a.bar, a.seq, seq, a.B);
}

size_t small_lds_floats_fwd(int B, int n, int din, int dout) {
    return (size_t)n * n + (size_t)n * din + (size_t)din * dout + 2 * (size_t)n * dout + 2 * n + dout +
           (size_t)B * n * 2 + 16;
}
size_t small_lds_floats_bwd(int B, int n, int din, int dout) {
    return 2 * (size_t)n * n + 2 * (size_t)n * din + (size_t)din * dout + 4 * (size_t)n * dout + 4 * n + 17 * (size_t)dout +
           (size_t)B * n * 2 + 16;
}

}  // namespace

bool small_level_supported(int B, int n, int din, int dout) {
    if (n > 64 || din > 256 || dout > 256) return false;
    const size_t f = small_lds_floats_fwd(B, n, din, dout), bw = small_lds_floats_bwd(B, n, din, dout);
    return (f > bw ? f : bw) * sizeof(float) <= 150 * 1024;
}

static void small_attr(Seq& q) {
    static DynLdsOnce fwd, bwd;
    ensure_dyn_lds(q, fwd, reinterpret_cast<const void*>(&k_small_gcn_fwd), 160 * 1024, "k_small_gcn_fwd");
    ensure_dyn_lds(q, bwd, reinterpret_cast<const void*>(&k_small_gcn_bwd), 160 * 1024, "k_small_gcn_bwd");
}

void small_gcn_fwd(Seq& q, const float* adj, const float* x0, int ldx0, const float* yprev, int ldyp,
                   const float* part_prev, float* stats_prev, float* xout, int ldxo, const float* W, const float* bias,
                   float* y, int ldy, float* invn, float* part, int B, int n, int din, int dout, int add_self,
                   int stats) {
    if (!q.ok()) return;
    small_attr(q);
    if (!q.ok()) return;
    SmallFwdArgs a{adj, x0, ldx0, yprev, ldyp, part_prev, stats_prev, xout, ldxo, W, bias, y, ldy, invn, part,
                   B, n, din, dout, add_self, stats, sm_div(din)};
    hipLaunchKernelGGL(k_small_gcn_fwd, dim3(B), dim3(1024), small_lds_floats_fwd(B, n, din, dout) * sizeof(float),
                       q.stream, a);
    q.check_launch("small_gcn_fwd");
}

void small_gcn_bwd(Seq& q, const float* adj, const float* xin, int ldxin, const float* W, const float* y, int ldy,
                   const float* xhat, int ldxh, const float* invn, const float* stats, const float* part2,
                   const float* dx, int lddx, float* dxin, int lddxin, float* part2_prev, float* dadj, float* dW,
                   float* db, long slab_stride, int B, int n, int din, int dout, int add_self, int has_bn,
                   int has_relu) {
    if (!q.ok()) return;
    small_attr(q);
    if (!q.ok()) return;
    SmallBwdArgs a{adj, xin, ldxin, W, y, ldy, xhat, ldxh, invn, stats, part2, dx, lddx, dxin, lddxin, part2_prev,
                   dadj, dW, db, slab_stride, B, n, din, dout, add_self, has_bn, has_relu, sm_div(din), sm_div(dout)};
    hipLaunchKernelGGL(k_small_gcn_bwd, dim3(B), dim3(1024), small_lds_floats_bwd(B, n, din, dout) * sizeof(float),
                       q.stream, a);
    q.check_launch("small_gcn_bwd");
}


// ---- whole-level launchers
static size_t small_level_lds_fwd(int B, int n, const int* dims, int L) {
    int dmax = 0, omax = 0, wtot = 0, btot = 0;
    for (int l = 0; l < L; ++l) {
        dmax = dims[l] > dmax ? dims[l] : dmax;
        omax = dims[l + 1] > omax ? dims[l + 1] : omax;
        wtot += dims[l] * dims[l + 1];
        btot += dims[l + 1];
    }
    return (size_t)n * n + (size_t)n * dmax + 2 * (size_t)n * omax + 2 * n + wtot + btot + 64 + 16;
}
static size_t small_level_lds_bwd(int B, int n, const int* dims, int L, bool dadj) {
    int omax = 0, wtot = 0, D = 0;
    for (int l = 0; l < L; ++l) {
        omax = dims[l + 1] > omax ? dims[l + 1] : omax;
        wtot += dims[l] * dims[l + 1];
        D += dims[l + 1];
    }
    return (size_t)n * n * (dadj ? 2 : 1) + (size_t)n * dims[0] + 2 * (size_t)n * D + (size_t)(L - 1) * n * omax + wtot +
           2 * (size_t)L * n + 3 * (size_t)n * omax + 2 * n + 17 * (size_t)omax + 64 + 16;
}
static int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : -1;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0 ? v : 0;
}
// The runtime admits at least one 1024-thread workgroup of each whole-level kernel per CU at the LDS size they are
// launched with (asked once per device; a kernel that could not be resident at all must never meet a grid barrier).
static bool level_kernels_admitted() {
    static std::atomic<int> state[64];      // 0 unknown, 1 yes, -1 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int v = state[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        v = 1;
        const void* fns[2] = {reinterpret_cast<const void*>(&k_small_level_fwd),
                              reinterpret_cast<const void*>(&k_small_level_bwd)};
        for (const void* fn : fns) {
            int blocks = 0;
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 1024, 150 * 1024);
            if (e != hipSuccess || blocks < 1) {
                (void)hipGetLastError();
                set_error("whole-level kernels disabled: the runtime admits %d resident 1024-thread workgroups with 150 KiB "
                          "of LDS per CU (%s)", blocks, hipGetErrorString(e));
                v = -1;
            }
        }
        state[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0;
}
// All layers of the level in one launch: shapes the kernels take AND a batch whose workgroups are certainly
// co-resident (one 1024-thread workgroup per CU; half the CUs is the margin for a partitioned or shared device).
bool small_level_fused_ok(int B, int n, const int* dims, int L, bool dadj) {
    if (knobs().no_level_fusion || L < 2 || L > DP_MAX_LAYERS || n > 64) return false;
    for (int l = 0; l <= L; ++l)
        if (dims[l] > 256 || (l > 0 && dims[l] > 64)) return false;
    if (small_level_lds_fwd(B, n, dims, L) * sizeof(float) > 150 * 1024) return false;
    if (small_level_lds_bwd(B, n, dims, L, dadj) * sizeof(float) > 150 * 1024) return false;
    const int cus = device_cus();
    return cus > 0 && B <= cus / 2 && B <= 16 * SM_BMAX16 && level_kernels_admitted();
}
size_t small_level_part_floats(int B, int n, int L) { return (size_t)(L > 1 ? L - 1 : 1) * B * n * 4; }

void small_level_fwd(Seq& q, const SmallLevelIO& io, int B, int n, const int* dims, int L, int add_self, int bn) {
    if (!q.ok()) return;
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_small_level_fwd), 159 * 1024, "k_small_level_fwd");
    if (!q.ok()) return;
    SmallLevelFwdArgs a{};
    a.adj = io.adj; a.x0 = io.x0; a.ldx0 = io.ldx0; a.params = io.params; a.L = L;
    a.Ze = io.Ze; a.ldz = io.ldz; a.part = io.part; a.bar = io.bar; a.seq = q.seq_word;
    a.B = B; a.n = n; a.add_self = add_self; a.bn = bn;
    for (int l = 0; l <= L; ++l) a.dims[l] = dims[l];
    for (int l = 0; l < L; ++l) {
        a.w_off[l] = io.w_off[l]; a.b_off[l] = io.b_off[l];
        a.Y[l] = io.Y[l]; a.ldY[l] = io.ldY[l]; a.invn[l] = io.invn[l]; a.stats[l] = io.stats[l];
        a.coff[l] = io.coff[l];
        a.dmax = dims[l] > a.dmax ? dims[l] : a.dmax;
        a.omax = dims[l + 1] > a.omax ? dims[l + 1] : a.omax;
        a.wtot += dims[l] * dims[l + 1];
        a.btot += dims[l + 1];
    }
    a.qd0 = sm_div(dims[0]);
    a.dev_err = device_error_word();
    a.spin_limit = knobs().test_barrier_fail ? 64 : SM_SPIN_LIMIT;
    a.target_bias = knobs().test_barrier_fail ? 1 : 0;
    hipLaunchKernelGGL(k_small_level_fwd, dim3(B), dim3(1024), small_level_lds_fwd(B, n, dims, L) * sizeof(float),
                       q.stream, a);
    q.check_launch("small_level_fwd");
}

void small_level_bwd(Seq& q, const SmallLevelIO& io, const float* dZe, float* dX0, float* dadj, float* slabs,
                     long slab_stride, int B, int n, const int* dims, int L, int add_self, int bn) {
    if (!q.ok()) return;
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_small_level_bwd), 159 * 1024, "k_small_level_bwd");
    if (!q.ok()) return;
    SmallLevelBwdArgs a{};
    a.adj = io.adj; a.x0 = io.x0; a.ldx0 = io.ldx0; a.params = io.params; a.L = L;
    a.Ze = io.Ze; a.ldz = io.ldz; a.part = io.part; a.bar = io.bar; a.seq = q.seq_word;
    a.dZe = dZe; a.dX0 = dX0; a.dadj = dadj; a.slabs = slabs; a.slab_stride = slab_stride;
    a.B = B; a.n = n; a.add_self = add_self; a.bn = bn;
    for (int l = 0; l <= L; ++l) a.dims[l] = dims[l];
    for (int l = 0; l < L; ++l) {
        a.w_off[l] = io.w_off[l]; a.b_off[l] = io.b_off[l];
        a.Y[l] = io.Y[l]; a.ldY[l] = io.ldY[l]; a.invn[l] = io.invn[l]; a.stats[l] = io.stats[l];
        a.coff[l] = io.coff[l];
        a.omax = dims[l + 1] > a.omax ? dims[l + 1] : a.omax;
        a.wtot += dims[l] * dims[l + 1];
        a.D += dims[l + 1];
        a.qdout[l] = sm_div(dims[l + 1]);
    }
    a.qd0 = sm_div(dims[0]);
    a.qD = sm_div(a.D);
    a.dev_err = device_error_word();
    a.spin_limit = knobs().test_barrier_fail ? 64 : SM_SPIN_LIMIT;
    a.target_bias = knobs().test_barrier_fail ? 1 : 0;
    hipLaunchKernelGGL(k_small_level_bwd, dim3(B), dim3(1024),
                       small_level_lds_bwd(B, n, dims, L, dadj != nullptr) * sizeof(float), q.stream, a);
    q.check_launch("small_level_bwd");
}

#ifdef DP_STAMP
extern "C" __attribute__((visibility("default"))) int dp_debug_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_stamps), sizeof(unsigned long long) * 64);
}
#endif

}  // namespace dp
