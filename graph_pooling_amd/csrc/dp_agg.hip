// Adjacency aggregation  U[b] = op(A[b]) · V[b]   with A [n x n] fp32 as delivered by the caller and
// V [n x C] a narrow feature panel (C <= 128) — GraphConv's `torch.matmul(adj, x)` (encoders.py:965),
// its transpose in the backward pass, and the adjacency passes of the pooling step (encoders.py:1279).
//
// This is the pass that touches the padded dense adjacency, the only large operand of the DiffPool
// path (B*N*N*4 bytes: 20 MB at the DD shape against 1.6 MB of features), so it is HBM-bound and the
// design goal is bytes in flight, not MFMA rate:
//   * one workgroup = 32 output rows of one graph; its whole adjacency panel (32 x n, or n x 32 for the
//     transposed pass) is requested at once with direct-to-LDS loads (global_load_lds_dwordx4, 1 KiB per
//     wave instruction, full 128-B lines, no VGPR staging) -> every byte of A is in flight right after
//     launch (320 workgroups x 64 KB = the whole 20 MB at the DD shape)
//   * the four waves split K; V fragments come straight from L2 (V is re-read by the 16 row tiles of a
//     graph but is only n*C*4 = 80 KB) with a one-step register prefetch
//   * exact fp32 MFMA (16x16x4), partial tiles reduced across waves through LDS, then the epilogue:
//     plain store (optionally accumulating), or the fused GraphConv tail  (+P) + bias -> l2-normalise ->
//     (ReLU statistics for apply_bn)  of encoders.py:966-972 / 1062-1064.
// NN panel image: [32][ldp], ldp = 4 mod 64 floats  -> ds_read_b128 fragment reads are conflict-free
//                 (each lane reads 4 consecutive k; the V fragment loads use the same k order)
// TN panel image: [n][32] (8 rows per 1-KiB DMA piece) -> ds_read_b32, 2-way conflicts (minor next to
//                 the 32-cycle MFMA)
#include <cstdlib>

#include "dp_common.h"

namespace dp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AGG_L2_EPS 1e-12f

#ifdef DP_STAMP
// diagnostic build only (see dp_small.hip): phase stamps of workgroup 0 of the last panel-kernel launch
__device__ unsigned long long g_agg_stamps[32];   // [0,16) plain-store launches, [16,32) fused-tail launches
#define AGG_STAMP(i)                                                                              \
    do {                                                                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_agg_stamps[(i) + (a.U ? 0 : 16)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define AGG_STAMP(i) \
    do {             \
    } while (0)
#endif

#ifdef DP_STAMP
#define AGG_DBG(a) ((a).dbg)
#else
#define AGG_DBG(a) 0
#endif

struct AggArgs {
    const float* A;      // [B, n, n]
    const float* V;      // [B, n, C] (ldv)
    int ldv;
    int n, C;
    int tiles;           // row tiles per graph
    int dbg;             // phase-ablation switches (DP_AGG_DEBUG; DP_STAMP diagnostic build only): 1 no multiply loop,
                         // 2 no panel DMA.  The product build compiles them out (AGG_DBG == 0).
    // exact-bf16 fast path (binary adjacency): packed op(A) [B, n, pk_ld] bf16, the 3-plane bf16 split of V
    // and the device flag written by k_adj_pack (0 = every entry of A is bf16-exact)
    const unsigned short* pk_A;
    int pk_ld;
    const int* pk_flag;
    const unsigned short* Vs;   // [B][3][vs_ct][K8][16][8]
    int vs_k8;
    int vs_ct;                  // ceil(C / 16)
    const int* run_if;          // non-null: the launch is a fallback that runs only when *run_if != 0
    // plain epilogue
    float* U;            // [B, n, C] (ldu) or null
    int ldu;
    float beta;          // U = acc + beta * U
    // fused GraphConv epilogue (NN only), U == null
    const float* P;      // add_self operand (same layout as V) or null
    GroupCPtrs bias;
    RowGroups g;
    GroupPtrs yout;
    float* invn;         // [B, n, G]
    float* part;         // [B, n, G, 2] or null
    int normalize, stats_mode;
};

constexpr int AGG_KP = 1024;       // K columns per LDS panel (NN); 128-KiB panels at most

__device__ inline void dma16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float agg_team_sum(float v) { return row16_sum(v); }

// ---------------------------------------------------------------------------------------------------------
// exact-fp32 accumulate: acc += op(A)[r0.., :] · V   (any adjacency values)
template <bool TRANS, int CT, int AGG_RT, int NW>
__device__ __forceinline__ void accumulate_fp32(const AggArgs& a, int b, int r0, float* lds,
                                                f32x4 (&acc)[AGG_RT / 16][CT]) {
    constexpr int MI = AGG_RT / 16;
    const int n = a.n;
    const float* A = a.A + (long)b * n * n;
    const float* V = a.V + (long)b * n * a.ldv;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int kpanel = TRANS ? n : min(n, AGG_KP);
    const int segs = (kpanel + 255) / 256;
    const int ldp = TRANS ? AGG_RT : segs * 256 + 4;  // NN: = 4 (mod 64)

    for (int kbase = 0; kbase < n; kbase += kpanel) {
        const int kw = min(kpanel, n - kbase);        // valid K columns in this panel
        // ---------------- V fragments: three statically named register buffers rotate through the k-steps
        // (the loop is unrolled x3, so no register copies and the compiler can wait with a COUNTED vmcnt: the
        // fragments multiplied in a step were requested two steps earlier).  Loads are unconditional on
        // clamped addresses (no exec-mask branches); out-of-range elements are zeroed by a select.  The first
        // two steps' fragments go out ahead of the panel burst so they do not queue behind 32-64 KB of adjacency.
        const int steps = (kw + 15) / 16;
        float b0[4][CT], b1[4][CT], b2[4][CT];
        auto load_b = [&](int step, float (&dst)[4][CT]) {
            // no predicate and no select on the loaded values: rows are clamped to n-1 (the padded K tail of the
            // LDS panel is zeroed below, so those products vanish), columns >= C are clamped duplicates that
            // land in output columns nobody stores
            const int k0 = kbase + step * 16 + kq * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vrow = V + (long)min(k0 + j, n - 1) * a.ldv;
#pragma unroll
                for (int cb = 0; cb < CT; ++cb) dst[j][cb] = vrow[min(cb * 16 + l15, a.C - 1)];
            }
        };
        auto mma = [&](int step, const float (&bf)[4][CT]) {
            const int kl = step * 16 + kq * 4;         // panel-local k of this lane's 4 values
            float av[MI][4];
#pragma unroll
            for (int rb = 0; rb < MI; ++rb) {
                if (!TRANS) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(lds + (rb * 16 + l15) * ldp + kl);
                    av[rb][0] = t[0]; av[rb][1] = t[1]; av[rb][2] = t[2]; av[rb][3] = t[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[rb][j] = lds[(kl + j) * AGG_RT + rb * 16 + l15];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < MI; ++rb)
#pragma unroll
                    for (int cb = 0; cb < CT; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][j], bf[j][cb], acc[rb][cb], 0, 0, 0);
        };
        load_b(wave, b0);
        load_b(wave + NW, b1);
        // ---------------- panel -> LDS (everything in flight at once)
        if (AGG_DBG(a) & 2) {
        } else if (!TRANS) {
            // rows r0..r0+RT-1, columns kbase..kbase+kw: pieces (row i, segment s) of 256 floats
            const int pieces = AGG_RT * segs;
            for (int pc = wave; pc < pieces; pc += NW) {
                const int i = pc / segs, s = pc % segs;
                const int row = min(r0 + i, n - 1);
                int col = kbase + s * 256 + lane * 4;
                col = min(col, n - 4);                 // clamp: finite duplicates, multiplied by zero V rows
                dma16(A + (long)row * n + col, lds + i * ldp + s * 256);
            }
        } else {
            // columns r0..r0+RT-1 of rows k: 1-KiB pieces of (256/RT) rows x RT floats; every row a k-step can
            // touch is written (rows >= n are finite duplicates of row n-1 and meet zero V rows)
            constexpr int RPP = 256 / AGG_RT;            // panel rows per piece
            constexpr int LPR = AGG_RT / 4;              // lanes per row
            const int pieces = (((kw + 15) / 16) * 16) / RPP;
            for (int pc = wave; pc < pieces; pc += NW) {
                const int k = min(kbase + pc * RPP + lane / LPR, n - 1);
                const int col = min(r0 + (lane % LPR) * 4, n - 4);
                dma16(A + (long)k * n + col, lds + pc * 256);
            }
        }
        __syncthreads();                               // drains the DMA (vmcnt(0)) and publishes the panel
        if (kw & 15) {
            // zero the padded K tail [kw, ceil16(kw)) of the panel (it holds finite duplicates)
            const int k1 = steps * 16, tail = k1 - kw;
            for (int e = threadIdx.x; e < tail * AGG_RT; e += NW * 64) {
                if (!TRANS) lds[(e / tail) * ldp + kw + e % tail] = 0.f;
                else lds[(kw + e / AGG_RT) * AGG_RT + e % AGG_RT] = 0.f;
            }
            __syncthreads();
        }
        const int send = (AGG_DBG(a) & 1) ? 0 : steps;
        for (int step = wave; step < send; step += 3 * NW) {
            load_b(step + 2 * NW, b2);
            mma(step, b0);
            if (step + NW < send) {
                load_b(step + 3 * NW, b0);
                mma(step + NW, b1);
            }
            if (step + 2 * NW < send) {
                load_b(step + 4 * NW, b1);
                mma(step + 2 * NW, b2);
            }
        }
        __syncthreads();                               // panel may be overwritten (next K panel / reduction)
    }

}

typedef short agg_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 agg_bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void dma16_raw(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// exact-bf16 accumulate for bf16-representable adjacency (0/1 graphs): acc += Abf[r0.., :] · (V_hi + V_mid + V_lo).
// Every product of two bf16 numbers is exact in fp32 and hi+mid+lo reproduces the fp32 V bit for bit, so the
// result is the same fp32-accumulated sum of exact products the fp32 MFMA gives — at 16/3 of its rate and half
// the adjacency bytes.  The packed operand is already oriented (A or A^T), so there is one loop for both passes.
// LDS panel image: [RT][ldp] bf16, ldp = 8 (mod 128)  -> ds_read_b128 A-fragment reads are conflict-free.
template <int CT, int AGG_RT, int NW>
__device__ __forceinline__ bool accumulate_bf16(const AggArgs& a, int b, int r0, float* ldsf, int flag,
                                                f32x4 (&acc)[AGG_RT / 16][CT]) {
    constexpr int MI = AGG_RT / 16;
    unsigned short* lds = reinterpret_cast<unsigned short*>(ldsf);
    const int n = a.n, np = a.pk_ld;
    const unsigned short* A = a.pk_A + (long)b * n * np;
    const int K8 = a.vs_k8;
    const int CTt = a.vs_ct;
    const unsigned short* Vs = a.Vs + (long)b * 3 * CTt * K8 * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int segs = (n + 511) / 512;
    const int ldp = segs * 512 + 8;
    const int steps = (n + 31) / 32;

    agg_s16x8 f0[3][CT], f1[3][CT];
    auto load_b = [&](int step, agg_s16x8 (&dst)[3][CT]) {
        const int st = min(step, steps - 1);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int cb = 0; cb < CT; ++cb)
                dst[p][cb] = *reinterpret_cast<const agg_s16x8*>(
                    Vs + ((((long)p * CTt + min(cb, CTt - 1)) * K8 + st * 4 + kq) * 16 + l15) * 8);
    };
    auto mma = [&](int step, const agg_s16x8 (&bf)[3][CT]) {
        agg_s16x8 av[MI];
#pragma unroll
        for (int rb = 0; rb < MI; ++rb)
            av[rb] = *reinterpret_cast<const agg_s16x8*>(lds + (rb * 16 + l15) * ldp + step * 32 + kq * 8);
#pragma unroll
        for (int p = 2; p >= 0; --p)      // lo, mid, hi: small terms first
#pragma unroll
            for (int rb = 0; rb < MI; ++rb)
#pragma unroll
                for (int cb = 0; cb < CT; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(agg_bf16x8, av[rb]),
                                                                          __builtin_bit_cast(agg_bf16x8, bf[p][cb]),
                                                                          acc[rb][cb], 0, 0, 0);
    };
    AGG_STAMP(1);
    // (Asking for four k-steps of V up front instead of two, registers permitting: slower, 11.9 vs 8.0 us per DD
    // launch -- a deeper burst only queues the adjacency panel behind more V traffic.)
    {
        load_b(wave, f0);
        load_b(wave + NW, f1);
        // rows r0..r0+RT-1 of the packed operand: pieces (row i, segment s) of 512 bf16 = 1 KiB
#pragma unroll
        for (int i = wave; i < AGG_RT; i += NW) {
            const unsigned short* arow = A + (long)min(r0 + i, n - 1) * np;
            for (int s = 0; s < segs; ++s) {
                const int col = min(s * 512 + lane * 8, np - 8);   // clamp: finite duplicates, they meet zero V planes
                dma16_raw(arow + col, lds + i * ldp + s * 512);
            }
        }
        AGG_STAMP(2);
        // The exactness flag was requested before any of the loads above; only now is it looked at, so its
        // latency hides behind their issue.  A non-exact adjacency (rare) drains the speculative DMA and
        // hands the tile to the fp32 path.
        if (__builtin_amdgcn_readfirstlane(flag) != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            return false;
        }
        __syncthreads();
        AGG_STAMP(3);
    }
    for (int step = wave; step < steps; step += 2 * NW) {
        mma(step, f0);
        if (step + NW < steps) {
            load_b(step + 2 * NW, f0);
            mma(step + NW, f1);
            load_b(step + 3 * NW, f1);
        }
    }
    AGG_STAMP(4);
    __syncthreads();
    AGG_STAMP(5);
    return true;
}

template <bool TRANS, int CT, int AGG_RT, int NW>
__global__ __launch_bounds__(NW * 64) void k_aggregate(AggArgs a) {
    constexpr int NT = NW * 64;
    constexpr int MI = AGG_RT / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD-aware work mapping: consecutive block ids are dealt round-robin over the 8 XCDs (each with its own
    // L2), so block id -> work id is remapped (bijectively) to give every XCD one CONTIGUOUS run of
    // (graph, row tile) items: the 16 row tiles of a graph then share one L2 for their V operand instead of
    // fetching it into all eight (measured: 32 MB -> ~22 MB of fabric reads per DD launch).
    if (a.run_if && __builtin_amdgcn_readfirstlane(*a.run_if) == 0) return;
    AGG_STAMP(0);
    const int nwg = gridDim.x, tiles = a.tiles;
    int wid;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rm = nwg & 7;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int b = wid / tiles;
    const int r0 = (wid % tiles) * AGG_RT;
    const int n = a.n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;

    const bool packed = a.pk_A != nullptr;
    const int flag = packed ? *a.pk_flag : 1;          // consumed inside accumulate_bf16, after its loads
    constexpr int CTP = CT * 16 + 1;
    constexpr int TOT = AGG_RT * CT * 16;
    constexpr int NE = (TOT + NT - 1) / NT;            // output elements per thread
    // Epilogue operands do not depend on the product: ask for them before anything else so their latency is
    // spent under the multiply (plain store: the old U when beta != 0; fused tail: bias + the add_self operand).
    float pre[NE], preb[NE];                           // kept apart: adding them here would wait for each load
    {
        const bool accum = a.U && a.beta != 0.f;
        const float* src = a.U ? a.U : a.P;
        const int ld = a.U ? a.ldu : a.ldv;
        const bool want = a.U ? accum : a.P != nullptr;
        const int c01 = (!a.U && a.g.G == 2) ? a.g.c0[1] : 0x7fffffff;
        const float* bias0 = a.U ? nullptr : a.bias.p[0];
        const float* bias1 = a.U ? nullptr : a.bias.p[1];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = min((int)threadIdx.x + k * NT, TOT - 1);
            const int r = e / (CT * 16), cc = min(e % (CT * 16), a.C - 1);
            pre[k] = 0.f;
            preb[k] = 0.f;
            if (want) pre[k] = src[((long)b * n + min(r0 + r, n - 1)) * ld + cc];
            const float* bias = cc >= c01 ? bias1 : bias0;
            const int cb = cc >= c01 ? cc - c01 : cc - (a.U ? 0 : a.g.c0[0]);
            if (bias) preb[k] = bias[cb];
        }
    }
    float* red = lds;                                  // [wave][RT][CTP]; overlays the panel
    float* tile = red + NW * AGG_RT * CTP;              // summed tile [RT][CTP]
    {
        f32x4 acc[MI][CT];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(packed && accumulate_bf16<CT, AGG_RT, NW>(a, b, r0, lds, flag, acc)))
            accumulate_fp32<TRANS, CT, AGG_RT, NW>(a, b, r0, lds, acc);

        // ---------------- cross-wave reduction through LDS
#pragma unroll
        for (int rb = 0; rb < MI; ++rb)
#pragma unroll
            for (int cb = 0; cb < CT; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[(wave * AGG_RT + rb * 16 + kq * 4 + r) * CTP + cb * 16 + l15] = acc[rb][cb][r];
        __syncthreads();
        AGG_STAMP(6);
        float sum[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {                  // all LDS reads of the NW-way sum in flight together
            const int e = min((int)threadIdx.x + k * NT, TOT - 1);
            const int r = e / (CT * 16), c = e % (CT * 16);
            float t = red[r * CTP + c];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += red[(w * AGG_RT + r) * CTP + c];
            sum[k] = t;
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = threadIdx.x + k * NT;
            const int r = e / (CT * 16), c = e % (CT * 16);
            if (a.U) {
                const int row = r0 + r;
                if (e < TOT && row < n && c < a.C)
                    a.U[((long)b * n + row) * a.ldu + c] = a.beta != 0.f ? sum[k] + a.beta * pre[k] : sum[k];
            } else if (e < TOT) {
                tile[r * CTP + c] = c < a.C ? (sum[k] + preb[k]) + pre[k] : sum[k];
            }
        }
    }
    AGG_STAMP(7);
    if (a.U) return;
    __syncthreads();

    // ---------------- fused GraphConv tail (encoders.py:966-972): one 16-lane team per (row, group)
    // All of a team's (row, group) items are in flight together (unrolled, predicated instead of branched) and a
    // row's values stay in registers across the three passes: the tail is a chain of LDS reads and 16-lane sums,
    // and run one item after another it costs as much as the multiply.
    const int tl = threadIdx.x & 15, team = threadIdx.x >> 4;
    constexpr int TEAMS = NT / 16;
    constexpr int ITER = (AGG_RT * 2 + TEAMS - 1) / TEAMS;
    const int G = a.g.G;
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
        if (j * TEAMS >= AGG_RT * G) break;             // uniform
        const int it = team + j * TEAMS;
        const int r = min(G == 2 ? it >> 1 : it, AGG_RT - 1), g = G == 2 ? (it & 1) : 0;
        const int node = r0 + r;
        const bool on = it < AGG_RT * G && node < n;
        const long row = (long)b * n + min(node, n - 1);
        const int c0 = g ? a.g.c0[1] : a.g.c0[0], w = g ? a.g.w[1] : a.g.w[0];
        const float* u = tile + r * CTP + c0;
        float uv[CT];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < CT; ++k) {
            const int c = tl + 16 * k;
            const float t = u[min(c, w - 1)];
            uv[k] = c < w ? t : 0.f;
            ss += uv[k] * uv[k];
        }
        ss = agg_team_sum(ss);
        const float inv = a.normalize ? 1.f / fmaxf(sqrtf(ss), AGG_L2_EPS) : 1.f;
        float* y = (g ? a.yout.p[1] : a.yout.p[0]) + row * (g ? a.yout.ld[1] : a.yout.ld[0]);
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < CT; ++k) {
            const int c = tl + 16 * k;
            uv[k] *= inv;
            if (on && c < w) y[c] = uv[k];
            if (a.stats_mode == 1) uv[k] = fmaxf(uv[k], 0.f);
            s1 += uv[k];
        }
        if (on && tl == 0 && a.invn) a.invn[row * G + g] = inv;
        if (a.stats_mode && a.part) {
            s1 = agg_team_sum(s1);
            const float mean = s1 / (float)w;
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < CT; ++k) {
                const float v = (tl + 16 * k < w) ? uv[k] - mean : 0.f;
                m2 += v * v;
            }
            m2 = agg_team_sum(m2);
            if (on && tl == 0) {
                a.part[(row * G + g) * 2 + 0] = mean;
                a.part[(row * G + g) * 2 + 1] = m2;
            }
        }
    }
    AGG_STAMP(8);
}


// Epilogue shared by the wide kernels.  Lane (kq, l15) holds rows rw0 + rb*16 + kq*4 + r, column cb*16 + l15.
template <int CT>
__device__ __forceinline__ void aggw_epilogue(const AggArgs& a, f32x4 (&acc)[2][CT], int b, int rw0) {
    const int n = a.n;
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, kq = lane >> 4;
    if (a.U) {
        const bool accum = a.beta != 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rw0 + rb * 16 + kq * 4 + r;
                float* u = a.U + ((long)b * n + min(row, n - 1)) * a.ldu;
                float old[CT];
                if (accum) {                     // one batch of unpredicated loads per row (clamped addresses)
#pragma unroll
                    for (int cb = 0; cb < CT; ++cb) old[cb] = u[min(cb * 16 + l15, a.C - 1)];
                }
#pragma unroll
                for (int cb = 0; cb < CT; ++cb) {
                    const int col = cb * 16 + l15;
                    const float v = accum ? acc[rb][cb][r] + a.beta * old[cb] : acc[rb][cb][r];
                    if (row < n && col < a.C) u[col] = v;
                }
            }
        return;
    }
    // fused GraphConv tail (encoders.py:966-972) in registers: a row lives in the 16 lanes of one kq group
    const int G = a.g.G, c01 = G == 2 ? a.g.c0[1] : 0x7fffffff;
    float bias_v[CT];
    int grp[CT];
#pragma unroll
    for (int cb = 0; cb < CT; ++cb) {
        const int col = cb * 16 + l15;
        grp[cb] = col >= c01 ? 1 : 0;
        const float* bp = a.bias.p[grp[cb]];
        bias_v[cb] = (col < a.C && bp) ? bp[col - a.g.c0[grp[cb]]] : 0.f;
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int node = rw0 + rb * 16 + kq * 4 + r;
            const bool rv = node < n;
            const long row = (long)b * n + min(node, n - 1);
            float u[CT];
            float ss[2] = {0.f, 0.f};
            if (a.P) {
#pragma unroll
                for (int cb = 0; cb < CT; ++cb) u[cb] = a.P[row * a.ldv + min(cb * 16 + l15, a.C - 1)];
            } else {
#pragma unroll
                for (int cb = 0; cb < CT; ++cb) u[cb] = 0.f;
            }
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) {
                const int col = cb * 16 + l15;
                const float v = col < a.C ? acc[rb][cb][r] + bias_v[cb] + u[cb] : 0.f;
                u[cb] = v;
                ss[grp[cb]] += v * v;
            }
            float inv[2], mean[2] = {0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float t = agg_team_sum(ss[g]);
                inv[g] = a.normalize ? 1.f / fmaxf(sqrtf(t), AGG_L2_EPS) : 1.f;
            }
            float s1[2] = {0.f, 0.f};
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) {
                const int col = cb * 16 + l15;
                const float v = u[cb] * inv[grp[cb]];
                u[cb] = v;
                if (col < a.C) {
                    if (rv) a.yout.p[grp[cb]][row * a.yout.ld[grp[cb]] + col - a.g.c0[grp[cb]]] = v;
                    s1[grp[cb]] += a.stats_mode == 1 ? fmaxf(v, 0.f) : v;
                }
            }
            if (rv && l15 == 0 && a.invn)
                for (int g = 0; g < G; ++g) a.invn[row * G + g] = inv[g];
            if (a.stats_mode && a.part) {
#pragma unroll
                for (int g = 0; g < 2; ++g) mean[g] = agg_team_sum(s1[g]) / (float)(g < G ? a.g.w[g] : 1);
                float m2[2] = {0.f, 0.f};
#pragma unroll
                for (int cb = 0; cb < CT; ++cb) {
                    const int col = cb * 16 + l15;
                    if (col < a.C) {
                        float v = u[cb];
                        if (a.stats_mode == 1) v = fmaxf(v, 0.f);
                        v -= mean[grp[cb]];
                        m2[grp[cb]] += v * v;
                    }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) m2[g] = agg_team_sum(m2[g]);
                if (rv && l15 == 0)
                    for (int g = 0; g < G; ++g) {
                        a.part[(row * G + g) * 2 + 0] = mean[g];
                        a.part[(row * G + g) * 2 + 1] = m2[g];
                    }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Wide-tile variant for big batches and wide operands (ER: B = 256, n = 1024, C up to 320).
// The 16/32-row panel kernel above re-reads the whole split V operand once per row tile: at n = 1024 that is 32
// reads of V per graph and the L2 traffic (2.4 GB per pass at C = 40) — not HBM — sets its time; at C = 256 the
// same product on the fp32 GEMM costs 137 GFLOP of exact-fp32 MFMA.  Here one workgroup owns 128 rows x ALL
// columns (each wave 32 rows), so V is read n/128 times, and the product runs on the bf16 MFMA against the
// 3-plane exact split (same arithmetic as accumulate_bf16: bit-identical products, fp32 accumulation).
//   * A fragments go global -> registers (a wave's rows are its own; nothing to share), four k-steps deep
//   * the split-V pieces of a k-step (1 KiB each, already in MFMA B-fragment order) are shared by the four
//     waves: register-staged into a double-buffered LDS image, lane-linear (conflict-free b128 both ways),
//     one barrier per step.  No LDS-DMA here on purpose: with a glds in flight hipcc drains vmcnt(0) at every
//     use of an ordinary load (cdna_hip_programming.md §5), which would serialise the A prefetch.
// Runs only when the pack flag says A is bf16-exact; the caller queues a predicated fp32 launch behind it.
#ifndef DP_AGGW_KS
#define DP_AGGW_KS 2      // (tuning: experimental builds override these two.  KS = 4 — 256-byte instead of 128-byte runs per
                          // adjacency row and step — with NA = 2 or 4 measured 157-158 us against 154 at the ER shape: the
                          // pass is not limited by the granularity of its row reads; NA = 6 / 8 (five / seven steps of A in
                          // flight per wave): 161 / 156 us — nor by what it keeps in flight.  537 MB of adjacency in 154 us
                          // is 3.5 TB/s of pure READ traffic, against a device whose copy kernel moves 3.1 read + 3.1 write)
#endif
#ifndef DP_AGGW_NA
#define DP_AGGW_NA 4
#endif
template <int CT>
__global__ __launch_bounds__(256) void k_aggregate_wide(AggArgs a) {
    constexpr int KS = CT <= 4 ? DP_AGGW_KS : CT <= 8 ? 2 : 1; // k32 sub-steps per pipeline step
    constexpr int NP = 3 * CT * KS;              // 1-KiB pieces of split V per step
    constexpr int PPW = (NP + 3) / 4;            // pieces staged per wave
    constexpr int NA = CT <= 4 ? DP_AGGW_NA : CT <= 8 ? 4 : 2; // A register sets (prefetch distance NA - 1 steps; wide steps are long)
    constexpr int NB = CT <= 8 ? 3 : 6;          // split-V fragments kept in flight from LDS ahead of their MFMAs
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    if (__builtin_amdgcn_readfirstlane(*a.pk_flag) != 0) return;
    unsigned short* lds = reinterpret_cast<unsigned short*>(ldsf);
    const int nwg = gridDim.x, tiles = a.tiles;
    int wid;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rm = nwg & 7;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int b = wid / tiles;
    const int n = a.n, np = a.pk_ld;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int rw0 = (wid % tiles) * 128 + wave * 32;  // first row of this wave
    const int steps = (n + 31) / 32;
    const int psteps = (steps + KS - 1) / KS;
    const int K8 = a.vs_k8, CTt = a.vs_ct;
    const unsigned short* Ab = a.pk_A + (long)b * n * np;
    const unsigned short* Vb = a.Vs + (long)b * 3 * CTt * K8 * 128 + lane * 8;
    const unsigned short* arow[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) arow[rb] = Ab + (long)min(rw0 + rb * 16 + l15, n - 1) * np;

    agg_s16x8 areg[NA][KS][2];
    agg_s16x8 vreg[PPW];
    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // columns past np - 8 are clamped (finite duplicates; they meet zero rows of the split V).  k-steps past the
    // end (the rounded-up trip count, the odd half of a two-step stage) read 16 zero bytes instead — the pack
    // flag block: adj_pack clears 256 bytes of it and this kernel only runs while flag[0] == 0 — so the multiply
    // loop needs no branch and the loaded registers no select (a select would pull the wait up to the load)
    const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(a.pk_flag);
    auto load_a = [&](int ps, agg_s16x8 (&dst)[KS][2]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int s32 = ps * KS + ks;
            const int col = min(min(s32, steps - 1) * 32 + kq * 8, np - 8);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                dst[ks][rb] = *reinterpret_cast<const agg_s16x8*>(s32 < steps ? arow[rb] + col : zsrc);
        }
    };
    // no predicates around the staging loads/stores (a branch per piece makes hipcc drain vmcnt(0) at each one):
    // piece ids past NP are clamped to the last piece — a duplicate load and an identical duplicate LDS write
    auto load_v = [&](int ps) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = min(wave + 4 * i, NP - 1);
            const int ks = q / (3 * CT), rem = q % (3 * CT), plane = rem / CT, cb = rem % CT;
            const int s32 = min(ps * KS + ks, steps - 1);
            vreg[i] = *reinterpret_cast<const agg_s16x8*>(
                Vb + (((long)plane * CTt + min(cb, CTt - 1)) * K8 + s32 * 4) * 128);
        }
    };
    auto write_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = min(wave + 4 * i, NP - 1);
            *reinterpret_cast<agg_s16x8*>(lds + (buf * NP + q) * 512 + lane * 8) = vreg[i];
        }
    };
    // `buf` is a compile-time constant at every call site, so each fragment read is one base VGPR + an immediate.
    // Fragment t of a step = (ks, plane, cb) in issue order; NB reads run ahead of the MFMAs that consume them
    // (sched_group_barrier pins that interleave: hipcc otherwise sinks every read to just before its use).
    auto compute = [&](int buf, const agg_s16x8 (&af)[KS][2]) {
        constexpr int T = KS * 3 * CT;
        constexpr int PRE = NB < T ? NB : T;
        const unsigned short* base = lds + buf * NP * 512 + lane * 8;
        auto frag = [&](int t) {
            const int ks = t / (3 * CT), rem = t % (3 * CT), p = 2 - rem / CT, cb = rem % CT;   // lo, mid, hi
            return *reinterpret_cast<const agg_s16x8*>(base + ((ks * 3 + p) * CT + cb) * 512);
        };
        agg_s16x8 ring[NB];
#pragma unroll
        for (int t = 0; t < PRE; ++t) ring[t] = frag(t);
        __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int ks = t / (3 * CT), cb = t % CT;
            const agg_s16x8 bf = ring[t % NB];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(agg_bf16x8, af[ks][rb]),
                                                                      __builtin_bit_cast(agg_bf16x8, bf), acc[rb][cb],
                                                                      0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (t + NB < T) {
                ring[t % NB] = frag(t + NB);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };

    load_v(0);
#pragma unroll
    for (int u = 0; u < NA - 1; ++u) load_a(u, areg[u]);
    write_v(0);
    load_v(min(1, psteps - 1));
    // the trip count is rounded up to a multiple of NA (static register-set indices); stages past the end only
    // move clamped duplicates and multiply zeros
    for (int ps0 = 0; ps0 < psteps; ps0 += NA) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int ps = ps0 + u;
            __syncthreads();              // stage ps is visible; everyone is done reading the other buffer
            write_v((u + 1) & 1);         // NA is even, so the stage parity is u's
            __builtin_amdgcn_sched_barrier(0);   // the staging registers are free before they are refilled
            load_v(min(ps + 2, psteps - 1));
            load_a(ps + NA - 1, areg[(u + NA - 1) % NA]);
            __builtin_amdgcn_sched_barrier(0);
            compute(u & 1, areg[u]);
        }
    }

    aggw_epilogue<CT>(a, acc, b, rw0);
}

__device__ inline unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// ---------------------------------------------------------------------------------------------------------
// Wide operands (C > 128: the pooling products A^T S / A (S dA'^T) with K up to 320 clusters, and the last
// assign-stack layer): 2*CT*3 MFMAs per 32-deep k-step make the kernel MFMA-bound with ONE workgroup per CU
// (accumulators: 2 x CT tiles = up to 160 registers), so nothing but the wave itself can hide its staging.
// Everything the k-loop reads therefore arrives by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ordinary loads in the loop — with one in flight hipcc would drain vmcnt(0) at each use):
//   * split-V pieces (1 KiB, already in B-fragment order): the 3*CT pieces of step s+1 are requested right after
//     the barrier that opens step s, into the other half of a double buffer
//   * A fragments: each lane names its own 16 bytes (row l15, k-slice kq), so the 1-KiB piece lands in exactly
//     the order the wave reads it back (lane-linear both ways, wave-private: no swizzle, no cross-wave hazard);
//     a ring of three steps keeps the HBM stream two steps ahead
//   * one raw s_barrier per step behind a COUNTED s_waitcnt vmcnt(2): V(s), A(s) have landed, A(s+2) stays in
//     flight (issue order per step: V pieces, then A; the count is the two A pieces issued after V).
// All LDS is one array (a second __shared__ object makes hipcc wait vmcnt(0) before every fragment read).
// WAVES = 8 (a 256-row tile, two waves per SIMD) when the accumulators leave room for it (CT <= 16): the V pieces
// are shared by twice the rows — half the LDS-DMA issue cost per MFMA (a glds costs the issuing wave 60-180 cycles,
// MI355X_MICROARCH.md cycle constants) — and one wave's requests hide under its SIMD partner's MFMAs.
template <int CT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_aggregate_wide_dma(AggArgs a) {
    constexpr int NPV = 3 * CT;                  // V pieces per step
    constexpr int PPW = (NPV + WAVES - 1) / WAVES;   // V pieces requested per wave
    constexpr int NB = 6;                        // fragment reads in flight ahead of their MFMAs
    constexpr int VBUF = NPV * 512;              // elements per V buffer
    constexpr int AOFF = 2 * VBUF;               // A ring [3][WAVES][2][512] behind the two V buffers
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    if (__builtin_amdgcn_readfirstlane(*a.pk_flag) != 0) return;
    unsigned short* lds = reinterpret_cast<unsigned short*>(ldsf);
    const int nwg = gridDim.x, tiles = a.tiles;
    int wid;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rm = nwg & 7;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int b = wid / tiles;
    const int n = a.n, np = a.pk_ld;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int rw0 = (wid % tiles) * (32 * WAVES) + wave * 32;
    const int steps = (n + 31) / 32;
    const int K8 = a.vs_k8, CTt = a.vs_ct;
    const unsigned short* Ab = a.pk_A + (long)b * n * np;
    const unsigned short* Vb = a.Vs + (long)b * 3 * CTt * K8 * 128 + lane * 8;
    const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(a.pk_flag);   // 256 zero bytes (flag == 0)
    const unsigned short* arow[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) arow[rb] = Ab + (long)min(rw0 + rb * 16 + l15, n - 1) * np;

    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // requests (every wave issues the same number per step: the vmcnt arithmetic depends on it)
    auto req_v = [&](int s) {
        const int sc = min(s, steps - 1);
        unsigned short* dst = lds + (s & 1) * VBUF;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = min(wave + WAVES * i, NPV - 1);    // clamped duplicate: same bytes to the same place
            const int plane = q / CT, cb = q % CT;
            dma16_raw(Vb + (((long)plane * CTt + min(cb, CTt - 1)) * K8 + sc * 4) * 128, dst + q * 512);
        }
    };
    auto req_a = [&](int s) {
        const int col = min(min(s, steps - 1) * 32 + kq * 8, np - 8);
        unsigned short* dst = lds + AOFF + ((s % 3) * WAVES + wave) * 1024;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) dma16_raw(s < steps ? arow[rb] + col : zsrc, dst + rb * 512);
    };

    req_a(0);
    req_v(0);
    req_a(1);
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // V(s), A(s) landed; A(s+1) may still fly
        __builtin_amdgcn_s_barrier();                          // ... for every wave; all are done with step s-1
        asm volatile("" ::: "memory");
        req_v(s + 1);
        req_a(s + 2);
        // The multiply stream is inline asm: fragment reads (ds_read_b128), counted lgkmcnt waits and MFMAs with the
        // accumulator tied in place in VGPRs, in exactly this order.  Left to the builtins hipcc either sinks every
        // read to just before its use or (with sched_group_barrier) gives each MFMA of the unrolled stream a fresh
        // destination tuple and spends ~8 v_accvgpr moves per MFMA permuting them back; around asm MFMAs it waits
        // lgkmcnt(0).  LDS returns in order, so before MFMA pair t the reads issued after fragment t number
        // min(NB - 1, NPV - 1 - t): that is the wait count.  (The compiler does not know these reads are pending;
        // nothing but the asm below touches `af` / `ring`.)
        const unsigned vaddr = lds_addr(lds + (s & 1) * VBUF + lane * 8);
        const unsigned aaddr = lds_addr(lds + AOFF + ((s % 3) * WAVES + wave) * 1024 + lane * 8);
        agg_s16x8 af[2], ring[NB];
#define DP_LDS_READ(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        DP_LDS_READ(af[0], aaddr, 0);
        DP_LDS_READ(af[1], aaddr, 1024);
#pragma unroll
        for (int t = 0; t < NB; ++t) DP_LDS_READ(ring[t], vaddr, ((2 - t / CT) * CT + t % CT) * 1024);
#pragma unroll
        for (int t = 0; t < NPV; ++t) {
            const int cb = t % CT;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NB - 1 < NPV - 1 - t ? NB - 1 : NPV - 1 - t));
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                             : "+v"(acc[rb][cb])
                             : "v"(af[rb]), "v"(ring[t % NB]));
            if (t + NB < NPV)
                DP_LDS_READ(ring[t % NB], vaddr, ((2 - (t + NB) / CT) * CT + (t + NB) % CT) * 1024);
        }
#undef DP_LDS_READ
    }
    // the asm MFMAs are invisible to the hazard recogniser: let the last results retire before VALU reads them
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the look-ahead requests before LDS is released
    aggw_epilogue<CT>(a, acc, b, rw0);
}

constexpr int AGGW_MAX_C = 320;
static int aggw_ct(int C) {
    const int ct = (C + 15) / 16;
    return ct <= 8 ? ct : (ct + 1) & ~1;
}
// the wide kernel pays when it fills the chip (>= 2 workgroups per CU) or when the operand is too wide for the
// panel kernel; it needs the packed adjacency (n >= 128)
static bool aggw_usable(const PackedAdj* pk, const unsigned short* vs, int B, int n, int C) {
    if (!pk || !vs || n < 128 || C < 1 || C > AGGW_MAX_C) return false;
    const int force = knobs().agg_wide;   // tests: 0 never, 1 always
    if (force >= 0) return force != 0;
    return C > 128 || (long)B * ((n + 127) / 128) >= 512;
}

template <int CT>
static void launch_aggw(Seq& q, const AggArgs& a, int B) {
    constexpr int KS = CT <= 4 ? DP_AGGW_KS : CT <= 8 ? 2 : 1;
    constexpr size_t lds = (size_t)2 * 3 * CT * KS * 1024;
    static_assert(lds <= 160 * 1024, "wide aggregation LDS");
    static DynLdsOnce attr;
    if (lds > 64 * 1024)
        ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_aggregate_wide<CT>), (int)lds, "k_aggregate_wide");
    if (!q.ok()) return;
    AggArgs aa = a;
    aa.tiles = (a.n + 127) / 128;
    aa.vs_ct = (a.C + 15) / 16;
    hipLaunchKernelGGL((k_aggregate_wide<CT>), dim3(aa.tiles * B), dim3(256), lds, q.stream, aa);
}
template <int CT, int WAVES>
static void launch_aggw_dma(Seq& q, const AggArgs& a, int B) {
    constexpr size_t lds = ((size_t)2 * 3 * CT * 512 + 3 * WAVES * 1024) * sizeof(unsigned short);
    static_assert(lds <= 160 * 1024, "wide aggregation LDS");
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_aggregate_wide_dma<CT, WAVES>), (int)lds,
                   "k_aggregate_wide_dma");
    if (!q.ok()) return;
    AggArgs aa = a;
    aa.tiles = (a.n + 32 * WAVES - 1) / (32 * WAVES);
    aa.vs_ct = (a.C + 15) / 16;
    hipLaunchKernelGGL((k_aggregate_wide_dma<CT, WAVES>), dim3(aa.tiles * B), dim3(WAVES * 64), lds, q.stream, aa);
}
static void dispatch_wide(Seq& q, const AggArgs& a, int B) {
    switch (aggw_ct(a.C)) {
#define W(T) case T: launch_aggw<T>(q, a, B); break;
#define D(T, WV) case T: launch_aggw_dma<T, WV>(q, a, B); break;
        W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) D(10, 8) D(12, 8) D(14, 8) D(16, 8) D(18, 8) D(20, 4)
#undef W
#undef D
        default: break;
    }
}

static int agg_ct(int C) { return (C + 15) / 16; }
static size_t agg_panel_floats(bool trans, int n, int RT) {
    const int kpanel = trans ? n : (n < AGG_KP ? n : AGG_KP);
    const int segs = (kpanel + 255) / 256;
    const size_t panel = trans ? (size_t)((n + 15) / 16) * 16 * RT : (size_t)RT * (segs * 256 + 4);
    const size_t pk = ((size_t)RT * (((n + 511) / 512) * 512 + 8) * 2 + 3) / 4;     // bf16 panel, in floats
    return ((panel > pk ? panel : pk) + 3) & ~size_t(3);
}
static size_t agg_lds_bytes(bool trans, int n, int CT, int RT, int NW = 4) {
    const size_t panel = agg_panel_floats(trans, n, RT);
    const size_t red = (size_t)(NW + 1) * RT * (CT * 16 + 1);
    return (panel > red ? panel : red) * sizeof(float);    // the reduction area overlays the panel
}
// 16-row tiles when 32-row tiles would leave CUs idle (< 1 workgroup per CU): twice the workgroups.  Otherwise 32
// rows: every row tile re-reads the whole split V from L2 (147 KB at the DD shape against a 16-32 KB adjacency
// panel), so halving the tile count halves the dominant on-chip traffic (DD: 0.424 -> 0.418 ms per step).
static int agg_row_tile(int B, int n, int C, bool trans) {
    if (agg_lds_bytes(trans, n, agg_ct(C), 32) > 160 * 1024) return 16;   // 32 rows do not fit
    const int force = knobs().agg_rt;   // tuning knob
    if (force == 16 || force == 32) return force;
    return ((long)((n + 31) / 32) * B >= 256) ? 32 : 16;
}

bool aggregate_supported(const float* A, int n, int C, bool trans) {
    // C > 128 is not a panel shape: a 16/32-row panel re-reads the whole V operand per row tile (25 GB of L2
    // traffic per pass at the ER shape).  Wide operands go to k_aggregate_wide_dma when the adjacency is packed and
    // bf16-exact, to the tiled fp32 GEMM otherwise.
    if (n < 4 || n % 4 != 0 || C < 1 || C > 128) return false;
    if ((reinterpret_cast<uintptr_t>(A) & 15) != 0) return false;
    return agg_lds_bytes(trans, n, agg_ct(C), 16) <= 160 * 1024;
}

template <bool TRANS, int CT, int RT, int NW>
static void launch_agg_rt(Seq& q, const AggArgs& a, int B) {
    // (Sized for the fp32 panel although the bf16 loop needs half of it — 66 KB at n = 500, two workgroups per CU.
    // Measured with the bf16-only size, four per CU: DD step 0.3447 vs 0.3449 ms, probe launch 8.1 us both: co-residency
    // is not what this launch waits for.)
    const size_t lds = agg_lds_bytes(TRANS, a.n, CT, RT, NW);
    static DynLdsOnce attr;   // per instantiation: allow > 64 KiB of dynamic LDS
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_aggregate<TRANS, CT, RT, NW>), 160 * 1024, "k_aggregate");
    if (!q.ok()) return;
    AggArgs aa = a;
    aa.tiles = (a.n + RT - 1) / RT;
    aa.vs_ct = (a.C + 15) / 16;
    aa.dbg = knobs().agg_debug;     // phase-ablation mask: DP_STAMP diagnostic builds only, 0 in the product
    hipLaunchKernelGGL((k_aggregate<TRANS, CT, RT, NW>), dim3(aa.tiles * B), dim3(NW * 64), lds, q.stream, aa);
}
template <bool TRANS, int CT>
static void launch_agg(Seq& q, const AggArgs& a, int B) {
    // (8 waves per 32-row tile, NW = 8: measured slower at DD, 9.5 vs 8.0 us per launch and 0.427 vs 0.407 ms per
    // step -- the launch is bound by the L2 burst of V fragments, not by per-wave latency, and the extra waves only
    // add barrier and reduction work.  64-row tiles: also slower, 9.1 us per launch, 0.412 vs 0.398 ms.)
    if (agg_row_tile(B, a.n, a.C, TRANS) == 32) {
        launch_agg_rt<TRANS, CT, 32, 4>(q, a, B);
    } else launch_agg_rt<TRANS, CT, 16, 4>(q, a, B);     // (64-row tiles: measured slower at DD, 0.437 vs 0.417 ms)
}

template <bool TRANS>
static void dispatch_ct(Seq& q, const AggArgs& a, int B) {
    switch (agg_ct(a.C)) {
        case 1: launch_agg<TRANS, 1>(q, a, B); break;
        case 2: launch_agg<TRANS, 2>(q, a, B); break;
        case 3: launch_agg<TRANS, 3>(q, a, B); break;
        case 4: launch_agg<TRANS, 4>(q, a, B); break;
        case 5: launch_agg<TRANS, 5>(q, a, B); break;
        case 6: launch_agg<TRANS, 6>(q, a, B); break;
        case 7: launch_agg<TRANS, 7>(q, a, B); break;
        default: launch_agg<TRANS, 8>(q, a, B); break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_adj_pack: one pass over the fp32 adjacency -> bf16 copies of A and A^T (rows padded to `ld` = a multiple
// of 8 elements, zero filled) + a device flag that ends up non-zero iff some entry is NOT exactly
// representable in bf16 (low 16 mantissa bits set).  0/1 adjacency (graph_sampler.py:26) is always exact.
// Every thread asks for its four 16-byte quads (16 column quads x 16 rows per pass, clamped, no predicate) before it
// looks at any of them; rows go out as 8-byte bf16 quads.  (One conditional 4-byte load per element, as first written,
// was 16 dependent round trips per thread: 11.9 us for the DD batch.)  n % 4 == 0, ld % 8 == 0.
typedef float agg_f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short agg_u16x4 __attribute__((ext_vector_type(4)));
// side job of both pack kernels: the launch's workgroups clear `zn16` 16-byte words at zp between them
__device__ inline void pack_side_zero(uint4* zp, long zn16) {
    if (!zp) return;
    const long nwg = (long)gridDim.x * gridDim.y * gridDim.z;
    const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const long per = (zn16 + nwg - 1) / nwg;
    const long end = min(zn16, (wg + 1) * per);
    for (long i = wg * per + threadIdx.x; i < end; i += 256) zp[i] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_adj_pack(const float* A, unsigned short* P, unsigned short* Pt, int* flag,
                                                  int n, int ld, uint4* zp, long zn16) {
    __shared__ __attribute__((aligned(8))) unsigned short tile[64][68];
    pack_side_zero(zp, zn16);
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* Ab = A + (long)b * n * n;
    unsigned short* Pb = P + (long)b * n * ld;
    unsigned short* Ptb = Pt + (long)b * n * ld;
    const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c = c0 + 4 * tq;
    agg_f32x4_u v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
        v[p] = *reinterpret_cast<const agg_f32x4_u*>(Ab + (long)min(r0 + tr + 16 * p, n - 1) * n + min(c, n - 4));
    bool bad = false;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + tr + 16 * p;
        const bool in = r < n && c < n;
        agg_u16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned bits = in ? __float_as_uint(v[p][j]) : 0u;
            bad |= (bits & 0xFFFFu) != 0;
            h[j] = (unsigned short)(bits >> 16);
        }
        *reinterpret_cast<agg_u16x4*>(&tile[tr + 16 * p][4 * tq]) = h;
        if (r < n && c < ld) *reinterpret_cast<agg_u16x4*>(Pb + (long)r * ld + c) = h;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = c0 + tr + 16 * p, cc = r0 + 4 * tq;       // transposed tile: row = source column
        if (r < n && cc < ld) {
            agg_u16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = tile[4 * tq + j][tr + 16 * p];
            *reinterpret_cast<agg_u16x4*>(Ptb + (long)r * ld + cc) = h;
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// any n (the public entry point takes any): one element per thread and pass
__global__ __launch_bounds__(256) void k_adj_pack_any(const float* A, unsigned short* P, unsigned short* Pt, int* flag,
                                                      int n, int ld, uint4* zp, long zn16) {
    __shared__ unsigned short tile[64][66];
    pack_side_zero(zp, zn16);
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* Ab = A + (long)b * n * n;
    unsigned short* Pb = P + (long)b * n * ld;
    unsigned short* Ptb = Pt + (long)b * n * ld;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    bool bad = false;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const unsigned raw = __float_as_uint(Ab[(long)min(r, n - 1) * n + min(c, n - 1)]);
        const unsigned bits = (r < n && c < n) ? raw : 0u;
        bad |= (bits & 0xFFFFu) != 0;
        const unsigned short h = (unsigned short)(bits >> 16);
        tile[i][tx] = h;
        if (r < n && c < ld) Pb[(long)r * ld + c] = h;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int r = c0 + i, c = r0 + tx;       // transposed tile
        if (r < n && c < ld) Ptb[(long)r * ld + c] = tile[tx][i];
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

void adj_pack(Seq& q, const float* A, unsigned short* P, unsigned short* Pt, int* flag, int B, int n, int ld,
              bool flag_zeroed, void* zero_p, size_t zero_bytes) {
    if (!q.ok()) return;
    if (!flag_zeroed) zero_fill(q, flag, 256);   // (a kernel: captured memset nodes misbehave on replay, see zero_fill)
    const int t = (ld + 63) / 64;
    if (B <= 0 || n <= 0 || (zero_p && ((reinterpret_cast<uintptr_t>(zero_p) | zero_bytes) & 15))) {
        if (zero_p) zero_fill(q, zero_p, zero_bytes);
        zero_p = nullptr;
        if (B <= 0 || n <= 0) return;
    }
    uint4* zp = static_cast<uint4*>(zero_p);
    const long zn16 = zero_p ? (long)(zero_bytes / 16) : 0;
    if (n % 4 == 0)
        hipLaunchKernelGGL(k_adj_pack, dim3(t, (n + 63) / 64, B), dim3(256), 0, q.stream, A, P, Pt, flag, n, ld, zp,
                           zn16);
    else
        hipLaunchKernelGGL(k_adj_pack_any, dim3(t, (n + 63) / 64, B), dim3(256), 0, q.stream, A, P, Pt, flag, n, ld,
                           zp, zn16);
    q.check_launch("adj_pack");
}
int adj_pack_ld(int n) { return (n + 7) & ~7; }
bool adj_pack_supported(int n, int C) {
    const bool off = knobs().no_pack;      // ablation: fp32 adjacency passes everywhere
    if (off) return false;
    // worth it only for big levels; the bf16 panel of a 32-row tile must fit LDS beside the reduction area
    return n >= 128 && C >= 1 && C <= AGGW_MAX_C &&
           ((size_t)16 * (((n + 511) / 512) * 512 + 8) * 2 + 5 * 16 * 129 * 4 <= 156 * 1024);
}

// k_split3: V [B, n, C] fp32 -> three bf16 planes hi, mid, lo with hi + mid + lo == V exactly, in the layout
// the bf16 MFMA B-operand wants: Vs[b][plane][cb][k8][c][j] = plane(V[b][8*k8 + j][16*cb + c]), zero padded in k
// (to a multiple of 32) and c (to 16*CT).  One 16-byte store per (plane, k8, c).
__device__ inline unsigned short bf16_rn(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);     // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
}
__global__ __launch_bounds__(256) void k_split3(const float* V, int ldv, unsigned short* Vs, int n, int C, int CT,
                                                int K8) {
    const int b = blockIdx.y;
    const int item = blockIdx.x * 256 + threadIdx.x;          // (cb, k8, c)
    if (item >= CT * K8 * 16) return;
    const int c = item & 15, k8 = (item >> 4) % K8, cb = (item >> 4) / K8;
    const int col = cb * 16 + c;
    agg_s16x8 hi, mid, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k8 * 8 + j;
        float v = 0.f;
        if (k < n && col < C) v = V[((long)b * n + k) * ldv + col];
        const unsigned short h = bf16_rn(v);
        const float r1 = v - __uint_as_float((unsigned)h << 16);
        const unsigned short m = bf16_rn(r1);
        const float r2 = r1 - __uint_as_float((unsigned)m << 16);
        const unsigned short l = bf16_rn(r2);
        hi[j] = (short)h; mid[j] = (short)m; lo[j] = (short)l;
    }
    unsigned short* base = Vs + (long)b * 3 * CT * K8 * 128;
    *reinterpret_cast<agg_s16x8*>(base + ((((long)0 * CT + cb) * K8 + k8) * 16 + c) * 8) = hi;
    *reinterpret_cast<agg_s16x8*>(base + ((((long)1 * CT + cb) * K8 + k8) * 16 + c) * 8) = mid;
    *reinterpret_cast<agg_s16x8*>(base + ((((long)2 * CT + cb) * K8 + k8) * 16 + c) * 8) = lo;
}
size_t split3_elems(int B, int n, int C) {
    const int CT = (C + 15) / 16, K8 = ((n + 31) / 32) * 4;
    return (size_t)B * 3 * CT * K8 * 128;
}
static void split3(Seq& q, const float* V, int ldv, unsigned short* Vs, int B, int n, int C) {
    const int CT = (C + 15) / 16, K8 = ((n + 31) / 32) * 4;
    hipLaunchKernelGGL(k_split3, dim3((CT * K8 * 16 + 255) / 256, B), dim3(256), 0, q.stream, V, ldv, Vs, n, C, CT, K8);
    q.check_launch("split3");
}

// true when aggregate()/aggregate_rownorm_fwd() will take the packed path for this shape (so a producer may
// emit the split operand itself)
bool aggregate_packed_usable(const float* A, int n, int C) {
    return adj_pack_supported(n, C) && (C > 128 || aggregate_supported(A, n, C, false));
}

static void fill_packed(AggArgs& a, const PackedAdj* pk, bool trans, const unsigned short* vs, int n) {
    a.pk_A = trans ? pk->At : pk->A;
    a.pk_ld = pk->ld;
    a.pk_flag = pk->flag;
    a.Vs = vs;
    a.vs_k8 = ((n + 31) / 32) * 4;
}

static bool agg_prefers_split_gemm(const float* A, const float* V, float* U, int B, int n, int C, int ldv, int ldu,
                                  bool trans, float beta) {
    if (C < 48) return false;                 // narrow operands: the pass is bound by reading A, the panel kernel's case
    GemmDesc d{A, V, U, nullptr, n, C, n, n, ldv, ldu, (long)n * n, (long)n * ldv, (long)n * ldu, trans, false, 1.f, beta, 0};
    return gemm_split_usable(d, B, 1);
}
// U[b] (ldu) = op(A[b]) V[b] (+ beta U[b]);  falls back to the generic GEMM for shapes the panel kernel
// does not take (n not a multiple of 4, C > 128, unaligned A).  With `pk` (a packed copy of A from adj_pack) and a
// scratch buffer `vs` (split3_elems) the bf16 path is taken when the device flag says A is bf16-exact.
void aggregate(Seq& q, const float* A, const float* V, int ldv, float* U, int ldu, int B, int n, int C, bool trans,
               float beta, const PackedAdj* pk, unsigned short* vs, bool vs_ready) {
    if (!q.ok()) return;
    const bool packed = pk && vs && adj_pack_supported(n, C);
    const bool panel = aggregate_supported(A, n, C, trans);
    const bool wide = packed && aggw_usable(pk, vs, B, n, C);
    if (!packed && agg_prefers_split_gemm(A, V, U, B, n, C, ldv, ldu, trans, beta)) {
        // a general (pooled, weighted) adjacency at a big batch: both operands are general fp32, which is the split-bf16
        // GEMM's case — the fp32 panel kernel ran the 84-column level-1 passes of the ER shape at 30 TFLOP/s
        bgemm(q, A, V, U, nullptr, B, n, C, n, n, ldv, ldu, (long)n * n, (long)n * ldv, (long)n * ldu, trans, false, 1.f,
              beta, 0);
        return;
    }
    AggArgs a{};
    a.A = A; a.V = V; a.ldv = ldv; a.n = n; a.C = C; a.U = U; a.ldu = ldu; a.beta = beta;
    if (wide) {
        if (!vs_ready) split3(q, V, ldv, vs, B, n, C);
        fill_packed(a, pk, trans, vs, n);
        dispatch_wide(q, a, B);
        q.check_launch("aggregate_wide");
        // fp32 fallback for adjacency that is not bf16-exact, gated on the device flag (no host sync)
        if (panel) {
            AggArgs f{};
            f.A = A; f.V = V; f.ldv = ldv; f.n = n; f.C = C; f.U = U; f.ldu = ldu; f.beta = beta;
            f.run_if = pk->flag;
            if (trans) dispatch_ct<true>(q, f, B); else dispatch_ct<false>(q, f, B);
            q.check_launch("aggregate");
        } else {
            const int* keep = q.pred;
            q.pred = pk->flag;
            bgemm(q, A, V, U, nullptr, B, n, C, n, n, ldv, ldu, (long)n * n, (long)n * ldv, (long)n * ldu, trans, false,
                  1.f, beta, 0);
            q.pred = keep;
        }
        return;
    }
    if (!panel) {
        bgemm(q, A, V, U, nullptr, B, n, C, n, n, ldv, ldu, (long)n * n, (long)n * ldv, (long)n * ldu, trans, false,
              1.f, beta, 0);
        return;
    }
    if (packed) {
        if (!vs_ready) split3(q, V, ldv, vs, B, n, C);
        fill_packed(a, pk, trans, vs, n);
    }
    if (trans) dispatch_ct<true>(q, a, B); else dispatch_ct<false>(q, a, B);
    q.check_launch("aggregate");
}

// Fused forward GraphConv tail: y = l2norm(A V (+ P) + bias) per column group, written to yout, with the
// apply_bn partial statistics.  Returns false (nothing launched) when the shape needs the generic path.
bool aggregate_rownorm_fwd(Seq& q, const float* A, const float* V, int ldv, const float* P, GroupCPtrs bias,
                           RowGroups g, GroupPtrs yout, float* invn, float* part, int B, int n, int normalize,
                           int stats_mode, const PackedAdj* pk, unsigned short* vs, bool vs_ready) {
    const int C = g.c0[g.G - 1] + g.w[g.G - 1];
    if (!aggregate_supported(A, n, C, false)) return false;
    if (!(pk && vs && adj_pack_supported(n, C)) && agg_prefers_split_gemm(A, V, nullptr, B, n, C, ldv, C, false, 0.f))
        return false;                          // (the caller's two-launch path takes the split GEMM in aggregate())
    if (!q.ok()) return true;
    AggArgs a{};
    a.A = A; a.V = V; a.ldv = ldv; a.n = n; a.C = C; a.U = nullptr;
    a.P = P; a.bias = bias; a.g = g; a.yout = yout; a.invn = invn; a.part = part;
    a.normalize = normalize; a.stats_mode = stats_mode;
    const bool packed = pk && vs && adj_pack_supported(n, C);
    if (packed && !vs_ready) split3(q, V, ldv, vs, B, n, C);
    if (packed && aggw_usable(pk, vs, B, n, C)) {
        AggArgs w = a;
        fill_packed(w, pk, false, vs, n);
        dispatch_wide(q, w, B);
        q.check_launch("aggregate_wide_rownorm");
        a.run_if = pk->flag;               // fp32 fallback, runs only for adjacency that is not bf16-exact
    } else if (packed) {
        fill_packed(a, pk, false, vs, n);
    }
    dispatch_ct<false>(q, a, B);
    q.check_launch("aggregate_rownorm_fwd");
    return true;
}

#ifdef DP_STAMP
extern "C" __attribute__((visibility("default"))) int dp_debug_agg_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_agg_stamps), sizeof(unsigned long long) * 32);
}
#endif

}  // namespace dp
