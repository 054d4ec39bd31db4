// Level 0 of the DiffPool forward as ONE persistent launch (round 3).
//
//   gcn_forward of the embed + assign stacks          encoders.py:1054-1081, 1254, 1269   (GraphConv :962-974, apply_bn :1048-1052)
//   max readout of the embedding                      encoders.py:1257
//   S = softmax(Linear(Za)) * mask                    encoders.py:1273-1275
//   X' = S^T Z,  A' = S^T A S  (as Tt = A^T S, A' = Tt^T S)                               encoders.py:1278-1279
//
// Rounds 1-2 ran this as 11 dependent launches (111 us of the 344 us DD step): every kernel a ~4 us floor plus cold
// memory round trips, every adjacency pass re-reading its panel.  Here a workgroup owns RB rows of ONE graph for the
// whole level: it converts its rows of the fp32 adjacency to bf16 ONCE into LDS (and writes the packed A / A^T copies
// the backward pass reads), and keeps them there for every GraphConv pass; row-local work (the x W transforms, bias,
// l2-normalise, ReLU, BatchNorm apply, the assign head, softmax, mask) never leaves the workgroup.
//
// What crosses workgroups goes through global memory behind barriers of this launch:
//   * the operand of an aggregation U = A_rows . P needs ALL rows of P of the graph: each workgroup writes the exact
//     3-plane bf16 split of its rows (the layout dp_agg.hip reads), then a PER-GRAPH barrier (T arrivals);
//   * apply_bn couples the graphs, but only at equal node index (statistics per node index over batch x features): row
//     partials (mean, M2) travel as TAGGED 16-byte entries that the B workgroups holding the same row block poll — no
//     barrier at all (l0_poll_entries; the first builds used a grid-wide, then a per-row-block barrier here);
//   * X' = S^T Z and A' = Tt^T S contract over the node index: per-workgroup partial tiles, a per-graph barrier, then
//     the graph's workgroups sum disjoint slices in block order (deterministic); the max readout likewise.
// Barrier counters only grow inside a launch (the e-th crossing waits for e * arrivals; nothing is reset between
// crossings and never by the host: the last workgroup to FINISH the launch clears every count, so a launch that timed
// out leaves a clean block too); waits are bounded (one that gives up raises DP_DEVERR_BARRIER and a device word the
// prediction head turns into NaN logits) and need every workgroup resident: the launcher only takes this path when
// B * T <= CUs with one 512-thread workgroup per CU.
//
// Memory model (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed by another CU's stores): every byte
// another workgroup of the launch reads is stored write-through (`sc1`, raw buffer builtins, aux 16) and every storing
// wave drains vmcnt before the workgroup barrier that precedes its arrival (cdna guide, Guideline 16 R1).  The regions
// are WRITE-ONCE per launch — one per aggregation pass / BatchNorm layer / combine, never reused — and no workgroup
// touches a line of one before the barrier that publishes it, so no L1 or L2 of the launch can hold a stale copy
// (caches start a launch invalidated) and the consumers read them with plain cached loads.
//
// Exactness: as dp_agg.hip — a bf16-exact adjacency (0/1 graphs) is multiplied as bf16 x (hi + mid + lo) planes on the
// bf16 MFMA with fp32 accumulation, every product exact; a graph whose adjacency is NOT bf16-exact takes an fp32 MFMA
// loop that reads the fp32 adjacency from global memory (correct, slow; decided per graph on the device).
#include "dp_common.h"
#include <utility>
#include <vector>

namespace dp {

#ifdef DP_STAMP
// diagnostic build only (csrc/build.sh DP_STAMP=1, tools/l0_stamps.py): shader-clock stamps of three workgroups
__device__ unsigned long long g_l0_stamps[3][64];
#define L0_STAMP(i)                                                                                    \
    do {                                                                                               \
        if (threadIdx.x == 0 && (wid == 0 || wid == (int)gridDim.x / 2 || wid == (int)gridDim.x - 1)) \
            g_l0_stamps[wid == 0 ? 0 : (wid == (int)gridDim.x - 1 ? 2 : 1)][i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define L0_STAMP(i) \
    do {            \
    } while (0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int L0_NT = 512;            // threads per workgroup
constexpr int L0_NW = 8;              // waves: (k quarter) x (column half) in the aggregations
constexpr int L0_TEAMS = L0_NT / 16;
constexpr int L0_NK = 4;              // a 16-lane team holds a row group of <= 64 columns in 4 registers per lane
constexpr int L0_BPAIRS = 4;          // BatchNorm combine: <= 64 graphs (4 partial pairs per lane)
constexpr float L0_L2_EPS = 1e-12f;
constexpr float L0_BN_EPS = 1e-5f;
constexpr int L0_SPIN_LIMIT = 1 << 22;

// barrier block (ints; every word that is polled or added to sits on a 64-byte line of its own)
constexpr int BAR_GCOUNT = 0, BAR_DONE = 32, BAR_GFLAG = 48, BAR_ERR = 64, BAR_SEQ = 96, BAR_GRAPH0 = 128;
constexpr int BAR_GSTRIDE = 48;       // per graph: +0 count, +16 generation, +32 "adjacency not bf16-exact"

// ---- write-through / L1-bypassing access to what other workgroups of this launch write or read
__device__ __forceinline__ int ag_ld(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ag_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct L0Args {
    Level0Fwd f;
    int T, RB;                     // row blocks per graph, rows per block (16 * MI)
    int ldp;                       // bf16 elements per LDS adjacency row (8 mod 128)
    int K8;                        // k8 groups of the split operand: ceil(N / 32) * 4
    int steps;                     // 32-deep k-steps: ceil(N / 32)
    int epad;                      // K*D + K*K padded to a multiple of 4
    long vs_off[DP_MAX_LAYERS + 1]; // element offset of pass p's region in f.vs (passes 0..L-1: the layers' operands, L: S)
    int lds_act[2];                // float offsets of the activation rows [RB][ldz[g]] in LDS
    int lds_scr;                   // float offset of the scratch region
    int scr_floats;
    int* dev_err;
    int spin_limit, target_bias;
    long zero_n16;
};

// C[M x N] = op(A)[M x K] . op(B)[K x N], both operands in LDS, fp32 MFMA 16x16x4, tiles dealt to the waves round
// robin (dp_small.hip lds_mma2: running fragment pointers, only the partial k-step is masked).  Rows / columns past
// M / N are clamped duplicates whose results nobody stores.
template <bool TA, bool TB, typename Store>
__device__ inline void l0_mma(const float* A, int lda, const float* B, int ldb, int M, int N, int K, Store store,
                              int wave_shift = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = ((threadIdx.x >> 6) + L0_NW - wave_shift % L0_NW) % L0_NW;
    const int l15 = lane & 15, kq = lane >> 4;
    const int tm = (M + 15) / 16, tn = (N + 15) / 16;
    const int sa4 = 4 * (TA ? lda : 1), sb4 = 4 * (TB ? 1 : ldb);
    for (int t = wave; t < tm * tn; t += L0_NW) {
        const int tr = t / tn, tc = t - tr * tn;
        const int i = tr * 16 + l15, j = tc * 16 + l15;
        const float* ap = (TA ? A + min(i, M - 1) : A + min(i, M - 1) * lda) + kq * (sa4 >> 2);
        const float* bp = (TB ? B + min(j, N - 1) * ldb : B + min(j, N - 1)) + kq * (sb4 >> 2);
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        int k = 0;
        // 16-deep batches, software-pipelined over two register sets: the eight LDS reads of the next batch are in
        // flight under the four MFMAs of this one (with one set the loop was LDS-latency-bound: ~250 cycles per batch
        // against 128 of MFMA issue)
        float pa[4], pb[4], qa[4], qb[4];
        auto ld = [&](float (&xa)[4], float (&xb)[4]) {
            xa[0] = ap[0]; xa[1] = ap[sa4]; xa[2] = ap[2 * sa4]; xa[3] = ap[3 * sa4];
            xb[0] = bp[0]; xb[1] = bp[sb4]; xb[2] = bp[2 * sb4]; xb[3] = bp[3 * sb4];
            ap += 4 * sa4;
            bp += 4 * sb4;
        };
        auto mm = [&](const float (&xa)[4], const float (&xb)[4]) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0], xb[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1], xb[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[2], xb[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[3], xb[3], acc1, 0, 0, 0);
        };
        if (k + 16 <= K) ld(pa, pb);
        while (k + 16 <= K) {
            const bool more = k + 32 <= K;
            if (more) ld(qa, qb);
            mm(pa, pb);
            k += 16;
            if (!more) break;
            const bool more2 = k + 32 <= K;
            if (more2) ld(pa, pb);
            mm(qa, qb);
            k += 16;
            if (!more2) break;
        }
        for (; k + 4 <= K; k += 4) {
            const float a0 = ap[0], b0 = bp[0];
            ap += sa4;
            bp += sb4;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        }
        if (k < K) {
            const bool kin = k + kq < K;
            const float a0 = kin ? ap[0] : 0.f, b0 = kin ? bp[0] : 0.f;      // (reads past K stay inside the LDS region)
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tr * 16 + kq * 4 + r;
            if (row < M && j < N) store(row, j, acc0[r] + acc1[r]);
        }
    }
}

// ---- barriers.  Every thread calls; returns false once a wait of this workgroup has given up (sticky).
// Counters only ever grow inside a launch (the last workgroup to FINISH clears them): the e-th graph barrier of a
// launch waits for the graph's counter to reach e * T.  `epoch` = crossings so far, kept by the caller.
struct L0Epoch {
    int graph;
};
template <typename Args>
__device__ inline bool l0_barrier(const Args& a, int b, int* sflag /*LDS: [0] ok, [1] failed (sticky)*/, L0Epoch& ep) {
    ep.graph += 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = sflag[1] ? 0 : 1;
        int* gc = a.f.bar + BAR_GRAPH0 + b * BAR_GSTRIDE;
        int* pollw = gc;
        const int target = ep.graph * a.T + a.target_bias;
        ag_add(gc, 1);
        if (ok) {
            int it = 0;
            while (ag_ld(pollw) < target) {
                if (++it > a.spin_limit) {
                    ok = 0;
                    ag_st(a.f.bar + BAR_ERR, 1);
                    dev_err_raise(a.dev_err, DP_DEVERR_BARRIER);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        sflag[0] = ok;
        if (!ok) sflag[1] = 1;
    }
    __syncthreads();
    return sflag[0] != 0;
}

// ---- BatchNorm exchange without a barrier: tagged entries (the scheme of RCCL's LL protocol).
// A row's partial pair travels as ONE 16-byte write-through store {v0, tag, v1, tag}; tag = (launch sequence number,
// layer), different in every launch and layer and never 0 (the sequence number lives in the barrier block, read by
// everybody at kernel start and counted up by the last workgroup to finish).  A reader polls the entries it needs with
// 16-byte sc1 loads until both tags of each are this launch's: the data is its own "ready" flag — no arrival counter,
// no acknowledgement wait on the producer's side, and one memory round trip on the consumer's instead of two (poll the
// counter, then load).  Each 8-byte half carries its own tag, so an entry torn at 8 bytes cannot be mistaken for a
// whole one.  Bounded like the barriers.
// NI items per thread, L0_BPAIRS graph slots per item: entry (item j, slot u) at byte offset off[j] + 16 * min(tl + 16 u, B - 1)
template <int NI, typename Args>
__device__ __forceinline__ bool l0_poll_entries(const Args& a, ScBuf buf, const unsigned (&off)[NI], int B, unsigned tag,
                                                float (&v0)[NI][L0_BPAIRS], float (&v1)[NI][L0_BPAIRS], int* sflag) {
    const int tl = threadIdx.x & 15;
    const unsigned want = tag ^ (a.target_bias ? 0x40000000u : 0u);      // (test knob: a tag nobody writes)
    int it = 0;
    for (;;) {
        bool ready = true;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int u = 0; u < L0_BPAIRS; ++u) {
                if (16 * u < B) {                                               // (uniform)
                    const u32x4 q = sc_ld16(buf, off[j] + 16u * (unsigned)min(tl + 16 * u, B - 1));
                    v0[j][u] = __uint_as_float(q[0]);
                    v1[j][u] = __uint_as_float(q[2]);
                    ready = ready && q[1] == want && q[3] == want;
                } else {
                    v0[j][u] = 0.f;
                    v1[j][u] = 0.f;
                }
            }
        }
        if (__all(ready)) return true;
        if (++it > a.spin_limit || sflag[1]) {
            if (!sflag[1]) {
                sflag[1] = 1;
                ag_st(a.f.bar + BAR_ERR, 1);
                dev_err_raise(a.dev_err, DP_DEVERR_BARRIER);
            }
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// Bulk copy global -> LDS of `count` floats (4-byte aligned source): every thread asks for all its 16-byte quads before
// it writes any (one memory round trip for the whole region, not one per loop iteration).  QN quads per thread at most
// (count <= QN * 4 * 512); the <= 3 floats past the last whole quad go one by one.  zero_to: LDS floats [count, zero_to)
// are cleared.
template <int QN>
__device__ __forceinline__ void l0_copy_in(float* dst, const float* src, int count, int zero_to = 0) {
    const int nq = count >> 2;
    float tail = 0.f;
    const int te = (nq << 2) + (int)threadIdx.x;
    if (te < count) tail = src[te];
    for (int base = 0; base < nq; base += QN * L0_NT) {
        f32x4_u q[QN];
#pragma unroll
        for (int u = 0; u < QN; ++u)
            q[u] = *reinterpret_cast<const f32x4_u*>(src + 4 * min(base + u * L0_NT + (int)threadIdx.x, nq - 1));
#pragma unroll
        for (int u = 0; u < QN; ++u) {
            const int e4 = base + u * L0_NT + (int)threadIdx.x;
            if (e4 < nq) *reinterpret_cast<f32x4*>(dst + 4 * e4) = q[u];        // (dst is 16-byte aligned)
        }
    }
    if (te < count) dst[te] = tail;
    for (int e = count + (int)threadIdx.x; e < zero_to; e += L0_NT) dst[e] = 0.f;
}

// ---- the exact 3-plane bf16 split of PT [rows][ct] (LDS) into the graph's Vs block (dp_agg.hip layout
// Vs[plane][cb][k8][c][j]), k8 groups [k8_0, k8_0 + nk8) where local row lr = (k8 - k8_0) * 8; rows >= nrows and columns
// >= ct are written as zeros.  One 16-byte store per (plane, cb, k8, c).
__device__ inline void l0_write_split(ScBuf vsb, const float* PT, int ct, int CTt, int K8, int k8_0, int nk8, int nrows) {
    const long pl = (long)CTt * K8 * 128;                   // elements per plane
    const int items = nk8 * CTt * 16;
    for (int item = threadIdx.x; item < items; item += L0_NT) {
        const int c = item & 15, t = item >> 4;
        const int k8l = t / CTt, cb = t - k8l * CTt;
        const int vc = cb * 16 + c, lr = k8l * 8;
        u16x8 h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (vc < ct && lr + j < nrows) ? PT[(lr + j) * ct + min(vc, ct - 1)] : 0.f;
            unsigned short hh, mm, ll;
            bf16_split3(v, hh, mm, ll);
            h[j] = hh; m[j] = mm; l[j] = ll;
        }
        const long o = vs_index(0, CTt, K8, cb, k8_0 + k8l, c, 0);
        sc_st16(vsb, (unsigned)(o * 2), __builtin_bit_cast(u32x4, h));
        sc_st16(vsb, (unsigned)((o + pl) * 2), __builtin_bit_cast(u32x4, m));
        sc_st16(vsb, (unsigned)((o + 2 * pl) * 2), __builtin_bit_cast(u32x4, l));
    }
}

// ---- aggregation  acc += Ablk[RB x N] . V[N x cols cb0*16 ..),  Ablk bf16 in LDS, V as 3 bf16 planes from global (sc1).
// Wave (kh = wave & 3, ch = wave >> 2): k-steps kh, kh + 4, ...; column tiles cb0 .. cb0 + CTH - 1 (clamped duplicates
// beyond the operand's last tile: the caller ignores them).
template <int MI, int CTH>
__device__ __forceinline__ void l0_agg_bf16(const unsigned short* Alds, int ldp, const unsigned short* vsp, int CTt, int K8,
                                            int steps, int rot, int cb0, f32x4 (&acc)[MI][CTH]) {
    const int lane = threadIdx.x & 63;
    const int kh = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3;
    const int l15 = lane & 15, kq = lane >> 4;
    s16x8 f0[3][CTH], f1[3][CTH];
    // The blocks of a graph walk the k-steps from different starting points (`rot`): the operand was just written
    // through to memory, so the first block to touch a line pays the miss and the others find it in the XCD's L2 —
    // started together on the same lines, every block would sit at the per-CU miss rate (~10 B / cycle) for the whole
    // operand.  (A k-step is a term of the sum: any order gives the same exact products, summed in a fixed order per block.)
    auto eff = [&](int step) {
        const int s2 = min(step, steps - 1) + rot;
        return s2 >= steps ? s2 - steps : s2;
    };
    auto load_b = [&](int step, s16x8 (&dst)[3][CTH]) {
        const int st = eff(step);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int cbi = 0; cbi < CTH; ++cbi) {
                const int cb = min(cb0 + cbi, CTt - 1);
                const long o = ((((long)p * CTt + cb) * K8 + st * 4 + kq) * 16 + l15) * 8;
                dst[p][cbi] = *reinterpret_cast<const s16x8*>(vsp + o);
            }
    };
    auto mma = [&](int step, const s16x8 (&bf)[3][CTH]) {
        s16x8 av[MI];
#pragma unroll
        for (int rb = 0; rb < MI; ++rb)
            av[rb] = *reinterpret_cast<const s16x8*>(Alds + (rb * 16 + l15) * ldp + eff(step) * 32 + kq * 8);
#pragma unroll
        for (int p = 2; p >= 0; --p)          // lo, mid, hi: small terms first
#pragma unroll
            for (int rb = 0; rb < MI; ++rb)
#pragma unroll
                for (int cbi = 0; cbi < CTH; ++cbi)
                    acc[rb][cbi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[rb]),
                                                                           __builtin_bit_cast(bf16x8, bf[p][cbi]),
                                                                           acc[rb][cbi], 0, 0, 0);
    };
    if constexpr (CTH <= 2) {
        // narrow halves: the fragments of FOUR k-steps (all of a DD-sized graph's share) are requested before the first
        // multiply — the pass is a chain of memory round trips, not of MFMAs
        s16x8 f2[3][CTH], f3[3][CTH];
        for (int step = kh; step < steps; step += 16) {
            load_b(step, f0);
            load_b(step + 4, f1);
            load_b(step + 8, f2);
            load_b(step + 12, f3);
            mma(step, f0);
            if (step + 4 < steps) mma(step + 4, f1);
            if (step + 8 < steps) mma(step + 8, f2);
            if (step + 12 < steps) mma(step + 12, f3);
        }
    } else {
        load_b(kh, f0);
        load_b(kh + 4, f1);
        for (int step = kh; step < steps; step += 8) {
            mma(step, f0);
            if (step + 4 < steps) {
                load_b(step + 8, f0);
                mma(step + 4, f1);
                load_b(step + 12, f1);
            }
        }
    }
}

// The same product for a graph whose adjacency is not bf16-exact: fp32 MFMA, op(A) rows straight from the fp32 input
// in global memory (element (i, k) at Ag[i * rs + k * ks]: rs = N, ks = 1 for A; rs = 1, ks = N for A^T), V rebuilt
// exactly from its planes (hi + mid + lo == v).  Slow; only weighted adjacency ever comes here.
template <int MI, int CTH>
__device__ __forceinline__ void l0_agg_f32(const float* Ag, long rs, long ks, int nrows, int N, const unsigned short* vsp,
                                           int CTt, int K8, int steps, int cb0, f32x4 (&acc)[MI][CTH]) {
    const int lane = threadIdx.x & 63;
    const int kh = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3;
    const int l15 = lane & 15, kq = lane >> 4;
    for (int step = kh; step < steps; step += 4) {
        float bv[CTH][8];
#pragma unroll
        for (int cbi = 0; cbi < CTH; ++cbi) {
            const int cb = min(cb0 + cbi, CTt - 1);
            u16x8 pl3[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const long o = ((((long)p * CTt + cb) * K8 + step * 4 + kq) * 16 + l15) * 8;
                pl3[p] = *reinterpret_cast<const u16x8*>(vsp + o);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                bv[cbi][j] = (__uint_as_float((unsigned)pl3[0][j] << 16) + __uint_as_float((unsigned)pl3[1][j] << 16)) +
                             __uint_as_float((unsigned)pl3[2][j] << 16);
        }
#pragma unroll
        for (int rb = 0; rb < MI; ++rb) {
            const int i = rb * 16 + l15;
            float av[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = step * 32 + kq * 8 + j;
                const float t = Ag[(long)min(i, nrows - 1) * rs + (long)min(k, N - 1) * ks];
                av[j] = (i < nrows && k < N) ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int cbi = 0; cbi < CTH; ++cbi)
                    acc[rb][cbi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[cbi][j], acc[rb][cbi], 0, 0, 0);
        }
    }
}

// One aggregation pass into the k-quarter slots red[kh][RB][ctp] (the caller sums the four quarters).
template <int MI, int CTH, typename Args>
__device__ __forceinline__ void l0_aggregate_pass(const Args& a, const unsigned short* Alds, bool exact, const float* Ag,
                                                  long rs, long ks, int nrows, const unsigned short* vsb, int CTt, float* red,
                                                  int ctp, int rot) {
    constexpr int RB = MI * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = wave & 3, ch = wave >> 2;
    const int l15 = lane & 15, kq = lane >> 4;
    const int cta = (CTt + 1) >> 1;                        // column tiles of half 0
    const int cb0 = ch ? cta : 0, ncb = ch ? CTt - cta : cta;
    f32x4 acc[MI][CTH];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < CTH; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ncb > 0) {
        if (exact) l0_agg_bf16<MI, CTH>(Alds, a.ldp, vsb, CTt, a.K8, a.steps, rot, cb0, acc);
        else l0_agg_f32<MI, CTH>(Ag, rs, ks, nrows, a.f.N, vsb, CTt, a.K8, a.steps, cb0, acc);
    }
#pragma unroll
    for (int rb = 0; rb < MI; ++rb)
#pragma unroll
        for (int cbi = 0; cbi < CTH; ++cbi)
            if (cbi < ncb) {
                if ((cb0 + cbi) * 16 + l15 < ctp) {       // (a slot row is ctp floats: >= the operand's width, maybe < the tiles')
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(kh * RB + rb * 16 + kq * 4 + r) * ctp + (cb0 + cbi) * 16 + l15] = acc[rb][cbi][r];
                }
            }
}
template <int MI, typename Args>
__device__ __forceinline__ void l0_aggregate(const Args& a, const unsigned short* Alds, bool exact, const float* Ag, long rs,
                                             long ks, int nrows, const unsigned short* vsb, int CTt, float* red, int ctp, int rot) {
    // tiles of THIS wave's column half (wave-uniform): the narrower half of an odd tile count takes the narrower
    // instantiation instead of multiplying a clamped duplicate of the last tile
    const int cta_ = (CTt + 1) >> 1;
    const int cth = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >> 2) ? CTt - cta_ : cta_;
    if (cth <= 1) l0_aggregate_pass<MI, 1>(a, Alds, exact, Ag, rs, ks, nrows, vsb, CTt, red, ctp, rot);
    else if (cth == 2) l0_aggregate_pass<MI, 2>(a, Alds, exact, Ag, rs, ks, nrows, vsb, CTt, red, ctp, rot);
    else l0_aggregate_pass<MI, 3>(a, Alds, exact, Ag, rs, ks, nrows, vsb, CTt, red, ctp, rot);
}

// 16-byte-quad staging of RB rows of a bf16 operand [*, ld] into the LDS adjacency block (columns >= ld and
// rows >= nrows become zeros up to the k-step padding): issue() asks for everything, commit() writes LDS.
template <int MI>
struct L0RowStage {
    static constexpr int NQ = 2 * MI;                      // rows tr, tr + 8, ... of one 512-column segment
    u32x4 q[NQ];
};
template <int MI>
__device__ __forceinline__ void l0_stage_issue(L0RowStage<MI>& s, const unsigned short* rows, int ld, int nrows, int seg) {
    const int tq = threadIdx.x & 63, tr = threadIdx.x >> 6;
    const int c8 = seg * 64 + tq;
#pragma unroll
    for (int u = 0; u < L0RowStage<MI>::NQ; ++u) {
        const int row = min(tr + 8 * u, max(nrows - 1, 0));
        const int cc = min(c8 * 8, ld - 8);
        s.q[u] = *reinterpret_cast<const u32x4*>(rows + (long)row * ld + cc);
    }
}
template <int MI>
__device__ __forceinline__ void l0_stage_commit(const L0RowStage<MI>& s, unsigned short* Alds, int ldp, int ld, int nrows, int seg) {
    const int tq = threadIdx.x & 63, tr = threadIdx.x >> 6;
    const int c8 = seg * 64 + tq;
#pragma unroll
    for (int u = 0; u < L0RowStage<MI>::NQ; ++u) {
        const int row = tr + 8 * u;
        const bool in = row < nrows && c8 * 8 < ld;
        const u32x4 v = in ? s.q[u] : (u32x4){0u, 0u, 0u, 0u};
        if (c8 * 8 + 8 <= ldp) *reinterpret_cast<u32x4*>(Alds + row * ldp + c8 * 8) = v;
    }
}

template <int MI>
__global__ __launch_bounds__(L0_NT) void k_level0_fwd(L0Args a) {
    constexpr int RB = MI * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Level0Fwd& f = a.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tl = tid & 15, team = tid >> 4;
    const int N = f.N, G = f.G, L = f.L;
    // XCD-aware work mapping (as k_aggregate): every XCD gets one contiguous run of (graph, row block) items, so a
    // graph's blocks share one L2 for the split operand they all read
    int wid;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rm = nwg & 7;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int b = wid / a.T, rb = wid - b * a.T;
    const int r0 = rb * RB;
    const int nrows = min(RB, N - r0);                      // valid rows of this block (>= 1)
    const int nb = f.num_nodes ? min(f.num_nodes[b], N) : N;
    const int rot = (rb * 4) % a.steps;                     // this block's starting k-step in the aggregations
    L0_STAMP(0);

    unsigned short* Alds = reinterpret_cast<unsigned short*>(lds);              // [RB][ldp] bf16
    float* ACT0 = lds + a.lds_act[0];                      // [RB][ldz[0]]  this block's rows of Ze
    float* ACT1 = lds + a.lds_act[1];                      // [RB][ldz[1]]  ... of Za
    float* SCR = lds + a.lds_scr;
    float* BIAS = lds + a.lds_scr + a.scr_floats;          // every layer's biases [l][g][64], then assign_pred's [64]
    int* sflag = reinterpret_cast<int*>(BIAS + (2 * DP_MAX_LAYERS + 1) * 64);   // [0] ok, [1] failed, [2] block flag, [3] last
    if (tid < 4) sflag[tid] = 0;
    L0Epoch ep{0};
    const unsigned seq = (unsigned)f.bar[BAR_SEQ];          // launch sequence number (stable until the last workgroup finishes)

    // Exchange buffers are WRITE-ONCE per launch (one region per pass): a reader can only ever fetch a line after its
    // writers are done, so no cache of this launch can hold a stale copy and the reads are ordinary cached loads (the
    // blocks of a graph share the XCD's L2); the writes are write-through (sc1) and drained before the barrier.
    auto vs_wr = [&](int pass, int CTt) {
        return sc_buf(f.vs + a.vs_off[pass] + (long)b * 3 * CTt * a.K8 * 128, (size_t)3 * CTt * a.K8 * 128 * 2);
    };
    auto vs_rd = [&](int pass, int CTt) { return f.vs + a.vs_off[pass] + (long)b * 3 * CTt * a.K8 * 128; };

    // ------------------------------------------------------------------ phase 0
    // side jobs: the pooled level's barrier block, last call's error word
    if (wid == 0) {
        if (f.next_bar && tid < 64) f.next_bar[tid] = 0;
        if (tid == 64) ag_st(f.bar + BAR_ERR, 0);
    }
    L0_STAMP(1);
    const int din0_0 = f.st[0].dims[0], din0_1 = G == 2 ? f.st[1].dims[0] : 0;
    // (a) my rows of the fp32 adjacency -> bf16: LDS block, packed A rows, exactness.  The layer-0 inputs, weights and
    // every bias are asked for right behind the adjacency quads, so they arrive under the same burst.
    bool bad = false;
    float* const X0s0 = SCR;
    float* const X0s1 = X0s0 + RB * din0_0;
    float* const W0s0 = X0s1 + RB * din0_1;
    float* const W0s1 = W0s0 + ((din0_0 * f.st[0].dims[1] + 3) & ~3);
    float* const PT = W0s1 + (G == 2 ? ((din0_1 * f.st[1].dims[1] + 3) & ~3) : 0);
    const auto side_loads = [&]() {
        l0_copy_in<4>(X0s0, f.x0[0] + ((long)b * N + r0) * din0_0, nrows * din0_0, RB * din0_0);
        l0_copy_in<4>(W0s0, f.params + f.st[0].w_off[0], din0_0 * f.st[0].dims[1]);
        if (G == 2) {
            l0_copy_in<4>(X0s1, f.x0[1] + ((long)b * N + r0) * din0_1, nrows * din0_1, RB * din0_1);
            l0_copy_in<4>(W0s1, f.params + f.st[1].w_off[0], din0_1 * f.st[1].dims[1]);
        }
        // biases: one 64-float slot per (layer, stack), zeros where a layer has none
        for (int e = tid; e < (2 * L + 1) * 64; e += L0_NT) {
            const int slot = e >> 6, c = e & 63;
            float v0 = 0.f;
            if (slot < 2 * L) {
                const int l = slot >> 1, g = slot & 1;
                if (g < G && f.st[g].b_off[l] >= 0 && c < f.st[g].dims[l + 1]) v0 = f.params[f.st[g].b_off[l] + c];
            } else if (G == 2 && f.bp_off >= 0 && c < f.K) {
                v0 = f.params[f.bp_off + c];
            }
            BIAS[(slot < 2 * L ? slot : 2 * DP_MAX_LAYERS) * 64 + c] = v0;
        }
    };
    if (f.A) {
        const float* Ab = f.A + ((long)b * N + r0) * N;
        unsigned short* Pb = f.pkA + ((long)b * N + r0) * f.pk_ld;
        const int tq = tid & 127, tr = tid >> 7;           // 128 column quads x 4 row lanes
        const int segs = (a.steps * 32 + 511) / 512;
        for (int seg = 0; seg < segs; ++seg) {
            const int c = seg * 512 + 4 * tq;
            f32x4_u v[4 * MI];
#pragma unroll
            for (int u = 0; u < 4 * MI; ++u)
                v[u] = *reinterpret_cast<const f32x4_u*>(Ab + (long)min(tr + 4 * u, nrows - 1) * N + min(c, N - 4));
            if (seg == 0) side_loads();
#pragma unroll
            for (int u = 0; u < 4 * MI; ++u) {
                const int row = tr + 4 * u;
                const bool in = row < nrows && c < N;
                u16x4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned bits = in ? __float_as_uint(v[u][j]) : 0u;
                    bad |= (bits & 0xFFFFu) != 0;
                    h[j] = (unsigned short)(bits >> 16);
                }
                if (c < a.ldp) *reinterpret_cast<u16x4*>(Alds + row * a.ldp + c) = h;
                if (row < nrows && c < f.pk_ld) *reinterpret_cast<u16x4*>(Pb + (long)row * f.pk_ld + c) = h;
            }
        }
    } else {
        // the adjacency arrives packed (dp_encoder_forward_packed): my bf16 rows go to LDS as they are
        L0RowStage<MI> q;
        const unsigned short* rows = f.pkA + ((long)b * N + r0) * f.pk_ld;
        l0_stage_issue<MI>(q, rows, f.pk_ld, nrows, 0);
        side_loads();
        l0_stage_commit<MI>(q, Alds, a.ldp, f.pk_ld, nrows, 0);
        const int segs = (a.steps * 32 + 511) / 512;
        for (int seg = 1; seg < segs; ++seg) {
            l0_stage_issue<MI>(q, rows, f.pk_ld, nrows, seg);
            l0_stage_commit<MI>(q, Alds, a.ldp, f.pk_ld, nrows, seg);
        }
    }
    L0_STAMP(2);
    // side job: clear the backward accumulators (behind the adjacency burst: the stores drain under the phases below)
    if (f.zero_p) {
        uint4* zp = reinterpret_cast<uint4*>(f.zero_p);
        const long nwg = gridDim.x, per = (a.zero_n16 + nwg - 1) / nwg;
        const long end = min(a.zero_n16, ((long)wid + 1) * per);
        for (long i = (long)wid * per + tid; i < end; i += L0_NT) zp[i] = make_uint4(0, 0, 0, 0);
    }
    if (__any(bad) && lane == 0) sflag[2] = 1;
    lds_barrier();
    L0_STAMP(3);
    const bool blk_bad = sflag[2] != 0;
    if (blk_bad && tid == 0) {
        ag_st(f.bar + BAR_GRAPH0 + b * BAR_GSTRIDE + 32, 1);
        ag_st(f.bar + BAR_GFLAG, 1);
    }
    // (c) my column strip of A^T: rows k of the packed transpose get my RB rows as 8-element (16-byte) pieces
    if (f.A) {
        const ScBuf atb = sc_buf(f.pkAt + (long)b * N * f.pk_ld, (size_t)N * f.pk_ld * 2);
        constexpr int P8 = RB / 8;
        for (int e = tid; e < N * P8; e += L0_NT) {
            const int k = e / P8, i8 = e - k * P8;
            if (r0 + i8 * 8 < f.pk_ld) {
                u16x8 h;
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = Alds[(i8 * 8 + j) * a.ldp + k];
                sc_st16(atb, (unsigned)(((long)k * f.pk_ld + r0 + i8 * 8) * 2), __builtin_bit_cast(u32x4, h));
            }
        }
    }
    L0_STAMP(4);
    // (d) P_0 = [x_e W_e | x_a W_a] of my rows, then its split
    int ct = f.st[0].dims[1] + (G == 2 ? f.st[1].dims[1] : 0);
    for (int g = 0; g < G; ++g) {
        const int dout = f.st[g].dims[1], c0 = g ? f.st[0].dims[1] : 0;
        l0_mma<false, false>(g ? X0s1 : X0s0, g ? din0_1 : din0_0, g ? W0s1 : W0s0, dout, RB, dout, g ? din0_1 : din0_0,
                             [&](int r, int c, float v) { PT[r * ct + c0 + c] = v; }, g * 3);
    }
    lds_barrier();
    L0_STAMP(5);
    const int k8_0 = r0 / 8;
    const int nk8 = min(rb == a.T - 1 ? a.K8 - k8_0 : RB / 8, a.K8 - k8_0);
    {
        const int CTt = (ct + 15) / 16;
        l0_write_split(vs_wr(0, CTt), PT, ct, CTt, a.K8, k8_0, nk8, nrows);
    }
    L0_STAMP(6);
    bool ok = l0_barrier(a, b, sflag, ep);
    L0_STAMP(7);
    const bool exact = !f.A || ag_ld(f.bar + BAR_GRAPH0 + b * BAR_GSTRIDE + 32) == 0;      // (packed in: bf16 IS the adjacency)
    const float* Arows = f.A + ((long)b * N + r0) * N;             // fp32 fallback operands
    const float* Acols = f.A + (long)b * N * N + r0;

    // ------------------------------------------------------------------ the GraphConv layers
    constexpr int ITEMS = (RB * 2 + L0_TEAMS - 1) / L0_TEAMS;      // (row, group) items per team
    for (int l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        const int w0 = f.st[0].dims[l + 1], w1 = G == 2 ? f.st[1].dims[l + 1] : 0;
        const int wmax = max(w0, w1);
        const int CTt = (ct + 15) / 16;
        const int ctp = CTt * 16 + 1;
        float* red = SCR;
        // U = A_rows . P_l   (four k-quarter partials)
        l0_aggregate<MI>(a, Alds, exact, Arows, N, 1, nrows, vs_rd(l, CTt), CTt, red, ctp, rot);
        L0_STAMP(8 + 8 * l);
        // next layer's weights go to LDS behind the reduce slots now: their latency hides under the tail
        int wn0 = 0, wn1 = 0;
        float* const Wn0 = red + 4 * RB * ctp;
        if (!last) {
            wn0 = f.st[0].dims[l + 1] * f.st[0].dims[l + 2];
            wn1 = G == 2 ? f.st[1].dims[l + 1] * f.st[1].dims[l + 2] : 0;
        }
        float* const Wn1 = Wn0 + ((wn0 + 3) & ~3);
        if (!last) {
            l0_copy_in<4>(Wn0, f.params + f.st[0].w_off[l + 1], wn0);
            if (G == 2) l0_copy_in<4>(Wn1, f.params + f.st[1].w_off[l + 1], wn1);
        }
        lds_barrier();
        L0_STAMP(9 + 8 * l);
        // tail: + bias, l2-normalise, save, BatchNorm partials — one 16-lane team per (row, group); y also goes to the
        // layer's slice of the LDS activation rows (BatchNorm rewrites it in place after the exchange)
        const ScBuf partw = sc_buf(f.part + (long)l * f.B * N * G * 4, (size_t)f.B * N * G * 4 * sizeof(float));
        const unsigned bn_tag = (seq << 4) | (unsigned)(l + 1);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            float yv[L0_NK];
            const int it = team + j * L0_TEAMS;
            const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
            const bool on = it < RB * G && r < nrows;
            const long row = (long)b * N + min(r0 + r, N - 1);
            const int wg = g ? w1 : w0, c0gg = g ? w0 : 0;
            const float* bias = BIAS + (2 * l + g) * 64;
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                yv[k] = 0.f;
                if (16 * k < wmax) {                       // (uniform: no lane of the workgroup has a column there)
                    const int c = min(tl + 16 * k, wg - 1);
                    const int o = r * ctp + c0gg + c;
                    float v = ((red[o] + red[RB * ctp + o]) + (red[2 * RB * ctp + o] + red[3 * RB * ctp + o])) + bias[c];
                    v = tl + 16 * k < wg ? v : 0.f;
                    yv[k] = v;
                    ss += v * v;
                }
            }
            ss = row16_sum(ss);
            // x / max(||x||, eps)  (F.normalize, encoders.py:972): v_rsq_f32 is within 1 ulp of 1 / sqrt
            const float inv = ss > L0_L2_EPS * L0_L2_EPS ? __builtin_amdgcn_rsqf(ss) : 1.f / L0_L2_EPS;
            float s1 = 0.f;
            float* yg = last ? f.Z[g] + row * f.ldz[g] + f.coff[g][l] : f.Y[l] + row * ct + c0gg;
            float* act = (g ? ACT1 : ACT0) + r * f.ldz[g] + f.coff[g][l];
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                if (16 * k < wmax) {
                    const int c = tl + 16 * k;
                    yv[k] *= inv;
                    if (on && c < wg) yg[c] = yv[k];
                    if (it < RB * G && c < wg) act[c] = r < nrows ? yv[k] : 0.f;
                    s1 += fmaxf(yv[k], 0.f);
                }
            }
            if (on && tl == 0) f.invn[l][row * G + g] = inv;
            if (!last && f.bn) {
                s1 = row16_sum(s1);
                const float mean = s1 / (float)wg;
                float m2 = 0.f;
#pragma unroll
                for (int k = 0; k < L0_NK; ++k) {
                    if (16 * k < wmax) {
                        const float v = (tl + 16 * k < wg) ? fmaxf(yv[k], 0.f) - mean : 0.f;
                        m2 += v * v;
                    }
                }
                m2 = row16_sum(m2);
                if (on && tl == 0) {               // [node][group][graph]: a node's B entries are contiguous for the reader
                    const long po = (((long)min(r0 + r, N - 1) * G + g) * f.B + b) * 16;
                    sc_st16(partw, (unsigned)po, sc_tagged(mean, m2, bn_tag));
                }
            }
        }
        L0_STAMP(10 + 8 * l);
        if (last) break;
        // ---- apply_bn: every graph's partials of my node indices (tagged entries, polled), then x = (relu(y) - mu) * rstd
        if (!f.bn) lds_barrier();
        L0_STAMP(11 + 8 * l);
        const int ctn = f.st[0].dims[l + 2] + (G == 2 ? f.st[1].dims[l + 2] : 0);
        float* PTn = Wn1 + ((wn1 + 3) & ~3);               // behind the weights (which sit behind the reduce slots)
        {
            // all the partials of all my items in one polled round, then the combines
            float pm[ITEMS][L0_BPAIRS], pq[ITEMS][L0_BPAIRS];
            if (f.bn) {
                unsigned off[ITEMS];
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int it = team + j * L0_TEAMS;
                    const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
                    off[j] = (unsigned)((((long)min(r0 + r, N - 1) * G + g) * f.B) * 16);
                }
                ok = l0_poll_entries<ITEMS>(a, partw, off, f.B, bn_tag, pm, pq, sflag) && ok;
            }
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int it = team + j * L0_TEAMS;
                const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
                const bool on = it < RB * G && r < nrows;
                const int node = min(r0 + r, N - 1);
                const long row = (long)b * N + node;
                const int wg = g ? w1 : w0;
                float mu = 0.f, rstd = 1.f;
                if (f.bn) {
                    float sm = 0.f;
#pragma unroll
                    for (int u = 0; u < L0_BPAIRS; ++u) sm += (tl + 16 * u < f.B) ? pm[j][u] : 0.f;
                    mu = row16_sum(sm) / (float)f.B;
                    float s2 = 0.f;
#pragma unroll
                    for (int u = 0; u < L0_BPAIRS; ++u) {
                        const float d = pm[j][u] - mu;
                        s2 += (tl + 16 * u < f.B) ? pq[j][u] + (float)wg * d * d : 0.f;
                    }
                    const float var = row16_sum(s2) / ((float)f.B * (float)wg);
                    rstd = __builtin_amdgcn_rsqf(var + L0_BN_EPS);
                    if (!ok) mu = rstd = __builtin_nanf("");
                    if (on && b == 0 && tl == 0) {
                        f.stats[l][((long)node * G + g) * 2] = mu;
                        f.stats[l][((long)node * G + g) * 2 + 1] = rstd;
                    }
                }
                float* xg = f.Z[g] + row * f.ldz[g] + f.coff[g][l];
                float* act = (g ? ACT1 : ACT0) + r * f.ldz[g] + f.coff[g][l];
#pragma unroll
                for (int k = 0; k < L0_NK; ++k) {
                    if (16 * k < wmax) {
                        const int c = tl + 16 * k;
                        const float xv = (fmaxf(act[min(c, wg - 1)], 0.f) - mu) * rstd;
                        if (on && c < wg) xg[c] = xv;
                        if (it < RB * G && c < wg) act[c] = r < nrows ? xv : 0.f;
                    }
                }
            }
        }
        lds_barrier();
        L0_STAMP(12 + 8 * l);
        // ---- P_{l+1} = [x_e W_e | x_a W_a] of my rows and its split
        for (int g = 0; g < G; ++g) {
            const int din = f.st[g].dims[l + 1], dout = f.st[g].dims[l + 2], c0 = g ? f.st[0].dims[l + 2] : 0;
            l0_mma<false, false>((g ? ACT1 : ACT0) + f.coff[g][l], f.ldz[g], g ? Wn1 : Wn0, dout, RB, dout, din,
                                 [&](int r, int c, float v) { PTn[r * ctn + c0 + c] = v; }, g * 3);
        }
        lds_barrier();
        L0_STAMP(13 + 8 * l);
        {
            const int CTn = (ctn + 15) / 16;
            l0_write_split(vs_wr(l + 1, CTn), PTn, ctn, CTn, a.K8, k8_0, nk8, nrows);
        }
        L0_STAMP(14 + 8 * l);
        ok = l0_barrier(a, b, sflag, ep) && ok;
        L0_STAMP(15 + 8 * l);
        ct = ctn;
    }
    lds_barrier();                                       // the last layer's rows are in ACT0 / ACT1

    const int K = f.K, D = f.ldz[0];
    float* XP = SCR;                                       // partial X' [K][D] | A' [K][K] (after the A^T S pass)
    if (G == 2 && K > 0) {
        // ------------------------------------------------------------------ assign head + softmax + mask
        const int Da = f.ldz[1];
        const int CTk = (K + 15) / 16, ctpk = CTk * 16 + 1;
        float* TL = SCR + 4 * RB * ctpk;                   // [RB][K]  T rows        (beyond the reduce slots of the A^T S pass)
        float* SL = TL + RB * K;                           // [RB][K]  logits, then S rows
        float* WP = SCR;                                   // [K][Da]  (the reduce area is free now)
        l0_copy_in<4>(WP, f.params + f.wp_off, K * Da);
        lds_barrier();
        L0_STAMP(40);
        {
            const float* bp = BIAS + 2 * DP_MAX_LAYERS * 64;
            l0_mma<false, true>(ACT1, Da, WP, Da, RB, K, Da, [&](int r, int c, float v) { SL[r * K + c] = v + bp[c]; });
        }
        lds_barrier();
        L0_STAMP(41);
        for (int r = team; r < RB; r += L0_TEAMS) {
            const int node = r0 + r;
            const bool valid = node < nb;
            float lv[L0_NK];
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) lv[k] = SL[r * K + min(tl + 16 * k, K - 1)];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) m = fmaxf(m, lv[k]);
            m = row16_max(m);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                lv[k] = expf(lv[k] - m);
                sum += (tl + 16 * k < K) ? lv[k] : 0.f;
            }
            sum = row16_sum(sum);
            const float rinv = 1.f / sum;
            const long row = (long)b * N + min(node, N - 1);
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                const int c = tl + 16 * k;
                if (c < K) {
                    const float v = (valid && node < N) ? lv[k] * rinv : 0.f;
                    SL[r * K + c] = v;
                    if (node < N) {
                        f.S[row * K + c] = v;
                        if (f.S2) f.S2[row * K + c] = v;
                    }
                }
            }
        }
        lds_barrier();
        L0_STAMP(42);
        l0_write_split(vs_wr(L, CTk), SL, K, CTk, a.K8, k8_0, nk8, nrows);
        L0_STAMP(43);
        // my rows of A^T replace my rows of A (every wave is long past the last GraphConv pass; the strips were complete
        // at barrier 0): ordinary loads, asked for in front of the barrier, written to LDS behind it
        {
            const int segs = (a.steps * 32 + 511) / 512;
            const unsigned short* atp = f.pkAt + ((long)b * N + r0) * f.pk_ld;
            L0RowStage<MI> atq;
            l0_stage_issue<MI>(atq, atp, f.pk_ld, nrows, 0);
            ok = l0_barrier(a, b, sflag, ep) && ok;
            L0_STAMP(44);
            l0_stage_commit<MI>(atq, Alds, a.ldp, f.pk_ld, nrows, 0);
            for (int seg = 1; seg < segs; ++seg) {
                l0_stage_issue<MI>(atq, atp, f.pk_ld, nrows, seg);
                l0_stage_commit<MI>(atq, Alds, a.ldp, f.pk_ld, nrows, seg);
            }
        }
        lds_barrier();
        L0_STAMP(45);
        // ------------------------------------------------------------------ T = A^T S (my rows), partial X', A'
        l0_aggregate<MI>(a, Alds, exact, Acols, 1, N, nrows, vs_rd(L, CTk), CTk, SCR, ctpk, rot);
        L0_STAMP(46);
        lds_barrier();
        L0_STAMP(47);
        for (int r = team; r < RB; r += L0_TEAMS) {
            const long row = (long)b * N + min(r0 + r, N - 1);
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                const int c = tl + 16 * k;
                const int o = r * ctpk + min(c, K - 1);
                const float v = (SCR[o] + SCR[RB * ctpk + o]) + (SCR[2 * RB * ctpk + o] + SCR[3 * RB * ctpk + o]);
                if (c < K) {
                    TL[r * K + c] = r < nrows ? v : 0.f;
                    if (r < nrows) f.Tt[row * K + c] = v;
                }
            }
        }
        lds_barrier();
        L0_STAMP(48);
        float* AP = XP + K * D;
        l0_mma<true, false>(SL, K, ACT0, D, K, D, RB, [&](int i, int j, float v) { XP[i * D + j] = v; });
        l0_mma<true, false>(TL, K, SL, K, K, K, RB, [&](int i, int j, float v) { AP[i * K + j] = v; }, 4);
        for (int e = K * D + K * K + tid; e < a.epad; e += L0_NT) XP[e] = 0.f;
        lds_barrier();
        L0_STAMP(49);
        {
            const ScBuf xpb = sc_buf(f.xpart + ((long)b * a.T + rb) * a.epad, (size_t)a.epad * 4);
            for (int e4 = tid; e4 < a.epad / 4; e4 += L0_NT)
                sc_st16(xpb, (unsigned)(e4 * 16), *reinterpret_cast<const u32x4*>(XP + e4 * 4));
        }
    }
    // ------------------------------------------------------------------ max readout: partial over my rows
    if (f.do_max) {
        const int nm = f.mask_readout ? max(min(nb - r0, nrows), 0) : nrows;   // rows that take part
        const ScBuf mpb = sc_buf(f.mpart + ((long)b * a.T + rb) * f.rw * 2, (size_t)f.rw * 2 * 4);
        // (column, row group) per thread: eight groups of RB / 8 rows, merged in row order (strict >: lowest row wins ties)
        float* MV = XP + a.epad;                           // [8][rw] best value, [8][rw] row (behind the partial products)
        int* MI_ = reinterpret_cast<int*>(MV + 8 * f.rw);
        constexpr int RG = RB / 8;
        for (int c0 = 0; c0 < f.rw; c0 += 64) {
            const int c = c0 + (tid & 63), grp = tid >> 6;
            float best = -INFINITY;
            int bi = -1;
            if (c < f.rw) {
#pragma unroll
                for (int rr = 0; rr < RG; ++rr) {
                    const int r = grp * RG + rr;
                    const float v = ACT0[r * D + f.zoff + c];
                    if (r < nm && v > best) {
                        best = v;
                        bi = r0 + r;
                    }
                }
                MV[grp * f.rw + c] = best;
                MI_[grp * f.rw + c] = bi;
            }
        }
        lds_barrier();
        for (int c = tid; c < f.rw; c += L0_NT) {
            float best = -INFINITY;
            int bi = -1;
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                const float v = MV[grp * f.rw + c];
                const int i = MI_[grp * f.rw + c];
                if (i >= 0 && v > best) {
                    best = v;
                    bi = i;
                }
            }
            sc_stf(mpb, (unsigned)(c * 8), best);
            sc_stf(mpb, (unsigned)(c * 8 + 4), __int_as_float(bi));
        }
    }
    L0_STAMP(50);
    if ((G == 2 && K > 0) || f.do_max) ok = l0_barrier(a, b, sflag, ep) && ok;
    L0_STAMP(51);
    // ------------------------------------------------------------------ combine (block order: deterministic)
    if (G == 2 && K > 0) {
        const int n4 = a.epad / 4;
        const int per = (n4 + a.T - 1) / a.T;
        const float* xg = f.xpart + (long)b * a.T * a.epad;
        for (int i = tid; i < per; i += L0_NT) {
            const int e4 = rb * per + i;
            if (e4 < n4) {
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int t0 = 0; t0 < a.T; t0 += 16) {       // sixteen blocks' quads in flight, added in block order
                    f32x4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        v[u] = *reinterpret_cast<const f32x4*>(xg + (long)min(t0 + u, a.T - 1) * a.epad + e4 * 4);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (t0 + u < a.T) s += v[u];
                }
                if (!ok) s = (f32x4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e4 * 4 + j;
                    if (e < K * D) f.Xn[(long)b * K * D + e] = s[j];
                    else if (e < K * D + K * K) f.An[(long)b * K * K + e - K * D] = s[j];
                }
            }
        }
    }
    if (f.do_max && rb == 0) {
        const float* mg = f.mpart + (long)b * a.T * f.rw * 2;
        for (int c = tid; c < f.rw; c += L0_NT) {
            float best = -INFINITY;
            int bi = -1;
            for (int t0 = 0; t0 < a.T; t0 += 8) {
                float2 pr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    pr[u] = *reinterpret_cast<const float2*>(mg + ((long)min(t0 + u, a.T - 1) * f.rw + c) * 2);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = __float_as_int(pr[u].y);
                    if (t0 + u < a.T && i >= 0 && pr[u].x > best) {   // blocks in row order: strict > keeps the lowest row
                        best = pr[u].x;
                        bi = i;
                    }
                }
            }
            if (f.mask_readout && nb < N && !(best > 0.f)) {   // a masked (zero) row wins unless a valid row ties it at 0
                if (best < 0.f || bi < 0) {
                    best = 0.f;
                    bi = -1;
                }
            }
            if (!ok) best = __builtin_nanf("");
            f.feat[(long)b * f.ldfeat + f.featoff + c] = best;
            f.argmax[(long)b * f.rw + c] = bi;
        }
    }
    // ------------------------------------------------------------------ the last workgroup to finish cleans up
    lds_barrier();
    L0_STAMP(52);
    if (tid == 0) {
        const int old = ag_add(f.bar + BAR_DONE, 1);
        sflag[3] = old == (int)gridDim.x - 1;
    }
    lds_barrier();
    if (sflag[3]) {
        const int gflag = ag_ld(f.bar + BAR_GFLAG);
        if (tid < 64) f.pk_flag[tid] = tid == 0 ? gflag : 0;
        for (int g = tid; g < f.B; g += L0_NT) {
            ag_st(f.bar + BAR_GRAPH0 + g * BAR_GSTRIDE, 0);
            ag_st(f.bar + BAR_GRAPH0 + g * BAR_GSTRIDE + 32, 0);
        }
        if (tid == 0) {
            ag_st(f.bar + BAR_GCOUNT, 0);
            ag_st(f.bar + BAR_GFLAG, 0);
            ag_st(f.bar + BAR_DONE, 0);
            ag_st(f.bar + BAR_SEQ, (int)((seq + 1u) & 0x03ffffffu));
        }
    }
}

// =========================================================================================================
// Level-0 BACKWARD as one persistent launch (the mirror of k_level0_fwd; same decomposition, same barriers).
//   X' = S^T Z, A' = Tt^T S   ->  dZ += S dX',  dS = Z dX'^T + Tt dA' + A (S dA'^T)      encoders.py:1278-1279
//   S = softmax(Za Wp^T + bp) * mask  ->  dlogits, dWp, dbp, dZa                          encoders.py:1273-1275
//   every GraphConv layer, last to first: BatchNorm / ReLU / l2-normalise backward -> dU, bias sums;
//   G = A^T dU;  dW = x_in^T G;  dx_in += G W^T (+ the BatchNorm-backward partials of the layer below)
// Parameter gradients: every block stores its partial (compact layout), the graph's blocks sum disjoint slices in block
// order and write ONE slab row per graph — no float atomics anywhere on this path.
struct L0BArgs {
    Level0Bwd f;
    int T, RB, ldp, K8, steps;
    long vs_off[DP_MAX_LAYERS + 1];          // pass 0: V = S dA'^T; pass 1 + (L-1-l): dU of layer l
    int lds_dz[2];                           // float offsets: dZ accumulators [RB][ldz[g]]
    int lds_scr, slots_floats, scr_floats;   // scratch: [reduce slots | extra]
    int P0;                                  // compact parameter-gradient floats per block (multiple of 4)
    int cw[2][DP_MAX_LAYERS], cb[2][DP_MAX_LAYERS], cwp, cbp;   // compact offsets (cb < 0: no bias)
    int nseg;
    int seg_c[4 * DP_MAX_LAYERS + 2], seg_len[4 * DP_MAX_LAYERS + 2];
    long seg_flat[4 * DP_MAX_LAYERS + 2];
    int* dev_err;
    int spin_limit, target_bias;
};

#ifdef DP_STAMP
__device__ unsigned long long g_l0b_stamps[3][64];
#define L0B_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (threadIdx.x == 0 && (wid == 0 || wid == (int)gridDim.x / 2 || wid == (int)gridDim.x - 1)) \
            g_l0b_stamps[wid == 0 ? 0 : (wid == (int)gridDim.x - 1 ? 2 : 1)][i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define L0B_STAMP(i) \
    do {             \
    } while (0)
#endif

// Several bulk copies global -> LDS with every load of every region in flight before the first LDS write (regions of
// up to QN * 4 * 512 floats; larger ones finish in a second round).
struct L0Copy {
    float* dst;
    const float* src;
    int count, zero_to;
};
template <int NJ, int QN>
__device__ __forceinline__ void l0_copy_many(const L0Copy (&jobs)[NJ]) {
    f32x4_u q[NJ][QN];
    float tail[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nq = jobs[j].count >> 2;
#pragma unroll
        for (int u = 0; u < QN; ++u)
            q[j][u] = *reinterpret_cast<const f32x4_u*>(jobs[j].src + 4 * min(u * L0_NT + (int)threadIdx.x, max(nq - 1, 0)));
        const int te = (nq << 2) + (int)threadIdx.x;
        tail[j] = te < jobs[j].count ? jobs[j].src[te] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nq = jobs[j].count >> 2;
#pragma unroll
        for (int u = 0; u < QN; ++u) {
            const int e4 = u * L0_NT + (int)threadIdx.x;
            if (e4 < nq) *reinterpret_cast<f32x4*>(jobs[j].dst + 4 * e4) = q[j][u];
        }
        const int te = (nq << 2) + (int)threadIdx.x;
        if (te < jobs[j].count) jobs[j].dst[te] = tail[j];
        for (int e4 = QN * L0_NT + (int)threadIdx.x; e4 < nq; e4 += L0_NT)          // (regions beyond the batch: rare)
            *reinterpret_cast<f32x4*>(jobs[j].dst + 4 * e4) = *reinterpret_cast<const f32x4_u*>(jobs[j].src + 4 * e4);
        for (int e = jobs[j].count + (int)threadIdx.x; e < jobs[j].zero_to; e += L0_NT) jobs[j].dst[e] = 0.f;
    }
}

// `len` floats of an LDS staging tile -> this block's compact gradient vector (write-through: the graph's blocks read it)
__device__ __forceinline__ void l0_put_compact(ScBuf gp, int cstart, const float* stg, int len) {
    if ((reinterpret_cast<uintptr_t>(stg) & 15) != 0) {        // (a slice of a tile: element by element)
        for (int e = threadIdx.x; e < len; e += L0_NT) sc_stf(gp, (unsigned)((cstart + e) * 4), stg[e]);
        return;
    }
    const int nq = len >> 2;
    for (int e4 = threadIdx.x; e4 < nq; e4 += L0_NT)
        sc_st16(gp, (unsigned)((cstart + e4 * 4) * 4), *reinterpret_cast<const u32x4*>(stg + e4 * 4));
    const int te = (nq << 2) + (int)threadIdx.x;
    if (te < len) sc_stf(gp, (unsigned)((cstart + te) * 4), stg[te]);
}

template <int MI>
__global__ __launch_bounds__(L0_NT) void k_level0_bwd(L0BArgs a) {
    constexpr int RB = MI * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Level0Bwd& f = a.f;
    const int tid = threadIdx.x;
    const int tl = tid & 15, team = tid >> 4;
    const int N = f.N, G = f.G, L = f.L;
    int wid;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rm = nwg & 7;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int b = wid / a.T, rb = wid - b * a.T;
    const int r0 = rb * RB;
    const int nrows = min(RB, N - r0);
    const int rot = (rb * 4) % a.steps;
    L0B_STAMP(0);

    unsigned short* Alds = reinterpret_cast<unsigned short*>(lds);
    float* RA = lds;                                       // the adjacency block's bytes as scratch (before / between passes)
    float* DZ0 = lds + a.lds_dz[0];                        // [RB][D]   gradient of my rows of Ze (running)
    float* DZ1 = lds + a.lds_dz[1];                        // [RB][Da]  ... of Za
    float* SCR = lds + a.lds_scr;                          // [reduce slots | extra]
    float* EXT = SCR + a.slots_floats;
    int* sflag = reinterpret_cast<int*>(SCR + a.scr_floats);
    if (tid < 4) sflag[tid] = 0;
    L0Epoch ep{0};
    const unsigned seq = (unsigned)f.bar[BAR_SEQ];          // launch sequence number (stable until the last workgroup finishes)
    const bool exact = f.pk_flag[0] == 0;
    const int D = f.ldz[0], Da = f.ldz[1], K = f.K;
    const float* Arows = f.A + ((long)b * N + r0) * N;
    const float* Acols = f.A + (long)b * N * N + r0;
    const ScBuf gp = sc_buf(f.gpart + ((long)b * a.T + rb) * a.P0, (size_t)a.P0 * 4);
    const int k8_0 = r0 / 8;
    const int nk8 = min(rb == a.T - 1 ? a.K8 - k8_0 : RB / 8, a.K8 - k8_0);
    auto vs_wr = [&](int pass, int CTt) {
        return sc_buf(f.vs + a.vs_off[pass] + (long)b * 3 * CTt * a.K8 * 128, (size_t)3 * CTt * a.K8 * 128 * 2);
    };
    auto vs_rd = [&](int pass, int CTt) { return f.vs + a.vs_off[pass] + (long)b * 3 * CTt * a.K8 * 128; };
    bool ok = true;
    lds_barrier();

    // my rows of dZe (the max-readout scatter of the head's backward) start the running gradient
    if (G == 2 && K > 0) {
        // ------------------------------------------------------------------ pooling products (row-local)
        const int CTk = (K + 15) / 16, ctpk = K | 1;
        float* SL = RA;                                    // [RB][K]
        float* ZL = SL + RB * K;                           // [RB][D]
        float* TL = ZL + RB * D;                           // [RB][K]
        float* DAN = TL + RB * K;                          // [K][K]
        float* DXN = SCR;                                  // [K][D]     (reduce-slot area: free until the A V pass)
        float* VL = DXN + ((K * D + 3) & ~3);              // [RB][K]    V = S dA'^T
        float* DS = EXT;                                   // [RB][K]    Z dX'^T
        float* DS2 = DS + RB * K;                          // [RB][K]    Tt dA'
        {
            const long rowo = (long)b * N + r0;
            const L0Copy jobs[6] = {{SL, f.S + rowo * K, nrows * K, RB * K},
                                    {ZL, f.Z[0] + rowo * D, nrows * D, RB * D},
                                    {TL, f.Tt + rowo * K, nrows * K, RB * K},
                                    {DAN, f.dAn + (long)b * K * K, K * K, 0},
                                    {DXN, f.dXn + (long)b * K * D, K * D, 0},
                                    {DZ0, f.dZe + rowo * D, nrows * D, RB * D}};
            l0_copy_many<6, 2>(jobs);
        }
        lds_barrier();
        L0B_STAMP(1);
        l0_mma<false, false>(SL, K, DXN, D, RB, D, K, [&](int r, int j, float v) { DZ0[r * D + j] += v; });
        l0_mma<false, true>(ZL, D, DXN, D, RB, K, D, [&](int r, int i, float v) { DS[r * K + i] = v; }, 2);
        l0_mma<false, true>(SL, K, DAN, K, RB, K, K, [&](int r, int i, float v) { VL[r * K + i] = v; }, 4);
        l0_mma<false, false>(TL, K, DAN, K, RB, K, K, [&](int r, int i, float v) { DS2[r * K + i] = v; }, 6);
        lds_barrier();
        L0B_STAMP(2);
        l0_write_split(vs_wr(0, CTk), VL, K, CTk, a.K8, k8_0, nk8, nrows);
        {
            // my rows of the packed A: asked for in front of the barrier, written over the staged rows behind it
            L0RowStage<MI> q;
            l0_stage_issue<MI>(q, f.pkA + ((long)b * N + r0) * f.pk_ld, f.pk_ld, nrows, 0);
            ok = l0_barrier(a, b, sflag, ep) && ok;
            l0_stage_commit<MI>(q, Alds, a.ldp, f.pk_ld, nrows, 0);
            const int segs = (a.steps * 32 + 511) / 512;
            for (int seg = 1; seg < segs; ++seg) {
                l0_stage_issue<MI>(q, f.pkA + ((long)b * N + r0) * f.pk_ld, f.pk_ld, nrows, seg);
                l0_stage_commit<MI>(q, Alds, a.ldp, f.pk_ld, nrows, seg);
            }
        }
        lds_barrier();
        L0B_STAMP(3);
        // ------------------------------------------------------------------ dS += A V;  softmax backward
        l0_aggregate<MI>(a, Alds, exact, Arows, N, 1, nrows, vs_rd(0, CTk), CTk, SCR, ctpk, rot);
        lds_barrier();
        L0B_STAMP(4);
        float* DLOG = DS;                                  // in place, row by row
        // S (and the link loss's d_assign) of all my rows first: one memory round trip, not one per row round
        constexpr int RROUNDS = (RB + L0_TEAMS - 1) / L0_TEAMS;
        float svq[RROUNDS][L0_NK], dvq[RROUNDS][L0_NK];
#pragma unroll
        for (int q = 0; q < RROUNDS; ++q) {
            const long row = (long)b * N + min(r0 + min(team + q * L0_TEAMS, RB - 1), N - 1);
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                const int c = min(tl + 16 * k, K - 1);
                svq[q][k] = f.S[row * K + c];
                dvq[q][k] = f.d_assign ? f.d_assign[row * K + c] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < RROUNDS; ++q) {
            const int r = team + q * L0_TEAMS;
            if (r >= RB) break;
            float sv[L0_NK], dv[L0_NK];
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                sv[k] = svq[q][k];
                dv[k] = dvq[q][k];
            }
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                const int c = min(tl + 16 * k, K - 1);
                const int o = r * ctpk + c;
                const float agg = (SCR[o] + SCR[RB * ctpk + o]) + (SCR[2 * RB * ctpk + o] + SCR[3 * RB * ctpk + o]);
                dv[k] += (agg + DS[r * K + c]) + DS2[r * K + c];
                dot += (tl + 16 * k < K) ? sv[k] * dv[k] : 0.f;
            }
            dot = row16_sum(dot);
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                const int c = tl + 16 * k;
                if (c < K) DLOG[r * K + c] = r < nrows ? sv[k] * (dv[k] - dot) : 0.f;
            }
        }
        lds_barrier();
        L0B_STAMP(5);
        // assign_pred: dWp = dlog^T Za, dbp = column sums, dZa = dlog Wp
        {
            float* ZAL = SCR;                              // [RB][Da]  (the reduce slots are free again)
            float* WP = ZAL + ((RB * Da + 3) & ~3);        // [K][Da]
            float* STG = RA;                               // staging of the parameter gradients (the A block is done with)
            const L0Copy jobs[2] = {{ZAL, f.Z[1] + ((long)b * N + r0) * Da, nrows * Da, RB * Da},
                                    {WP, f.params + f.wp_off, K * Da, 0}};
            l0_copy_many<2, 3>(jobs);
            lds_barrier();
            l0_mma<true, false>(DLOG, K, ZAL, Da, K, Da, RB, [&](int i, int j, float v) { STG[i * Da + j] = v; });
            l0_mma<false, false>(DLOG, K, WP, Da, RB, Da, K, [&](int r, int j, float v) { DZ1[r * Da + j] = v; }, 4);
            float* SB = STG + ((K * Da + 3) & ~3);
            // dbp: column sums of dlog over my rows — one 16-lane team per column, RB / 16 rows per lane, fixed tree order
            for (int c = team; c < K; c += L0_TEAMS) {
                float t = 0.f;
#pragma unroll
                for (int rr = 0; rr < RB / 16; ++rr) t += DLOG[(tl * (RB / 16) + rr) * K + c];
                t = row16_sum(t);
                if (tl == 0) SB[c] = t;
            }
            lds_barrier();
            l0_put_compact(gp, a.cwp, STG, K * Da);
            if (a.cbp >= 0) l0_put_compact(gp, a.cbp, SB, K);
        }
        L0B_STAMP(6);
    } else {
        const L0Copy jobs[1] = {{DZ0, f.dZe + ((long)b * N + r0) * D, nrows * D, RB * D}};
        l0_copy_many<1, 4>(jobs);
        if (G == 2)
            for (int e = tid; e < RB * Da; e += L0_NT) DZ1[e] = 0.f;
    }
    lds_barrier();

    // ------------------------------------------------------------------ the GraphConv layers, last to first
    constexpr int ITEMS = (RB * 2 + L0_TEAMS - 1) / L0_TEAMS;
    bool have_at = false;
    for (int l = L - 1; l >= 0; --l) {
        const bool last = l == L - 1;
        const bool has_bn = !last && f.bn;
        const int w0 = f.st[0].dims[l + 1], w1 = G == 2 ? f.st[1].dims[l + 1] : 0;
        const int d0 = f.st[0].dims[l], d1 = G == 2 ? f.st[1].dims[l] : 0;
        const int ct = w0 + w1, wmax = max(w0, w1);
        const int CTt = (ct + 15) / 16, ctp = ct | 1;
        const int pass = 1 + (L - 1 - l);
        float* DU = SCR;                                   // [RB][ct]  (reduce-slot area: free until the pass)
        // The saved forward values of ALL my row items (y, xhat, 1 / norm, rstd) are asked for first, in one go: taken item
        // by item inside the loop below, each item paid its own memory round trip (3 k cycles per item and layer)
        float rw_y[ITEMS][L0_NK], rw_x[ITEMS][L0_NK], rw_inv[ITEMS], rw_rstd[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = team + j * L0_TEAMS;
            const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
            const int node = min(r0 + r, N - 1);
            const long row = (long)b * N + node;
            const int wg = g ? w1 : w0, c0gg = g ? w0 : 0;
            const float* yp = last ? f.Z[g] + row * f.ldz[g] + f.coff[g][l] : f.Y[l] + row * ct + c0gg;
            const float* xp = f.Z[g] + row * f.ldz[g] + f.coff[g][l];
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                rw_y[j][k] = 0.f;
                rw_x[j][k] = 0.f;
                if (16 * k < wmax) {
                    const int c = min(tl + 16 * k, wg - 1);
                    rw_y[j][k] = yp[c];
                    if (has_bn) rw_x[j][k] = xp[c];
                }
            }
            rw_inv[j] = f.invn[l][row * G + g];
            rw_rstd[j] = has_bn ? f.stats[l][((long)node * G + g) * 2 + 1] : 1.f;
        }
        // ---- BatchNorm-backward means of my node indices: every graph's (sum dx, sum dx xhat) partials -> (m0, m1) in LDS
        float* M01 = EXT;                                  // [RB * G][2]
        if (has_bn) {
            const ScBuf partr = sc_buf(f.part + (long)l * f.B * N * G * 4, (size_t)f.B * N * G * 4 * sizeof(float));
            float p0[ITEMS][L0_BPAIRS], p1[ITEMS][L0_BPAIRS];
            {
                unsigned off[ITEMS];
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int it = team + j * L0_TEAMS;
                    const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
                    off[j] = (unsigned)((((long)min(r0 + r, N - 1) * G + g) * f.B) * 16);
                }
                ok = l0_poll_entries<ITEMS>(a, partr, off, f.B, (seq << 4) | (unsigned)(l + 1), p0, p1, sflag) && ok;
            }
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int it = team + j * L0_TEAMS;
                const int g = G == 2 ? (it & 1) : 0;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int u = 0; u < L0_BPAIRS; ++u) {
                    s0 += (tl + 16 * u < f.B) ? p0[j][u] : 0.f;
                    s1 += (tl + 16 * u < f.B) ? p1[j][u] : 0.f;
                }
                const float cnt = (float)f.B * (float)(g ? w1 : w0);
                s0 = row16_sum(s0) / cnt;
                s1 = row16_sum(s1) / cnt;
                if (!ok) s0 = __builtin_nanf("");
                if (tl == 0 && it < RB * G) {
                    M01[it * 2] = s0;
                    M01[it * 2 + 1] = s1;
                }
            }
        }
        L0B_STAMP(44 + 2 * (L - 1 - l));
        // ---- BatchNorm / ReLU / l2-normalise backward of my rows -> dU (each team reads back only its own means)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = team + j * L0_TEAMS;
            const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
            const int node = min(r0 + r, N - 1);
            const long row = (long)b * N + node;
            const int wg = g ? w1 : w0, c0gg = g ? w0 : 0;
            const float (&yv)[L0_NK] = rw_y[j];
            const float (&xh)[L0_NK] = rw_x[j];
            const float inv = rw_inv[j];
            const float rstd = rw_rstd[j];
            const float m0 = has_bn ? M01[min(it, RB * G - 1) * 2] : 0.f;
            const float m1 = has_bn ? M01[min(it, RB * G - 1) * 2 + 1] : 0.f;
            const float* dzr = (g ? DZ1 : DZ0) + r * f.ldz[g] + f.coff[g][l];
            const bool project = inv < 1.0f / L0_L2_EPS;
            float dv[L0_NK];
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                dv[k] = 0.f;
                if (16 * k < wmax) {
                    float d = dzr[min(tl + 16 * k, wg - 1)];
                    if (has_bn) d = rstd * (d - m0 - xh[k] * m1);
                    if (!last) d = yv[k] > 0.f ? d : 0.f;
                    if (tl + 16 * k >= wg) d = 0.f;
                    dv[k] = d;
                    dot += d * yv[k];
                }
            }
            dot = row16_sum(dot);
            if (j == 0) L0B_STAMP(45 + 2 * (L - 1 - l));
#pragma unroll
            for (int k = 0; k < L0_NK; ++k) {
                const int c = tl + 16 * k;
                if (16 * k < wmax && c < wg && it < RB * G) {
                    const float v = project ? inv * (dv[k] - yv[k] * dot) : inv * dv[k];
                    DU[r * ct + c0gg + c] = r < nrows ? v : 0.f;
                }
            }
        }
        lds_barrier();
        L0B_STAMP(8 + 8 * (L - 1 - l));
        // bias gradients: column sums of dU over my rows
        {
            // eight row groups in parallel, added in row order
            float* SB = EXT + RB * 4;
            float* SB8 = SB + ((ct + 3) & ~3);             // [8][ct]
            constexpr int RG = RB / 8;
            for (int e = tid; e < 8 * ct; e += L0_NT) {
                const int grp = e / ct, c = e - grp * ct;
                float t = 0.f;
#pragma unroll
                for (int rr = 0; rr < RG; ++rr) t += DU[(grp * RG + rr) * ct + c];
                SB8[e] = t;
            }
            lds_barrier();
            for (int c = tid; c < ct; c += L0_NT) {
                float t = 0.f;
#pragma unroll
                for (int grp = 0; grp < 8; ++grp) t += SB8[grp * ct + c];
                SB[c] = t;
            }
        }
        l0_write_split(vs_wr(pass, CTt), DU, ct, CTt, a.K8, k8_0, nk8, nrows);
        lds_barrier();
        if (a.cb[0][l] >= 0) l0_put_compact(gp, a.cb[0][l], EXT + RB * 4, w0);
        if (G == 2 && a.cb[1][l] >= 0) l0_put_compact(gp, a.cb[1][l], EXT + RB * 4 + w0, w1);
        // layer input rows (the left operand of dW, and xhat of the layer below) + this layer's weights: asked for now,
        // they land under the barrier
        // (layer 0: the second stack's input rows do not fit beside the first's — they go to the adjacency block's bytes
        // once the last pass is over)
        float* XIN0 = EXT + RB * 4 + 9 * ((ct + 3) & ~3);
        float* XIN1 = l > 0 ? XIN0 + RB * d0 : RA;
        float* W0 = XIN1 + RB * d1;                        // (l > 0 only)
        float* W1 = W0 + ((d0 * w0 + 3) & ~3);
        {
            const long rowo = (long)b * N + r0;
            if (l > 0) {
                // the previous layer's slices of the concat buffers are column slices: row by row
                {
                    float xv[ITEMS][L0_NK];
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) {
                        const int it = team + j * L0_TEAMS;
                        const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
                        const int dg = g ? d1 : d0;
                        const float* src = f.Z[g] + (rowo + min(r, nrows - 1)) * f.ldz[g] + f.coff[g][l - 1];
#pragma unroll
                        for (int k = 0; k < L0_NK; ++k) xv[j][k] = src[min(tl + 16 * k, dg - 1)];
                    }
                    // the weights are asked for behind the rows and land with them (one round trip for both)
                    const L0Copy jw[2] = {{W0, f.params + f.st[0].w_off[l], d0 * w0, 0},
                                          {W1, G == 2 ? f.params + f.st[1].w_off[l] : f.params, G == 2 ? d1 * w1 : 0, 0}};
                    l0_copy_many<2, 2>(jw);
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) {
                        const int it = team + j * L0_TEAMS;
                        const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
                        const int dg = g ? d1 : d0;
                        float* dst = (g ? XIN1 : XIN0) + r * dg;
#pragma unroll
                        for (int k = 0; k < L0_NK; ++k)
                            if (tl + 16 * k < dg && it < RB * G) dst[tl + 16 * k] = r < nrows ? xv[j][k] : 0.f;
                    }
                }
            } else {
                const L0Copy jx[1] = {{XIN0, f.x0[0] + rowo * d0, nrows * d0, RB * d0}};
                l0_copy_many<1, 4>(jx);
            }
        }
        L0B_STAMP(9 + 8 * (L - 1 - l));
        if (!have_at) {
            L0RowStage<MI> q;
            l0_stage_issue<MI>(q, f.pkAt + ((long)b * N + r0) * f.pk_ld, f.pk_ld, nrows, 0);
            ok = l0_barrier(a, b, sflag, ep) && ok;
            l0_stage_commit<MI>(q, Alds, a.ldp, f.pk_ld, nrows, 0);
            const int segs = (a.steps * 32 + 511) / 512;
            for (int seg = 1; seg < segs; ++seg) {
                l0_stage_issue<MI>(q, f.pkAt + ((long)b * N + r0) * f.pk_ld, f.pk_ld, nrows, seg);
                l0_stage_commit<MI>(q, Alds, a.ldp, f.pk_ld, nrows, seg);
            }
            have_at = true;
            lds_barrier();
        } else {
            ok = l0_barrier(a, b, sflag, ep) && ok;
        }
        L0B_STAMP(10 + 8 * (L - 1 - l));
        // ---- G = A^T dU (my rows), summed in place into the first reduce slot
        l0_aggregate<MI>(a, Alds, exact, Acols, 1, N, nrows, vs_rd(pass, CTt), CTt, SCR, ctp, rot);
        lds_barrier();
        L0B_STAMP(11 + 8 * (L - 1 - l));
        float* Gt = SCR;                                   // [RB][ctp]
        for (int e = tid; e < RB * ct; e += L0_NT) {
            const int r = e / ct, c = e - r * ct;
            const int o = r * ctp + c;
            const float v = (SCR[o] + SCR[RB * ctp + o]) + (SCR[2 * RB * ctp + o] + SCR[3 * RB * ctp + o]);
            Gt[o] = r < nrows ? v : 0.f;
        }
        lds_barrier();
        L0B_STAMP(12 + 8 * (L - 1 - l));
        if (l == 0 && G == 2) {
            const L0Copy jx[1] = {{XIN1, f.x0[1] + ((long)b * N + r0) * d1, nrows * d1, RB * d1}};
            l0_copy_many<1, 4>(jx);
            lds_barrier();
        }
        // ---- dW = x_in^T G  (staged behind the first slot);  dx_in += G W^T
        float* STG0 = SCR + RB * ctp;
        float* STG1 = STG0 + ((d0 * w0 + 3) & ~3);
        l0_mma<true, false>(XIN0, d0, Gt, ctp, d0, w0, RB, [&](int i, int j, float v) { STG0[i * w0 + j] = v; });
        if (G == 2)
            l0_mma<true, false>(XIN1, d1, Gt + w0, ctp, d1, w1, RB, [&](int i, int j, float v) { STG1[i * w1 + j] = v; }, 3);
        if (l > 0) {
            float* dz0 = DZ0 + f.coff[0][l - 1];
            l0_mma<false, true>(Gt, ctp, W0, w0, RB, d0, w0, [&](int r, int k, float v) { dz0[r * D + k] += v; }, 5);
            if (G == 2) {
                float* dz1 = DZ1 + f.coff[1][l - 1];
                l0_mma<false, true>(Gt + w0, ctp, W1, w1, RB, d1, w1, [&](int r, int k, float v) { dz1[r * Da + k] += v; }, 7);
            }
        }
        lds_barrier();
        L0B_STAMP(13 + 8 * (L - 1 - l));
        l0_put_compact(gp, a.cw[0][l], STG0, d0 * w0);
        if (G == 2) l0_put_compact(gp, a.cw[1][l], STG1, d1 * w1);
        if (l > 0 && f.bn) {
            // BatchNorm-backward partials of layer l - 1: (sum dx, sum dx * xhat) per (row, group); xhat = the layer input
            const ScBuf partw = sc_buf(f.part + (long)(l - 1) * f.B * N * G * 4, (size_t)f.B * N * G * 4 * sizeof(float));
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int it = team + j * L0_TEAMS;
                const int r = min(G == 2 ? it >> 1 : it, RB - 1), g = G == 2 ? (it & 1) : 0;
                const bool on = it < RB * G && r < nrows;
                const int dg = g ? d1 : d0;
                const float* dzr = (g ? DZ1 : DZ0) + r * f.ldz[g] + f.coff[g][l - 1];
                const float* xr = (g ? XIN1 : XIN0) + r * dg;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < L0_NK; ++k) {
                    const int c = min(tl + 16 * k, dg - 1);
                    const float d = (tl + 16 * k < dg) ? dzr[c] : 0.f;
                    s0 += d;
                    s1 += d * xr[c];
                }
                s0 = row16_sum(s0);
                s1 = row16_sum(s1);
                if (on && tl == 0) {
                    const long po = (((long)min(r0 + r, N - 1) * G + g) * f.B + b) * 16;
                    sc_st16(partw, (unsigned)po, sc_tagged(s0, s1, (seq << 4) | (unsigned)l));      // tag of layer l - 1
                }
            }
            L0B_STAMP(14 + 8 * (L - 1 - l));
        }
        lds_barrier();
        L0B_STAMP(15 + 8 * (L - 1 - l));
    }
    // ------------------------------------------------------------------ the graph's parameter gradients: one slab row
    ok = l0_barrier(a, b, sflag, ep) && ok;
    L0B_STAMP(40);
    {
        const int n4 = a.P0 / 4;
        const int per = (n4 + a.T - 1) / a.T;
        const float* gg = f.gpart + (long)b * a.T * a.P0;
        float* slab = f.slabs + (long)b * f.slab_gstride;
        for (int i = tid; i < per; i += L0_NT) {
            const int e4 = rb * per + i;
            if (e4 < n4) {
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int t0 = 0; t0 < a.T; t0 += 16) {
                    f32x4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        v[u] = *reinterpret_cast<const f32x4*>(gg + (long)min(t0 + u, a.T - 1) * a.P0 + e4 * 4);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (t0 + u < a.T) s += v[u];
                }
                if (!ok) s = (f32x4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                int sg = 0;
                while (sg + 1 < a.nseg && a.seg_c[sg + 1] <= e4 * 4) ++sg;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e4 * 4 + j - a.seg_c[sg];        // (segments start on quads: a quad never straddles two)
                    if (e < a.seg_len[sg]) slab[a.seg_flat[sg] + e] = s[j];
                }
            }
        }
    }
    lds_barrier();
    L0B_STAMP(41);
    if (tid == 0) {
        const int old = ag_add(f.bar + BAR_DONE, 1);
        sflag[3] = old == (int)gridDim.x - 1;
    }
    lds_barrier();
    if (sflag[3]) {
        for (int g = tid; g < f.B; g += L0_NT) ag_st(f.bar + BAR_GRAPH0 + g * BAR_GSTRIDE, 0);
        if (tid == 0) {
            ag_st(f.bar + BAR_GCOUNT, 0);
            ag_st(f.bar + BAR_DONE, 0);
            ag_st(f.bar + BAR_SEQ, (int)((seq + 1u) & 0x03ffffffu));
        }
    }
}

struct L0Geom {
    int RB, T, ldp, K8, steps, epad;
    int lds_act[2], lds_scr, scr_floats;
    size_t lds_bytes;
};
int l0_device_cus() {
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
// Rows per block, LDS layout.  Returns false when no block size fits (LDS, or more workgroups than CUs).
bool l0_geometry(const Level0Fwd& f, L0Geom& g) {
    const int cus = l0_device_cus();
    if (cus <= 0) return false;
    const int N = f.N, G = f.G, L = f.L;
    g.steps = (N + 31) / 32;
    g.K8 = g.steps * 4;
    g.ldp = ((N + 511) / 512) * 512 + 8;
    const int K = G == 2 ? f.K : 0, D = f.ldz[0];
    g.epad = (K * D + K * K + 3) & ~3;
    for (int RB = 16; RB <= 64; RB += 16) {
        const int T = (N + RB - 1) / RB;
        if ((long)f.B * T > cus) continue;
        size_t scr = 0;
        auto need = [&](size_t fl) { scr = fl > scr ? fl : scr; };
        // phase 0: X rows, W, PT
        {
            size_t fl = 0;
            for (int s = 0; s < G; ++s)        // (both stacks' inputs are staged even when they are the same tensor: the
                fl += (size_t)RB * f.st[s].dims[0] +                                          // layout must not depend on pointers)
                      (((size_t)f.st[s].dims[0] * f.st[s].dims[1] + 3) & ~size_t(3));
            int ct = 0;
            for (int s = 0; s < G; ++s) ct += f.st[s].dims[1];
            need(fl + (size_t)RB * ct);
        }
        for (int l = 0; l < L; ++l) {
            int ct = 0;
            for (int s = 0; s < G; ++s) ct += f.st[s].dims[l + 1];
            const size_t slots = (size_t)4 * RB * (((ct + 15) / 16) * 16 + 1);   // reduce slots
            need(slots);
            if (l < L - 1) {
                size_t fl = 0;
                int ctn = 0;
                for (int s = 0; s < G; ++s) {
                    fl += ((size_t)f.st[s].dims[l + 1] * f.st[s].dims[l + 2] + 3) & ~size_t(3);
                    ctn += f.st[s].dims[l + 2];
                }
                need(slots + fl + (size_t)RB * ctn);                       // ... | next layer's weights | P_{l+1}
            }
        }
        need((size_t)g.epad + 16 * (size_t)(f.rw > 0 ? f.rw : 0));             // partial products | max-readout merge
        if (K > 0) {
            const int ctpk = ((K + 15) / 16) * 16 + 1;
            need((size_t)4 * RB * ctpk + 2 * (size_t)RB * K);                 // reduce slots | TL | SL
            // the staged assign_pred weight (before the pass) and the partial X' | A' (after it) overlay the slots
            if ((size_t)K * f.ldz[1] > (size_t)4 * RB * ctpk || (size_t)g.epad > (size_t)4 * RB * ctpk) continue;
        }
        const size_t adj_fl = ((size_t)RB * g.ldp * 2 + 15) / 16 * 4;         // floats, 16-byte multiple
        g.lds_act[0] = (int)adj_fl;
        g.lds_act[1] = g.lds_act[0] + ((RB * f.ldz[0] + 3) & ~3);
        g.lds_scr = g.lds_act[1] + (G == 2 ? ((RB * f.ldz[1] + 3) & ~3) : 0);
        g.scr_floats = (int)((scr + 3) & ~size_t(3));
        g.lds_bytes = ((size_t)g.lds_scr + g.scr_floats + (2 * DP_MAX_LAYERS + 1) * 64 + 16) * sizeof(float);
        if (g.lds_bytes > 159 * 1024) continue;
        g.RB = RB;
        g.T = T;
        return true;
    }
    return false;
}

// Launch timing for bench.py's roofline object: when switched on (dp_profile_level0), every launch of the two persistent
// kernels is bracketed by a pair of HIP events on the launch stream.  Eager launches only — a capturing stream is left alone.
struct L0Prof {
    std::mutex m;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];
};
L0Prof& l0_prof() {
    static L0Prof p;
    return p;
}
struct L0Timed {
    hipStream_t st;
    int which;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    L0Timed(hipStream_t s, int w) : st(s), which(w) {
        L0Prof& p = l0_prof();
        if (!p.on) return;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            e0 = e1 = nullptr;
            return;
        }
        (void)hipEventRecord(e0, st);
    }
    ~L0Timed() {
        if (!e0) return;
        (void)hipEventRecord(e1, st);
        L0Prof& p = l0_prof();
        std::lock_guard<std::mutex> g(p.m);
        p.ev[which].emplace_back(e0, e1);
    }
};

template <int MI>
void l0_launch(Seq& q, const L0Args& a, size_t lds_bytes) {
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_level0_fwd<MI>), 160 * 1024, "k_level0_fwd");
    if (!q.ok()) return;
    L0Timed timed(q.stream, 0);
    hipLaunchKernelGGL(k_level0_fwd<MI>, dim3(a.f.B * a.T), dim3(L0_NT), lds_bytes, q.stream, a);
    q.check_launch("level0_forward");
}

}  // namespace

size_t level0_bar_ints(int B) { return (size_t)BAR_GRAPH0 + (size_t)BAR_GSTRIDE * B; }
// write-once exchange regions: one split operand per aggregation pass, one partial block per BatchNorm layer
static void l0_vs_layout(const Level0Fwd& f, long (&off)[DP_MAX_LAYERS + 1], size_t& total) {
    const int K8 = ((f.N + 31) / 32) * 4;
    size_t o = 0;
    for (int p = 0; p <= f.L; ++p) {
        int ct = 0;
        if (p < f.L) for (int s = 0; s < f.G; ++s) ct += f.st[s].dims[p + 1];
        else ct = f.G == 2 ? f.K : 0;
        off[p] = (long)o;
        o += (size_t)f.B * 3 * ((ct + 15) / 16) * K8 * 128;
    }
    total = o;
}
size_t level0_vs_elems(const Level0Fwd& f) {
    long off[DP_MAX_LAYERS + 1];
    size_t total;
    l0_vs_layout(f, off, total);
    return total + 64;
}
size_t level0_part_floats(const Level0Fwd& f) { return (size_t)(f.L > 1 ? f.L - 1 : 1) * f.B * f.N * f.G * 4 + 4; }
const int* level0_error_word(const int* bar) { return bar + BAR_ERR; }
int* level0_seq_word(int* bar) { return bar + BAR_SEQ; }
size_t level0_xpart_floats(const Level0Fwd& f) {
    L0Geom g;
    if (!l0_geometry(f, g)) return 4;
    return (size_t)f.B * g.T * g.epad + 4;
}
size_t level0_mpart_floats(const Level0Fwd& f) {
    L0Geom g;
    if (!l0_geometry(f, g)) return 4;
    return (size_t)f.B * g.T * (f.rw > 0 ? f.rw : 1) * 2 + 4;
}

// Shapes the persistent kernel takes (everything else runs as the launch sequence of rounds 1-2).
bool level0_persistent_ok(const Level0Fwd& f) {
    if (knobs().no_l0_persist) return false;
    if (f.G < 1 || f.G > 2 || f.L < 1 || f.L > DP_MAX_LAYERS) return false;
    if (knobs().no_pack) return false;                       // (the fp32-everywhere ablation)
    if (f.N < 64 || (f.N & 3) || f.B < 1 || f.B > 16 * L0_BPAIRS) return false;
    for (int l = 0; l < f.L; ++l) {
        int ct = 0;
        size_t wfl = 0;
        for (int s = 0; s < f.G; ++s) {
            if (f.st[s].dims[l + 1] < 1 || f.st[s].dims[l + 1] > 16 * L0_NK) return false;
            ct += f.st[s].dims[l + 1];
            wfl += (size_t)f.st[s].dims[l] * f.st[s].dims[l + 1];
        }
        if (ct > 96 || wfl > 8192) return false;              // <= 6 column tiles; a layer's weights <= 32 KiB
        for (int s = 0; s < f.G; ++s)                         // one stack's layer weights: <= 4 quads per thread in flight
            if ((size_t)f.st[s].dims[l] * f.st[s].dims[l + 1] > 4 * 4 * L0_NT) return false;
    }
    for (int s = 0; s < f.G; ++s)
        if (f.st[s].dims[0] < 1 || f.st[s].dims[0] > 256 || f.ldz[s] > 160) return false;      // (x rows: <= 64 x 256 floats)
    if (f.G == 2) {
        if (f.K < 1 || f.K > 16 * L0_NK || (size_t)f.K * f.ldz[1] > 8192) return false;
    }
    if (f.do_max && (f.rw < 1 || f.zoff + f.rw > f.ldz[0])) return false;
    L0Geom g;
    return l0_geometry(f, g);
}

void level0_forward(Seq& q, const Level0Fwd& f) {
    if (!q.ok()) return;
    L0Geom g;
    if (!l0_geometry(f, g)) {
        set_error("level0_forward: shape outside the persistent kernel's envelope");
        q.err = DP_ERR_UNSUPPORTED;
        return;
    }
    L0Args a{};
    a.f = f;
    a.T = g.T; a.RB = g.RB; a.ldp = g.ldp; a.K8 = g.K8; a.steps = g.steps; a.epad = g.epad;
    a.lds_act[0] = g.lds_act[0]; a.lds_act[1] = g.lds_act[1]; a.lds_scr = g.lds_scr; a.scr_floats = g.scr_floats;
    a.dev_err = device_error_word();
    a.spin_limit = knobs().test_barrier_fail ? 64 : L0_SPIN_LIMIT;
    a.target_bias = knobs().test_barrier_fail ? 1 : 0;
    {
        size_t total;
        l0_vs_layout(f, a.vs_off, total);
    }
    a.zero_n16 = f.zero_p ? (long)(f.zero_bytes / 16) : 0;
    if (f.zero_p && ((reinterpret_cast<uintptr_t>(f.zero_p) & 15) != 0 || (f.zero_bytes & 15) != 0)) {
        zero_fill(q, f.zero_p, f.zero_bytes);
        a.f.zero_p = nullptr;
        a.zero_n16 = 0;
    }
    switch (g.RB / 16) {
        case 1: l0_launch<1>(q, a, g.lds_bytes); break;
        case 2: l0_launch<2>(q, a, g.lds_bytes); break;
        case 3: l0_launch<3>(q, a, g.lds_bytes); break;
        default: l0_launch<4>(q, a, g.lds_bytes); break;
    }
}

// ------------------------------------------------------------------ backward: geometry, compact layout, launch
namespace {
struct L0BGeom {
    int RB, T, ldp, K8, steps;
    int lds_dz[2], lds_scr, slots_floats, scr_floats;
    size_t lds_bytes;
};
Level0Fwd l0_as_fwd(const Level0Bwd& f) {          // the fields the shared geometry / envelope helpers read
    Level0Fwd w{};
    w.B = f.B; w.N = f.N; w.L = f.L; w.G = f.G; w.bn = f.bn;
    w.st[0] = f.st[0]; w.st[1] = f.st[1];
    w.ldz[0] = f.ldz[0]; w.ldz[1] = f.ldz[1];
    w.K = f.K;
    return w;
}
bool l0b_geometry(const Level0Bwd& f, L0BGeom& g) {
    // the SAME block size as the forward pass (the two launches share nothing but the barrier block, but one
    // decomposition keeps the workspace sizing in one place)
    L0Geom fg;
    Level0Fwd w = l0_as_fwd(f);
    w.do_max = 0;
    if (!l0_geometry(w, fg)) return false;
    const int RB = fg.RB, N = f.N, G = f.G, L = f.L;
    const int K = G == 2 ? f.K : 0, D = f.ldz[0], Da = G == 2 ? f.ldz[1] : 0;
    g.RB = RB; g.T = fg.T; g.ldp = fg.ldp; g.K8 = fg.K8; g.steps = fg.steps;
    const size_t ra = ((size_t)RB * g.ldp * 2 + 15) / 16 * 4;               // floats
    size_t slots = 0, ext = 0;
    auto S_ = [&](size_t v) { slots = v > slots ? v : slots; };
    auto E_ = [&](size_t v) { ext = v > ext ? v : ext; };
    if (K > 0) {
        if ((size_t)2 * RB * K + (size_t)RB * D + (size_t)K * K > ra) return false;          // staged rows in the block's bytes
        if ((size_t)K * Da + K + 8 > ra) return false;                                        // dWp | dbp staging
        S_((size_t)4 * RB * (K | 1));
        S_((((size_t)K * D + 3) & ~size_t(3)) + (size_t)RB * K);                               // dX' | V
        S_((((size_t)RB * Da + 3) & ~size_t(3)) + (size_t)K * Da);                             // Za rows | Wp
        E_((size_t)2 * RB * K);
    }
    for (int l = 0; l < L; ++l) {
        int ct = 0, din = 0;
        size_t wfl = 0;
        for (int s2 = 0; s2 < G; ++s2) {
            ct += f.st[s2].dims[l + 1];
            din += f.st[s2].dims[l];
            wfl += ((size_t)f.st[s2].dims[l] * f.st[s2].dims[l + 1] + 3) & ~size_t(3);
        }
        const size_t ctp = (size_t)(ct | 1);
        S_((size_t)4 * RB * ctp);
        S_((size_t)RB * ctp + wfl);                                                            // G | staged dW
        const size_t ebase = (size_t)RB * 4 + 9 * (((size_t)ct + 3) & ~size_t(3));     // means | bias sums | their 8 row groups
        if (l > 0) E_(ebase + (size_t)RB * din + wfl);
        else {
            E_(ebase + (size_t)RB * f.st[0].dims[0]);
            if (G == 2 && (size_t)RB * f.st[1].dims[0] > ra) return false;
        }
    }
    g.lds_dz[0] = (int)ra;
    g.lds_dz[1] = g.lds_dz[0] + ((RB * D + 3) & ~3);
    g.lds_scr = g.lds_dz[1] + (G == 2 ? ((RB * Da + 3) & ~3) : 0);
    g.slots_floats = (int)((slots + 3) & ~size_t(3));
    g.scr_floats = g.slots_floats + (int)((ext + 3) & ~size_t(3));
    g.lds_bytes = ((size_t)g.lds_scr + g.scr_floats + 16) * sizeof(float);
    return g.lds_bytes <= 160 * 1024;
}
// compact parameter-gradient layout of one block: every tensor starts on a quad
void l0b_compact(const Level0Bwd& f, L0BArgs& a) {
    int o = 0, ns = 0;
    auto seg = [&](int& dst, long flat, int len) {
        dst = o;
        a.seg_c[ns] = o;
        a.seg_flat[ns] = flat;
        a.seg_len[ns] = len;
        ++ns;
        o += (len + 3) & ~3;
    };
    for (int l = 0; l < f.L; ++l)
        for (int g = 0; g < f.G; ++g) {
            seg(a.cw[g][l], f.st[g].w_off[l], f.st[g].dims[l] * f.st[g].dims[l + 1]);
            if (f.st[g].b_off[l] >= 0) seg(a.cb[g][l], f.st[g].b_off[l], f.st[g].dims[l + 1]);
            else a.cb[g][l] = -1;
        }
    a.cwp = a.cbp = -1;
    if (f.G == 2 && f.K > 0) {
        seg(a.cwp, f.wp_off, f.K * f.ldz[1]);
        if (f.bp_off >= 0) seg(a.cbp, f.bp_off, f.K);
    }
    a.nseg = ns;
    a.P0 = o;
}
void l0b_vs_layout(const Level0Bwd& f, long (&off)[DP_MAX_LAYERS + 1], size_t& total) {
    const int K8 = ((f.N + 31) / 32) * 4;
    size_t o = 0;
    for (int p = 0; p <= f.L; ++p) {
        int ct = 0;
        if (p == 0) ct = f.G == 2 ? f.K : 0;
        else for (int s2 = 0; s2 < f.G; ++s2) ct += f.st[s2].dims[f.L - p + 1];          // pass 1 + (L-1-l): layer l = L - p
        off[p] = (long)o;
        o += (size_t)f.B * 3 * ((ct + 15) / 16) * K8 * 128;
    }
    total = o;
}
template <int MI>
void l0b_launch(Seq& q, const L0BArgs& a, size_t lds_bytes) {
    static DynLdsOnce attr;
    ensure_dyn_lds(q, attr, reinterpret_cast<const void*>(&k_level0_bwd<MI>), 160 * 1024, "k_level0_bwd");
    if (!q.ok()) return;
    L0Timed timed(q.stream, 1);
    hipLaunchKernelGGL(k_level0_bwd<MI>, dim3(a.f.B * a.T), dim3(L0_NT), lds_bytes, q.stream, a);
    q.check_launch("level0_backward");
}
}  // namespace

bool level0_bwd_persistent_ok(const Level0Bwd& f) {
    Level0Fwd w = l0_as_fwd(f);
    w.do_max = 0;
    if (!level0_persistent_ok(w)) return false;
    L0BGeom g;
    return l0b_geometry(f, g);
}
size_t level0_bwd_vs_elems(const Level0Bwd& f) {
    long off[DP_MAX_LAYERS + 1];
    size_t total;
    l0b_vs_layout(f, off, total);
    return total + 64;
}
size_t level0_bwd_part_floats(const Level0Bwd& f) { return (size_t)(f.L > 1 ? f.L - 1 : 1) * f.B * f.N * f.G * 4 + 4; }
size_t level0_bwd_gpart_floats(const Level0Bwd& f) {
    L0BGeom g;
    if (!l0b_geometry(f, g)) return 4;
    L0BArgs a{};
    l0b_compact(f, a);
    return (size_t)f.B * g.T * a.P0 + 4;
}
void level0_backward(Seq& q, const Level0Bwd& f) {
    if (!q.ok()) return;
    L0BGeom g;
    if (!l0b_geometry(f, g)) {
        set_error("level0_backward: shape outside the persistent kernel's envelope");
        q.err = DP_ERR_UNSUPPORTED;
        return;
    }
    L0BArgs a{};
    a.f = f;
    a.T = g.T; a.RB = g.RB; a.ldp = g.ldp; a.K8 = g.K8; a.steps = g.steps;
    a.lds_dz[0] = g.lds_dz[0]; a.lds_dz[1] = g.lds_dz[1];
    a.lds_scr = g.lds_scr; a.slots_floats = g.slots_floats; a.scr_floats = g.scr_floats;
    l0b_compact(f, a);
    {
        size_t total;
        l0b_vs_layout(f, a.vs_off, total);
    }
    a.dev_err = device_error_word();
    a.spin_limit = knobs().test_barrier_fail ? 64 : L0_SPIN_LIMIT;
    a.target_bias = knobs().test_barrier_fail ? 1 : 0;
    switch (g.RB / 16) {
        case 1: l0b_launch<1>(q, a, g.lds_bytes); break;
        case 2: l0b_launch<2>(q, a, g.lds_bytes); break;
        case 3: l0b_launch<3>(q, a, g.lds_bytes); break;
        default: l0b_launch<4>(q, a, g.lds_bytes); break;
    }
}

// ---- launch timing of the persistent kernels (include/diffpool_hip.h: dp_profile_level0 / dp_profile_level0_read)
extern "C" __attribute__((visibility("default"))) int dp_profile_level0(int enable) {
    L0Prof& p = l0_prof();
    std::lock_guard<std::mutex> g(p.m);
    for (auto& v : p.ev) {
        for (auto& e : v) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        v.clear();
    }
    p.on = enable != 0;
    return DP_OK;
}
extern "C" __attribute__((visibility("default"))) int dp_profile_level0_read(int which, double* total_us, int* launches) {
    if (which < 0 || which > 1 || !total_us || !launches) return DP_ERR_INVALID_ARG;
    L0Prof& p = l0_prof();
    std::lock_guard<std::mutex> g(p.m);
    double tot = 0.0;
    int n = 0;
    for (auto& e : p.ev[which]) {
        float ms = 0.f;
        if (hipEventSynchronize(e.second) != hipSuccess || hipEventElapsedTime(&ms, e.first, e.second) != hipSuccess)
            return DP_ERR_UNSUPPORTED;
        tot += (double)ms * 1000.0;
        ++n;
    }
    *total_us = tot;
    *launches = n;
    return DP_OK;
}

#ifdef DP_STAMP
extern "C" __attribute__((visibility("default"))) int dp_debug_l0b_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_l0b_stamps), sizeof(unsigned long long) * 3 * 64);
}
#endif

#ifdef DP_STAMP
extern "C" __attribute__((visibility("default"))) int dp_debug_l0_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_l0_stamps), sizeof(unsigned long long) * 3 * 64);
}
#endif

}  // namespace dp
