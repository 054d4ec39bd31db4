// Internal helpers shared by the DiffPool HIP sources (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>

#include "../../include/diffpool_hip.h"

namespace dp {

// ----------------------------------------------------------------- error state
void set_error(const char* fmt, ...);
const char* last_error();

#define DP_CHECK_ARG(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            ::dp::set_error(__VA_ARGS__);            \
            return DP_ERR_INVALID_ARG;               \
        }                                            \
    } while (0)

// ----------------------------------------------------------------- device-side failures (see diffpool_hip.h)
// Device-visible pointer to the current device's error word (pinned, mapped host memory; nullptr when it could not be
// set up — kernels test for null), and the host-side read.  Kernels raise a bit with dev_err_raise().
int* device_error_word();
int device_error_take(bool clear);            // DP_DEVERR_* mask of the current device
int device_error_gate(const char* entry);     // DP_OK, or DP_ERR_DEVICE (+ message, word cleared) when a bit is pending
#ifdef __HIPCC__
__device__ inline void dev_err_raise(int* word, int bit) {
    if (word) __hip_atomic_fetch_or(word, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- write-through / L1-bypassing access to what other workgroups of the SAME launch write or read (raw buffer
// builtins, aux 16 = sc1: stores go through to memory, loads do not trust this XCD's L2), and the tagged 16-byte entry
// {v0, tag, v1, tag} of the barrier-free BatchNorm exchange (dp_level0.hip "tagged entries", dp_small.hip).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct ScBuf {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ ScBuf sc_buf(const void* p, size_t bytes) {
    return ScBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ u32x4 sc_ld16(ScBuf b, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(b.r, off, 0, 16); }
__device__ __forceinline__ void sc_st16(ScBuf b, unsigned off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, b.r, off, 0, 16); }
__device__ __forceinline__ float sc_ldf(ScBuf b, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.r, off, 0, 16));
}
__device__ __forceinline__ void sc_stf(ScBuf b, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), b.r, off, 0, 16);
}
__device__ __forceinline__ u32x4 sc_tagged(float v0, float v1, unsigned tag) {
    return (u32x4){__float_as_uint(v0), tag, __float_as_uint(v1), tag};
}
#endif

// ----------------------------------------------------------------- workgroup barrier for LDS hazards only
// __syncthreads() is a workgroup-scope fence: while global stores are in flight hipcc puts `s_waitcnt vmcnt(0)` in
// front of its s_barrier, so every barrier of a kernel that streams saves to global memory also pays a store round trip
// (~1-2k cycles).  Where the barrier only orders LDS traffic between the waves of the workgroup, this form waits for
// the LDS counter alone and leaves the stores flying.  NOT a release of global data to anyone.
#ifdef __HIPCC__
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif

// ----------------------------------------------------------------- cross-lane reductions on DPP
// __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe round trip per step); these run in the VALU.  A DPP row is 16
// lanes: two quad permutes, then the half-row and the row mirror, leave the row's total in every lane.  The wave
// forms add row_bcast15 / row_bcast31 (lane 63 ends up with the total of all four rows) and a readlane, so every
// lane receives the same bits.
#ifdef __HIPCC__
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_src(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, BOUND));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_src<0xB1, 0xF, true>(0.f, v);       // quad_perm [1,0,3,2]
    v += dpp_src<0x4E, 0xF, true>(0.f, v);       // quad_perm [2,3,0,1]
    v += dpp_src<0x141, 0xF, true>(0.f, v);      // row_half_mirror
    v += dpp_src<0x140, 0xF, true>(0.f, v);      // row_mirror
    return v;
}
__device__ __forceinline__ float quad4_sum(float v) {      // total of each aligned group of 4 lanes, in all 4
    v += dpp_src<0xB1, 0xF, true>(0.f, v);
    v += dpp_src<0x4E, 0xF, true>(0.f, v);
    return v;
}
__device__ __forceinline__ float oct8_sum(float v) {       // ... of 8 lanes (half a DPP row)
    v = quad4_sum(v);
    v += dpp_src<0x141, 0xF, true>(0.f, v);      // row_half_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_src<0xB1, 0xF, true>(v, v));
    v = fmaxf(v, dpp_src<0x4E, 0xF, true>(v, v));
    v = fmaxf(v, dpp_src<0x141, 0xF, true>(v, v));
    v = fmaxf(v, dpp_src<0x140, 0xF, true>(v, v));
    return v;
}
__device__ __forceinline__ float wave64_sum(float v) {
    v = row16_sum(v);
    v += dpp_src<0x142, 0xA, false>(0.f, v);     // row_bcast15 into rows 1 and 3
    v += dpp_src<0x143, 0xC, false>(0.f, v);     // row_bcast31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave64_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, dpp_src<0x142, 0xA, false>(v, v));
    v = fmaxf(v, dpp_src<0x143, 0xC, false>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif

// A launch sequence records the first HIP error it meets; later launches are skipped.
struct Seq {
    hipStream_t stream;
    char* ws;            // caller-owned workspace (bump allocated per call)
    size_t ws_bytes;
    size_t ws_off;
    int err;             // 0, DP_ERR_* (<0) or hipError_t (>0)
    bool dry;            // dry run: only walk the allocations (workspace sizing), launch nothing
    const int* pred = nullptr;   // when set: bgemm launches exit at once unless *pred != 0 (device-side fallback gate)
    // a small region (16-byte aligned, a multiple of 16 bytes, <= 4 KiB) the NEXT bgemm_group launch clears with its
    // first workgroup — saves the zero-fill launch in front of a kernel that needs a cleared flag / ticket block.
    // Whoever sets it must let a GEMM follow before the region's first reader (bgemm_group consumes and clears it).
    void* fold_zero_p = nullptr;
    int fold_zero_n16 = 0;
    // launch sequence number of the kernels that exchange tagged entries (a word of the workspace's first block, zero
    // once, counted up by every such launch): set by the model-level walks
    int* seq_word = nullptr;

    Seq(hipStream_t s, void* w, size_t wb) : stream(s), ws((char*)w), ws_bytes(wb), ws_off(0), err(0), dry(false) {}
    static Seq sizing() {
        Seq q(nullptr, nullptr, ~size_t(0) >> 1);
        q.dry = true;
        return q;
    }

    template <typename T>
    T* alloc(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        if (ws_off + bytes > ws_bytes) {
            if (!err) {
                set_error("workspace too small: need >= %zu bytes, have %zu", ws_off + bytes, ws_bytes);
                err = DP_ERR_WORKSPACE;
            }
            return nullptr;
        }
        T* p = (T*)(ws + ws_off);   // dry run: ws == nullptr, the pointer is never dereferenced
        ws_off += bytes;
        return p;
    }
    bool ok() const { return err == 0 && !dry; }
    void copy(void* dst, const void* src, size_t bytes) {
        if (err || dry || bytes == 0) return;
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) {
            set_error("hipMemcpyAsync: %s", hipGetErrorString(e));
            err = (int)e;
        }
    }
    void check_launch(const char* what) {
        if (err) return;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("%s: %s", what, hipGetErrorString(e));
            err = (int)e;
        }
    }
};

inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

// ----------------------------------------------------------------- process-wide read-only state
// Environment knobs are read ONCE, under std::call_once, on the first library call (dp_api.hip: knobs()); afterwards
// the table is read-only, so launch sequences on different host threads / streams share no mutable state.  They are
// tuning and ablation switches for bench runs; none changes results (the one switch that does — the aggregation
// kernel's phase-ablation mask DP_AGG_DEBUG — exists only in the DP_STAMP diagnostic build).
struct Knobs {
    int agg_wide;          // DP_AGG_WIDE: -1 heuristic, 0 never, 1 always take the wide aggregation kernels
    int agg_rt;            // DP_AGG_RT: 0 heuristic, 16 / 32 row tiles of the panel kernel
    int agg_debug;         // DP_AGG_DEBUG (DP_STAMP builds only; always 0 in the product library)
    bool no_pack;          // DP_NO_PACK: fp32 adjacency passes everywhere (ablation)
    bool gemm_trace;       // DP_GEMM_TRACE: host-side shape log, one line per launch
    long gemm_target_wgs;  // DP_GEMM_TARGET_WGS: workgroups wanted before GEMM tiles grow (0: default)
    int node_ksplit;       // DP_NODE_KSPLIT: 0 heuristic, 1..8 forced split-K of the node-index contractions
    bool no_head_fusion;   // DP_NO_HEAD_FUSION: pred_model on the generic GEMM
    bool no_level_fusion;  // DP_NO_LEVEL_FUSION: pooled-level GCN stacks one launch per layer
    bool no_split_gemm;    // DP_NO_SPLIT_GEMM: fp32 MFMA for every GEMM (no split-bf16 products)
    bool split_gemm_w4;    // DP_SPLIT_GEMM_W4: the 4-wave form of the split GEMM instead of the 8-wave one
    bool no_row_quads;     // DP_NO_ROW_QUADS: wide row kernels with 4-byte lanes (the pre-quad form)
    bool no_rowpart_hook;  // DP_NO_ROWPART_HOOK: BatchNorm-backward partials in a launch of their own
    bool no_widen_fusion;  // DP_NO_WIDEN_FUSION: widening layers' row-local products on the GEMM kernels
    bool no_agg_first;     // DP_NO_AGG_FIRST: every GraphConv as A (x W), also the layers that widen a lot
    bool no_l0_persist_bwd;  // DP_NO_L0_PERSIST_BWD: only the level-0 BACKWARD as the old launch sequence
    bool no_l0_persist;    // DP_NO_L0_PERSIST: level 0 as the launch sequence of rounds 1-2 instead of the persistent kernel
    bool test_barrier_fail;  // DP_TEST_BARRIER_FAIL: TEST ONLY — grid barriers wait for one arrival too many with a
                             // tiny spin limit, so the give-up path runs (tests/test_gpu_edge_cases.py)
};
const Knobs& knobs();

// Raising a kernel's dynamic-LDS limit (hipFuncSetAttribute) once per (kernel, device): thread-safe, and the
// hipError_t goes into the launch sequence instead of being dropped.  One function-local static per kernel.
struct DynLdsOnce {
    std::atomic<unsigned long long> done{0};   // bit d: device ordinal d has the attribute
    std::mutex m;
};
void ensure_dyn_lds(Seq& q, DynLdsOnce& st, const void* fn, int bytes, const char* what);

// ------------------------------------------------------------- kernel launchers
// (dp_gemm.hip)
void bgemm(Seq& q, const float* A, const float* B, float* C, const float* bias, int batch, int M, int N,
           int K, int lda, int ldb, int ldc, long sA, long sB, long sC, bool tA, bool tB, float alpha,
           float beta, int act);

// Several independent contractions (same batch count) in ONE launch: the DiffPool batches are small, so
// dependent chains of tiny GEMMs are launch/latency-bound; siblings run side by side in one grid.
#define GEMM_GROUP_MAX 4
struct GemmDesc {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    long sA, sB, sC;
    bool tA, tB;
    float alpha, beta;
    int act;
    long sK;       // split-K: partial of K range ks goes to C + ks*sK (0: all ranges share C)
    int atomic;    // split-K ranges add into C with float atomics
    // optional second output: the exact 3-plane bf16 split of C in the layout the bf16 aggregation wants
    // (Vs[b][plane][cb][k8][c][j], see dp_agg.hip) — lets the producer of an aggregation operand emit it
    // directly instead of a separate split pass.  Rows M..8*split_k8-1 are written as zeros.
    unsigned short* split_out;
    int split_ct, split_k8, split_c0;   // total column blocks, k8 groups, column offset of this problem in V
    int nosplit;   // this problem walks its whole K in one workgroup even when the launch is split-K (grouped siblings)
    // deterministic split-K (instead of `atomic`): every K range stores its partial tile to fix_part
    // ([batch][ksplit][M][N] floats), takes a ticket from fix_cnt (one zeroed int per (batch, 16x16 block row/col
    // pair upper bound: see gemm_fix_counters)), and the LAST range to arrive adds the partials in range order and
    // writes C.  Bit-reproducible, one launch, no workgroup ever waits for another.
    float* fix_part;
    int* fix_cnt;
    // optional row partials of the FINAL C (after beta / accumulation): part[((b * M + row) * rp_G + rp_g) * 2 + {0, 1}] =
    // (sum_c C[row, c], sum_c C[row, c] * xhat[row, c]) — the BatchNorm-backward partials of the layer whose gradient
    // this product completes (k_bn_bwd_partials without its launch).  Needs N <= 32, nosplit, no atomic
    // (gemm_rowpart_ok); xhat rows are rp_ldx apart, graphs M * rp_ldx.
    const float* rp_xhat;
    int rp_ldx;
    float* rp_part;
    int rp_G, rp_g;
};
inline bool gemm_rowpart_ok(int N) { return N >= 1 && N <= 32; }
// ints of fix_cnt a problem needs (an upper bound over every tile shape the launcher may pick)
inline size_t gemm_fix_counters(int batch, int M, int N) {
    return (size_t)batch * ((M + 15) / 16) * ((N + 15) / 16);
}

// hi/mid/lo bf16 planes with hi + mid + lo == v exactly (round-to-nearest-even at each step)
__device__ inline void bf16_split3(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    auto rn = [](float x) -> unsigned {
        unsigned u = __float_as_uint(x);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return u >> 16;
    };
    const unsigned hh = rn(v);
    const float r1 = v - __uint_as_float(hh << 16);
    const unsigned mm = rn(r1);
    const float r2 = r1 - __uint_as_float(mm << 16);
    h = (unsigned short)hh;
    m = (unsigned short)mm;
    l = (unsigned short)rn(r2);
}
// element offset of Vs[plane][cb][k8][c][j] inside one graph's block
__device__ __host__ inline long vs_index(int plane, int CTt, int K8, int cb, int k8, int c, int j) {
    return ((((long)plane * CTt + cb) * K8 + k8) * 16 + c) * 8 + j;
}
// ksplit > 1 cuts K into ranges run by different workgroups (contractions over the node index have K = n
// = 500 but outputs of a few KB: without it they are 40-80 workgroups walking 16 dependent k-steps).
void bgemm_group(Seq& q, const GemmDesc* d, int count, int batch, int ksplit = 1);
// (dp_gemm_split.hip) the same contraction with both operands split into three bf16 planes in registers and multiplied
// on the bf16 matrix cores (six plane products, fp32 accumulation): fp32-grade results at ~2x the fp32-MFMA rate.
// Large shapes only, no epilogue options (bias / act / split-K / split_out).
bool gemm_split_usable(const GemmDesc& d, int batch, int ksplit);
void gemm_split_bf16(Seq& q, const GemmDesc& d, int batch);

// Column groups of a row: the level-j embed and assign GCN stacks share one pass over the
// adjacency, so row-wise kernels work on up to two column groups of a joint buffer.
struct RowGroups {
    int G;               // 1 or 2
    int c0[2];           // first column of the group in the joint buffer
    int w[2];            // group width
};
struct GroupPtrs {       // per-group pointer + leading dimension (already offset to the group's first column)
    float* p[2];
    int ld[2];
};
struct GroupCPtrs {
    const float* p[2];
    int ld[2];
};

// (dp_rowops.hip)
void rownorm_fwd(Seq& q, const float* U, int ldu, const float* P /*add_self or null*/, GroupCPtrs bias,
                 RowGroups g, GroupPtrs yout, float* invn, float* part /*[rows,G,2] or null*/, long rows,
                 int normalize, int relu_stats);
// Bs: number of graphs whose row partials `part` holds ([Bs, n, G, 2]); 0 = B.  Bs > B is sync-BN over several
// data-parallel ranks: the statistics span every rank's batch, the rows normalised are this rank's B graphs.
void bn_apply_fwd(Seq& q, const float* Y, int ldy, const float* part /*[Bs,n,G,2]; null: no BN*/, float* stats,
                  RowGroups g, GroupPtrs xout, int B, int n, int relu, int Bs = 0);
// apply_bn of one layer + the next layer's transform P = x W in one launch (row-local; small batches, widths <= 64)
bool bn_transform_supported(RowGroups gin, RowGroups gout, int B);
void bn_transform_fwd(Seq& q, const float* Y, int ldy, const float* part, float* stats, RowGroups gin, GroupPtrs xout,
                      const float* const W[2], RowGroups gout, float* P, int ldp, int B, int n, int relu,
                      unsigned short* vs /*also emit the 3-plane bf16 split of P, or null*/);
void mask_rows(Seq& q, const float* src, int lds, float* dst, int ldd, const int* num_nodes, int B, int n, int F);
void bn_bwd_partials(Seq& q, GroupCPtrs dx, GroupCPtrs xhat, RowGroups g, float* part, long rows);
void rownorm_bwd(Seq& q, GroupCPtrs dx, GroupCPtrs xhat /*BN output, null when no BN*/, GroupCPtrs y,
                 const float* invn, const float* stats, const float* part2, RowGroups g, float* dU, int ldu,
                 const GroupPtrs* dbias /*per group: bias-gradient slab of graph 0 (ld = distance between graphs);
                 column sums of dU are atomically added; null: none*/, int B, int n, int has_relu, int has_bn,
                 int normalize, unsigned short* vs = nullptr /*also emit the 3-plane bf16 split of dU*/,
                 int Bs = 0 /*graphs in part2 (sync-BN: B x ranks); 0 = B*/);
int rownorm_bwd_chunks(int n);
void colsum_batched(Seq& q, const float* X, int ldx, long strideX, int rows, int cols, float* out,
                    long strideOut, int batch, int rowsplit = 1);
// vs: also emit the 3-plane bf16 split of S (dp_agg.hip layout); zero_p/zero_bytes: also zero-fill that region
void softmax_mask_fwd(Seq& q, const float* logits, int ldl, float* S, int lds, const int* num_nodes, int B,
                      int n, int K, float* S2 = nullptr, unsigned short* vs = nullptr, void* zero_p = nullptr,
                      size_t zero_bytes = 0);
// dbias: slab row of graph 0 for the column sums of dlogits (graphs dbias_stride floats apart, atomically added)
void softmax_mask_bwd(Seq& q, const float* S, int lds, const float* dS, int ldds, const int* num_nodes,
                      float* dlogits, int ldl, int B, int n, int K, float* dbias = nullptr, long dbias_stride = 0,
                      const float* dS2 = nullptr /*second addend of dS, same leading dimension*/);
void masked_max_fwd(Seq& q, const float* Z, int ldz, const int* num_nodes, float* out, int ldo, int* argmax,
                    int lda, int B, int n, int F);
void masked_max_bwd(Seq& q, const float* dout, int ldo, const int* argmax, int lda, float* dZ, int ldz, int B,
                    int n, int F);
void relu_bwd_inplace(Seq& q, float* d, const float* h, long count);
void argmax_rows(Seq& q, const float* X, int ldx, long long* out, int rows, int cols);   // first index on ties
void ce_fwd(Seq& q, const float* logits, const long long* label, float* loss, float* prob, int B, int C,
            float* also_zero = nullptr /*a second scalar to clear in the same launch*/,
            float* dunit = nullptr /*[B, C]: d loss / d logits for an upstream gradient of 1*/);
void ce_bwd(Seq& q, const float* prob, const long long* label, const float* dloss, float scale, float* dlogits,
            int B, int C);
void reduce_slabs(Seq& q, const float* slabs, long stride, int B, float* out, long count, int accumulate);
void axpy(Seq& q, float* y, const float* x, float a, long count);
void mask_mul(Seq& q, const float* x, int ldx, const float* m, float* out, long rows, int w);
void mask_axpy(Seq& q, float* dst, int ldd, const float* src, const float* m, long rows, int w);
bool rownorm_bwd_mv_supported(RowGroups g, const int din[2], int n);
void rownorm_bwd_mv(Seq& q, GroupCPtrs dx, GroupCPtrs xhat, GroupCPtrs y, const float* invn, const float* stats,
                    const float* part2, RowGroups g, float* dU, int ldu, const GroupPtrs* dbias, int B, int n,
                    int has_relu, int has_bn, int normalize, int Bs, const float* const W[2], const int din[2],
                    const int c0in[2], float* dUin, int lddu);
bool widen_fwd_supported(RowGroups g, const int din[2]);
void widen_fwd(Seq& q, const float* Uin, int ldin, const int c0in[2], const int din[2], const float* const W[2],
               GroupCPtrs bias, RowGroups g, GroupPtrs yout, float* invn, long rows, int normalize);
void gather_cols(Seq& q, const float* x0, int ld0, int w0, const float* x1, int ld1, int w1, float* out, long rows);
bool scatter_add_cols_part_supported(int w0, int w1);
void scatter_add_cols_part(Seq& q, const float* src, GroupPtrs d, GroupCPtrs xhat, int w0, int w1, int G, float* part,
                           long rows);
void scatter_add_cols(Seq& q, const float* src, float* d0, int ld0, int w0, float* d1, int ld1, int w1, long rows);
void zero_fill(Seq& q, void* p, size_t bytes);   // wide-store zero kernel (byte kernel for odd sizes); never a memset node
void zero_small(Seq& q, void* p, size_t bytes);

// (dp_agg.hip) adjacency-panel aggregation
struct PackedAdj {               // written by adj_pack: bf16 copies of A and A^T + the exactness flag
    const unsigned short* A;     // [B, n, ld]
    const unsigned short* At;    // [B, n, ld]
    int ld;                      // adj_pack_ld(n)
    const int* flag;             // device int: 0 = every entry of A is exactly representable in bf16
};
// flag_zeroed: the 256-byte flag block was already cleared in stream order (Seq::fold_zero_p through a GEMM);
// zero_p / zero_bytes: an unrelated region the pack kernel's workgroups clear on the side (the backward pass's
// accumulators: an HBM-bound kernel with 1000+ workgroups absorbs a few MB of stores for free)
void adj_pack(Seq& q, const float* A, unsigned short* P, unsigned short* Pt, int* flag, int B, int n, int ld,
              bool flag_zeroed = false, void* zero_p = nullptr, size_t zero_bytes = 0);
int adj_pack_ld(int n);
bool adj_pack_supported(int n, int C);
size_t split3_elems(int B, int n, int C);
// vs_ready: the producer of V already wrote its 3-plane split into `vs` (no split pass needed)
void aggregate(Seq& q, const float* A, const float* V, int ldv, float* U, int ldu, int B, int n, int C, bool trans,
               float beta, const PackedAdj* pk = nullptr, unsigned short* vs = nullptr, bool vs_ready = false);
bool aggregate_rownorm_fwd(Seq& q, const float* A, const float* V, int ldv, const float* P, GroupCPtrs bias,
                           RowGroups g, GroupPtrs yout, float* invn, float* part, int B, int n, int normalize,
                           int stats_mode, const PackedAdj* pk = nullptr, unsigned short* vs = nullptr,
                           bool vs_ready = false);
bool aggregate_packed_usable(const float* A, int n, int C);

// (dp_small.hip) one-workgroup-per-graph GCN layers of a pooled level (n <= 64)
bool small_level_supported(int B, int n, int din, int dout);
void small_gcn_fwd(Seq& q, const float* adj, const float* x0, int ldx0, const float* yprev, int ldyp,
                   const float* part_prev, float* stats_prev, float* xout, int ldxo, const float* W, const float* bias,
                   float* y, int ldy, float* invn, float* part, int B, int n, int din, int dout, int add_self,
                   int stats);
void small_gcn_bwd(Seq& q, const float* adj, const float* xin, int ldxin, const float* W, const float* y, int ldy,
                   const float* xhat, int ldxh, const float* invn, const float* stats, const float* part2,
                   const float* dx, int lddx, float* dxin, int lddxin, float* part2_prev, float* dadj, float* dW,
                   float* db, long slab_stride, int B, int n, int din, int dout, int add_self, int has_bn,
                   int has_relu);

// whole-level variants: every layer of the level's stack in ONE launch per direction (grid barrier between layers
// for the cross-graph BatchNorm statistics).  ld / offsets as the per-layer calls would pass them.
struct SmallLevelIO {
    const float* adj;              // [B, n, n]
    const float* x0;               // level input [B, n, dims[0]]
    int ldx0;
    const float* params;
    long w_off[DP_MAX_LAYERS], b_off[DP_MAX_LAYERS];
    float* Y[DP_MAX_LAYERS];       // non-last layers: normalised pre-ReLU output, ld ldY[l]; null for the last layer
    int ldY[DP_MAX_LAYERS];
    float* invn[DP_MAX_LAYERS];    // [B, n]
    float* stats[DP_MAX_LAYERS];   // [n, 2] of the non-last layers
    float* Ze;                     // concat buffer [B, n, ldz], slice l at column coff[l]
    int ldz;
    int coff[DP_MAX_LAYERS];
    float* part;                   // exchange scratch, small_level_part_floats() floats
    int* bar;                      // [0] spare, [1] error word, [2] finish ticket (zeroed in stream order before the launch)
};
bool small_level_fused_ok(int B, int n, const int* dims, int L, bool dadj);
size_t small_level_part_floats(int B, int n, int L);
void small_level_fwd(Seq& q, const SmallLevelIO& io, int B, int n, const int* dims, int L, int add_self, int bn);
void small_level_bwd(Seq& q, const SmallLevelIO& io, const float* dZe, float* dX0, float* dadj, float* slabs,
                     long slab_stride, int B, int n, const int* dims, int L, int add_self, int bn);

// (dp_level0.hip) the whole level-0 forward — adjacency pack, every GraphConv layer of the embed + assign stacks with
// BatchNorm, max readout, assign head + softmax, T = A^T S, X' = S^T Z, A' = T^T S — as ONE persistent launch: a
// workgroup per (graph, block of RB rows) keeps its rows of the bf16 adjacency in LDS across all passes; what crosses
// workgroups (the split operand of each aggregation, BatchNorm partials, the pooled partial products) travels through
// global memory behind per-graph / grid-wide barriers.
struct L0Stack {
    int dims[DP_MAX_LAYERS + 1];
    long w_off[DP_MAX_LAYERS], b_off[DP_MAX_LAYERS];
};
struct Level0Fwd {
    int B, N, L, G;                // G = 2: embed + assign stacks (a pooling level follows); 1: embed only
    int bn, train, do_max, mask_readout;
    const float* A;                // [B, N, N] fp32 as delivered
    const float* x0[2];            // per stack: [B, N, st[g].dims[0]] contiguous
    const int* num_nodes;          // int32[B] or null
    const float* params;
    L0Stack st[2];
    float* Y[DP_MAX_LAYERS];       // non-last layers: joint normalised pre-ReLU output [B, N, ctot[l]]
    float* invn[DP_MAX_LAYERS];    // [B, N, G]
    float* stats[DP_MAX_LAYERS];   // [N, G, 2] (mu, rstd) of the non-last layers
    float* Z[2];                   // concat buffers Ze [B, N, ldz[0]], Za [B, N, ldz[1]]
    int ldz[2];
    int coff[2][DP_MAX_LAYERS];    // column of layer l's slice in Z[g]
    // pooling (G == 2)
    int K;
    long wp_off, bp_off;           // assign_pred Linear [K, Da], [K] (bp_off may be -1)
    float* S;                      // [B, N, K]
    float* S2;                     // caller-visible copy or null
    float* Tt;                     // [B, N, K] = A^T S
    float* Xn;                     // [B, K, D]
    float* An;                     // [B, K, K]
    // max readout (do_max)
    float* feat;                   // [B, ldfeat], this level's slice at featoff
    int ldfeat, featoff, rw, zoff; // readout width, first Ze column it covers
    int* argmax;                   // [B, rw]
    // packed adjacency for the backward pass
    unsigned short *pkA, *pkAt;
    int pk_ld;
    int* pk_flag;                  // 256-byte block: word 0 = "some entry is not bf16-exact", rest zero
    // exchange scratch (workspace)
    unsigned short* vs;            // level0_vs_elems(): the 3-plane split operand of every aggregation pass (write-once regions)
    float* part;                   // level0_part_floats(): [L - 1][B, N, G, 2]
    float* xpart;                  // [B, T, epad] partial X' | A'
    float* mpart;                  // [B, T, rw, 2] partial (max, row)
    int* bar;                      // level0_bar_ints(B) ints, zero on first use, self-cleaning afterwards
    int* next_bar;                 // 64 ints cleared for the launch that follows (pooled level's grid barrier), or null
    void* zero_p;                  // side job: clear this region (the backward accumulators), 16-byte aligned
    size_t zero_bytes;
};
// ... and its mirror: the whole level-0 backward (pooling products, A V, softmax / assign head backward, every GraphConv
// layer with BatchNorm backward, A^T dU, weight / bias / input gradients) in ONE persistent launch; the parameter
// gradients of a graph leave as ONE slab row, combined over the graph's row blocks in block order (deterministic).
struct Level0Bwd {
    int B, N, L, G, bn;
    const float* A;                // fp32 adjacency (only read when it is not bf16-exact)
    const float* x0[2];
    const float* params;
    L0Stack st[2];
    const float* Y[DP_MAX_LAYERS];
    const float* invn[DP_MAX_LAYERS];
    const float* stats[DP_MAX_LAYERS];
    const float* Z[2];
    int ldz[2];
    int coff[2][DP_MAX_LAYERS];
    int K;
    long wp_off, bp_off;
    const float* S;                // [B, N, K]
    const float* Tt;               // [B, N, K]
    const float* dXn;              // [B, K, D]   gradient of the pooled features
    const float* dAn;              // [B, K, K]   gradient of the pooled adjacency
    const float* d_assign;         // [B, N, K] extra gradient of S (link loss) or null
    const float* dZe;              // [B, N, D]   gradient of the embedding concat (max-readout scatter)
    const unsigned short *pkA, *pkAt;
    int pk_ld;
    const int* pk_flag;
    float* slabs;                  // slab row of graph 0 (flat parameter offsets); graphs are slab_gstride floats apart
    long slab_gstride;
    unsigned short* vs;            // level0_bwd_vs_elems()
    float* part;                   // level0_part_floats()-sized
    float* gpart;                  // level0_bwd_gpart_floats()
    int* bar;                      // the forward's barrier block (same contract)
};
bool level0_bwd_persistent_ok(const Level0Bwd& a);
size_t level0_bwd_vs_elems(const Level0Bwd& a);
size_t level0_bwd_part_floats(const Level0Bwd& a);
size_t level0_bwd_gpart_floats(const Level0Bwd& a);
void level0_backward(Seq& q, const Level0Bwd& a);
bool level0_persistent_ok(const Level0Fwd& a);
size_t level0_bar_ints(int B);
int* level0_seq_word(int* bar);
size_t level0_xpart_floats(const Level0Fwd& a);
size_t level0_vs_elems(const Level0Fwd& a);
size_t level0_part_floats(const Level0Fwd& a);
size_t level0_mpart_floats(const Level0Fwd& a);
const int* level0_error_word(const int* bar);        // device word the head launch checks (ypred = NaN when set)
void level0_forward(Seq& q, const Level0Fwd& a);

// (dp_head.hip) last-level max readout + pred_model in one launch per direction
struct HeadArgs {
    const float* Z;          // last level's embedding [B, n, ldz] for the in-kernel max readout, or null
    int ldz, n, rw, featoff; // readout width and its column offset in the feature vector
    int* argmax;             // [B, lda]
    int lda;
    const float* params;
    int n_pred;
    int dims[DP_MAX_PRED + 2];
    long w_off[DP_MAX_PRED + 1], b_off[DP_MAX_PRED + 1];
    float* hid[DP_MAX_PRED + 2];   // hid[0] = features [B, dims[0]], hid[i] = post-ReLU activations, hid[n_pred] = ypred
    int B;
    long long* labels;             // optional: arg-max class per graph (evaluation)
    const int* poison[DP_MAX_LEVELS + 1];   // device error words of this forward's grid barriers (n_poison of them):
    int n_poison;                           // any non-zero -> ypred = NaN.  Max readouts and ReLUs swallow NaN, so the
                                            // poisoned BatchNorm statistics alone may not reach ypred
};
struct HeadBwdArgs {
    HeadArgs h;              // dims / offsets / saved activations (hid[] read-only here)
    const float* d_ypred;    // [B, dims[n_pred]]
    float* grads;            // flat gradient buffer (pred_model entries are written, not accumulated)
    int n_levels;
    struct Level {
        float* dZ;           // zero-initialised [B, n, ldz] (already offset to the first readout column)
        const int* argmax;
        int lda, n, ldz, rw, featoff;
    } lv[DP_MAX_LEVELS + 1];
};
bool head_supported(const HeadArgs& a);
void head_fwd(Seq& q, const HeadArgs& a);
void head_bwd(Seq& q, const HeadBwdArgs& a);

// (dp_batch.hip)
void build_batch(Seq& q, const int* src, const int* dst, const int* edge_ptr, const int* label, const int* node_ptr,
                 float* adj, unsigned short* pk, unsigned short* pkt, float* feats, float* assign, int* num_nodes,
                 int* errors, int* degree, int B, int N, int F, int mode, int symmetric, int max_edges_per_graph);

void gather_labels(Seq& q, const long long* src, long long* dst, int B);

// (dp_optim.hip)
void clip_adam_step(Seq& q, float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float max_norm,
                    float beta1, float beta2, float eps, float step_size, float inv_bc2_sqrt, float* total_norm_out,
                    int* step_counter = nullptr, float lr = 0.f);

// (dp_meanagg.hip) CSR neighbour aggregation: out[i] = (mean | sum) of table[indices[indptr[i]:indptr[i+1]]] (+ beta out)
void csr_aggregate_fwd(Seq& q, const float* table, int ldt, const int* indptr, const int* indices, float* out, int ldo,
                       int n_rows, int feat, int mean, float beta);
void csr_aggregate_bwd_scatter(Seq& q, const float* dout, int ldo, const int* indptr, const int* indices, float* dtable,
                               int ldt, int n_rows, int feat, int mean);

// (dp_linkpred.hip)
void linkpred_fwd(Seq& q, const float* S, int lds, const float* adj, const int* num_nodes, float* loss_out,
                  int B, int n, int K, const float* norm = nullptr /*device scalar replacing sum n_b^2*/);
void linkpred_bwd(Seq& q, const float* S, int lds, const float* adj, const int* num_nodes, const float* dloss,
                  float* dS, int ldds, int B, int n, int K, int accumulate, const float* norm = nullptr);

}  // namespace dp
