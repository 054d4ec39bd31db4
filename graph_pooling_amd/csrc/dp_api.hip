// extern "C" boundary of libdiffpool_hip.so: argument checks, then the launch sequences.
#include <cstdarg>

#include <cmath>

#include "dp_common.h"

namespace dp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

const Knobs& knobs() {
    static Knobs k;
    static std::once_flag once;
    std::call_once(once, [] {
        auto num = [](const char* name, long dflt) {
            const char* e = getenv(name);
            return e ? atol(e) : dflt;
        };
        k.agg_wide = (int)num("DP_AGG_WIDE", -1);
        k.agg_rt = (int)num("DP_AGG_RT", 0);
#ifdef DP_STAMP
        k.agg_debug = (int)num("DP_AGG_DEBUG", 0);
#else
        k.agg_debug = 0;
#endif
        k.no_pack = getenv("DP_NO_PACK") != nullptr;
        k.gemm_trace = getenv("DP_GEMM_TRACE") != nullptr;
        k.gemm_target_wgs = num("DP_GEMM_TARGET_WGS", 0);
        k.node_ksplit = (int)num("DP_NODE_KSPLIT", 0);
        k.no_head_fusion = getenv("DP_NO_HEAD_FUSION") != nullptr;
        k.no_level_fusion = getenv("DP_NO_LEVEL_FUSION") != nullptr;
        k.no_split_gemm = getenv("DP_NO_SPLIT_GEMM") != nullptr;
        k.split_gemm_w4 = getenv("DP_SPLIT_GEMM_W4") != nullptr;
        k.no_agg_first = getenv("DP_NO_AGG_FIRST") != nullptr;
        k.no_widen_fusion = getenv("DP_NO_WIDEN_FUSION") != nullptr;
        k.no_rowpart_hook = getenv("DP_NO_ROWPART_HOOK") != nullptr;
        k.no_row_quads = getenv("DP_NO_ROW_QUADS") != nullptr;
        k.test_barrier_fail = getenv("DP_TEST_BARRIER_FAIL") != nullptr;
        k.no_l0_persist = getenv("DP_NO_L0_PERSIST") != nullptr;
        k.no_l0_persist_bwd = getenv("DP_NO_L0_PERSIST_BWD") != nullptr;
    });
    return k;
}

// ---- device-side failure word: one pinned, device-mapped host int per device ordinal (diffpool_hip.h)
namespace {
struct DevErrSlot {
    std::once_flag once;
    int* host = nullptr;
    int* dev = nullptr;
};
DevErrSlot g_dev_err[64];
DevErrSlot* dev_err_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
    DevErrSlot& s = g_dev_err[d];
    std::call_once(s.once, [&s] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return; }
        memset(h, 0, 64);
        void* dv = nullptr;
        if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        s.host = (int*)h;
        s.dev = (int*)dv;
    });
    return s.host ? &s : nullptr;
}
}  // namespace
int* device_error_word() {
    DevErrSlot* s = dev_err_slot();
    return s ? s->dev : nullptr;
}
int device_error_take(bool clear) {
    DevErrSlot* s = dev_err_slot();
    if (!s) return 0;
    return clear ? __atomic_exchange_n(s->host, 0, __ATOMIC_ACQ_REL) : __atomic_load_n(s->host, __ATOMIC_ACQUIRE);
}
static const char* device_error_text(int mask) {
    if ((mask & DP_DEVERR_BARRIER) && (mask & DP_DEVERR_NONFINITE_GRAD))
        return "a whole-level kernel's grid barrier gave up (its workgroups were not co-resident: is the device shared? "
               "set DP_NO_LEVEL_FUSION=1) and dp_clip_adam_step refused the resulting non-finite gradients";
    if (mask & DP_DEVERR_BARRIER)
        return "a whole-level kernel's grid barrier gave up: its workgroups were not co-resident (is the device shared "
               "with another process or stream? set DP_NO_LEVEL_FUSION=1); that step's outputs are NaN";
    if (mask & DP_DEVERR_NONFINITE_GRAD)
        return "dp_clip_adam_step met a non-finite gradient norm and skipped the update (parameters untouched)";
    return mask ? "unknown device error bits" : "no device error";
}
int device_error_gate(const char* entry) {
    const int m = device_error_take(false);
    if (!m) return DP_OK;
    device_error_take(true);
    set_error("%s: a kernel of an EARLIER call on this device failed (mask %d): %s", entry, m, device_error_text(m));
    return DP_ERR_DEVICE;
}

void ensure_dyn_lds(Seq& q, DynLdsOnce& st, const void* fn, int bytes, const char* what) {
    if (q.err || q.dry) return;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && dev >= 0 && dev < 64 &&
        (st.done.load(std::memory_order_acquire) >> dev & 1ull))
        return;
    std::lock_guard<std::mutex> lock(st.m);
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(max dynamic LDS = %d): %s", what, bytes, hipGetErrorString(e));
        q.err = (int)e;
        return;
    }
    if (dev >= 0 && dev < 64) st.done.fetch_or(1ull << dev, std::memory_order_release);
}

// dp_model.hip
int encoder_forward(Seq& q, const dp_encoder_cfg& c, const float* params, const float* x, const float* adj,
                    const float* assign_x, const int* num_nodes, const float* dropout, float* ypred,
                    float* assign_out, void* save, int mode, long long* labels_out, const PackedAdj* given = nullptr);
int encoder_backward(Seq& q, const dp_encoder_cfg& c, const float* params, const float* x, const float* adj,
                     const float* assign_x, const int* num_nodes, const float* dropout, const float* d_ypred,
                     const float* d_assign, float* grads, const void* save, int prezeroed,
                     const PackedAdj* given = nullptr);
size_t encoder_save_bytes(const dp_encoder_cfg& c);
int encoder_validate(const dp_encoder_cfg* c);
int encoder_save_locate(const dp_encoder_cfg& c, int level, int field, size_t* offset, size_t* count);
// dp_set2set.hip
void set2set_fwd(Seq& q, const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                 const float* b_hh, const float* Wp, const float* bp, float* out, int B, int n, int d, void* save);
void set2set_bwd(Seq& q, const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                 const float* b_hh, const float* Wp, const float* bp, const float* out, const float* dout,
                 float* demb, int ldde, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dWp,
                 float* dbp, int B, int n, int d, const void* save);
size_t set2set_save_bytes(int B, int n, int d);
// dp_meanagg.hip
void mean_aggregate_fwd(Seq& q, const float* table, int ldt, const int* indptr, const int* indices, float* out,
                        int ldo, int n_rows, int feat);
void mean_aggregate_bwd(Seq& q, const float* dout, int ldo, const int* indptr, const int* indices, float* dtable,
                        int ldt, int n_rows, int feat);

namespace {

RowGroups one_group(int w) {
    RowGroups g{};
    g.G = 1;
    g.c0[0] = 0;
    g.w[0] = w;
    return g;
}
GroupPtrs gp(float* p, int ld) {
    GroupPtrs r{};
    r.p[0] = p;
    r.ld[0] = ld;
    return r;
}
GroupCPtrs gcp(const float* p, int ld) {
    GroupCPtrs r{};
    r.p[0] = p;
    r.ld[0] = ld;
    return r;
}

// ---- op-level sequences (each usable in dry mode for workspace sizing)
void gcn_layer_fwd_seq(Seq& q, const float* x, int ldx, const float* adj, const float* W, const float* bias, float* y,
                       int ldy, float* invn, int B, int n, int Fin, int Fout, int flags) {
    float* P = q.alloc<float>((size_t)B * n * Fout);
    float* U = q.alloc<float>((size_t)B * n * Fout);
    if (q.err) return;
    bgemm(q, x, W, P, nullptr, B, n, Fout, Fin, ldx, Fout, Fout, (long)n * ldx, 0, (long)n * Fout, false, false, 1.f,
          0.f, 0);
    const float* selfp = (flags & DP_F_ADD_SELF) ? P : nullptr;
    const int norm = (flags & DP_F_NORMALIZE) ? 1 : 0;
    if (!aggregate_rownorm_fwd(q, adj, P, Fout, selfp, gcp(bias, 0), one_group(Fout), gp(y, ldy), invn, nullptr, B, n,
                               norm, 0)) {
        bgemm(q, adj, P, U, nullptr, B, n, Fout, n, n, Fout, Fout, (long)n * n, (long)n * Fout, (long)n * Fout, false,
              false, 1.f, 0.f, 0);
        rownorm_fwd(q, U, Fout, selfp, gcp(bias, 0), one_group(Fout), gp(y, ldy), invn, nullptr, (long)B * n, norm, 0);
    }
}

void gcn_layer_bwd_seq(Seq& q, const float* x, int ldx, const float* adj, const float* W, const float* y, int ldy,
                       const float* invn, const float* dy, int lddy, float* dx, int lddx, float* dW, float* db,
                       float* dadj, int B, int n, int Fin, int Fout, int flags) {
    float* dU = q.alloc<float>((size_t)B * n * Fout);
    float* G = q.alloc<float>((size_t)B * n * Fout);
    float* P = dadj ? q.alloc<float>((size_t)B * n * Fout) : nullptr;
    if (q.err) return;
    const int norm = (flags & DP_F_NORMALIZE) ? 1 : 0;
    rownorm_bwd(q, gcp(dy, lddy), gcp(nullptr, 0), gcp(y, ldy), invn, nullptr, nullptr, one_group(Fout), dU, Fout,
                nullptr, B, n, 0, 0, norm);
    if (db) colsum_batched(q, dU, Fout, 0, B * n, Fout, db, 0, 1);
    aggregate(q, adj, dU, Fout, G, Fout, B, n, Fout, true, 0.f);
    if (flags & DP_F_ADD_SELF) axpy(q, G, dU, 1.f, (long)B * n * Fout);
    // dW = sum_b x_b^T G_b: one contraction over K = B*n rows (x rows of consecutive graphs are ldx apart)
    bgemm(q, x, G, dW, nullptr, 1, Fin, Fout, B * n, ldx, Fout, Fout, 0, 0, 0, true, false, 1.f, 0.f, 0);
    if (dx)
        bgemm(q, G, W, dx, nullptr, B, n, Fin, Fout, Fout, Fout, lddx, (long)n * Fout, 0, (long)n * lddx, false, true,
              1.f, 0.f, 0);
    if (dadj) {
        bgemm(q, x, W, P, nullptr, B, n, Fout, Fin, ldx, Fout, Fout, (long)n * ldx, 0, (long)n * Fout, false, false,
              1.f, 0.f, 0);
        bgemm(q, dU, P, dadj, nullptr, B, n, n, Fout, Fout, Fout, n, (long)n * Fout, (long)n * Fout, (long)n * n, false,
              true, 1.f, 0.f, 0);
    }
}

void bn_fwd_seq(Seq& q, const float* x, int ldx, float* y, int ldy, float* stats, int B, int n, int F, int relu) {
    float* part = q.alloc<float>((size_t)B * n * 2);
    float* tmp = q.alloc<float>((size_t)B * n * F);
    if (q.err) return;
    RowGroups g = one_group(F);
    rownorm_fwd(q, x, ldx, nullptr, gcp(nullptr, 0), g, gp(tmp, F), nullptr, part, (long)B * n, 0, relu ? 1 : 2);
    bn_apply_fwd(q, tmp, F, part, stats, g, gp(y, ldy), B, n, relu);
}

void bn_bwd_seq(Seq& q, const float* x, int ldx, const float* y, int ldy, const float* stats, const float* dy, int lddy,
                float* dx, int lddx, int B, int n, int F, int relu) {
    float* part = q.alloc<float>((size_t)B * n * 2);
    if (q.err) return;
    RowGroups g = one_group(F);
    bn_bwd_partials(q, gcp(dy, lddy), gcp(y, ldy), g, part, (long)B * n);
    // "y" operand of rownorm_bwd is only used for the ReLU mask: the forward input x
    rownorm_bwd(q, gcp(dy, lddy), gcp(y, ldy), gcp(relu ? x : y, relu ? ldx : ldy), nullptr, stats, part, g, dx, lddx,
                nullptr, B, n, relu, 1, 0);
}

void assign_fwd_seq(Seq& q, const float* z, int ldz, const float* Wp, const float* bp, const int* num_nodes, float* S,
                    int B, int n, int Din, int K) {
    float* logits = q.alloc<float>((size_t)B * n * K);
    if (q.err) return;
    bgemm(q, z, Wp, logits, bp, B, n, K, Din, ldz, Din, K, (long)n * ldz, 0, (long)n * K, false, true, 1.f, 0.f, 0);
    softmax_mask_fwd(q, logits, K, S, K, num_nodes, B, n, K);
}

void assign_bwd_seq(Seq& q, const float* z, int ldz, const float* Wp, const float* S, const float* dS,
                    const int* num_nodes, float* dz, int lddz, float* dWp, float* dbp, int B, int n, int Din, int K) {
    float* dlog = q.alloc<float>((size_t)B * n * K);
    if (q.err) return;
    softmax_mask_bwd(q, S, K, dS, K, num_nodes, dlog, K, B, n, K);
    bgemm(q, dlog, z, dWp, nullptr, 1, K, Din, B * n, K, ldz, Din, 0, 0, 0, true, false, 1.f, 0.f, 0);
    if (dbp) colsum_batched(q, dlog, K, 0, B * n, K, dbp, 0, 1);
    if (dz)
        bgemm(q, dlog, Wp, dz, nullptr, B, n, Din, K, K, Din, lddz, (long)n * K, 0, (long)n * lddz, false, false, 1.f,
              0.f, 0);
}

void pool_fwd_seq(Seq& q, const float* S, const float* Z, int ldz, const float* adj, float* Xp, float* Ap, float* T,
                  int B, int n, int K, int D) {
    bgemm(q, S, Z, Xp, nullptr, B, K, D, n, K, ldz, D, (long)n * K, (long)n * ldz, (long)K * D, true, false, 1.f, 0.f,
          0);
    bgemm(q, S, adj, T, nullptr, B, K, n, n, K, n, n, (long)n * K, (long)n * n, (long)K * n, true, false, 1.f, 0.f, 0);
    bgemm(q, T, S, Ap, nullptr, B, K, K, n, n, K, K, (long)K * n, (long)n * K, (long)K * K, false, false, 1.f, 0.f, 0);
}

void pool_bwd_seq(Seq& q, const float* S, const float* Z, int ldz, const float* adj, const float* T, const float* dXp,
                  const float* dAp, float* dS, float* dZ, int lddz, float* dadj, int B, int n, int K, int D) {
    float* V = q.alloc<float>((size_t)B * n * K);
    if (q.err) return;
    bgemm(q, S, dXp, dZ, nullptr, B, n, D, K, K, D, lddz, (long)n * K, (long)K * D, (long)n * lddz, false, false, 1.f,
          1.f, 0);
    bgemm(q, Z, dXp, dS, nullptr, B, n, K, D, ldz, D, K, (long)n * ldz, (long)K * D, (long)n * K, false, true, 1.f, 0.f,
          0);
    bgemm(q, T, dAp, dS, nullptr, B, n, K, K, n, K, K, (long)K * n, (long)K * K, (long)n * K, true, false, 1.f, 1.f, 0);
    bgemm(q, S, dAp, V, nullptr, B, n, K, K, K, K, K, (long)n * K, (long)K * K, (long)n * K, false, true, 1.f, 0.f, 0);
    bgemm(q, adj, V, dS, nullptr, B, n, K, n, n, K, K, (long)n * n, (long)n * K, (long)n * K, false, false, 1.f, 1.f, 0);
    if (dadj) {
        bgemm(q, S, dAp, V, nullptr, B, n, K, K, K, K, K, (long)n * K, (long)K * K, (long)n * K, false, false, 1.f, 0.f,
              0);
        bgemm(q, V, S, dadj, nullptr, B, n, n, K, K, K, n, (long)n * K, (long)n * K, (long)n * n, false, true, 1.f, 1.f,
              0);
    }
}

void loss_fwd_seq(Seq& q, const float* ypred, const long long* label, const float* S, const float* adj,
                  const int* num_nodes, const float* norm, float* loss_out, float* prob, float* dunit, int B, int C,
                  int N, int K, int linkpred);
void loss_bwd_seq(Seq& q, const float* prob, const long long* label, const float* S, const float* adj,
                  const int* num_nodes, const float* norm, const float* dloss, float* d_ypred, float* dS, int B, int C,
                  int N, int K, int linkpred) {
    if (d_ypred) ce_bwd(q, prob, label, dloss, 1.f, d_ypred, B, C);
    if (linkpred) linkpred_bwd(q, S, K, adj, num_nodes, dloss, dS, K, B, N, K, 0, norm);
}

__global__ void k_add2(float* out, const float* a, const float* b) { out[0] = a[0] + b[0]; out[1] = b[0]; }

void loss_fwd_seq(Seq& q, const float* ypred, const long long* label, const float* S, const float* adj,
                  const int* num_nodes, const float* norm, float* loss_out, float* prob, float* dunit, int B, int C,
                  int N, int K, int linkpred) {
    float* tmp = q.alloc<float>(64);
    if (q.err) return;
    if (!linkpred) {                       // loss_out = (CE, 0): one launch
        ce_fwd(q, ypred, label, loss_out, prob, B, C, loss_out + 1, dunit);
        return;
    }
    ce_fwd(q, ypred, label, tmp, prob, B, C, nullptr, dunit);
    if (linkpred) {
        linkpred_fwd(q, S, K, adj, num_nodes, tmp + 1, B, N, K, norm);
        if (q.ok()) {
            hipLaunchKernelGGL(k_add2, dim3(1), dim3(1), 0, q.stream, loss_out, tmp, tmp + 1);
            q.check_launch("loss_add");
        }
    }
}

template <typename F>
size_t sized(F&& f) {
    Seq q = Seq::sizing();
    f(q);
    return q.ws_off + 256;
}

}  // namespace
}  // namespace dp

using namespace dp;

#define STREAM(s) ((hipStream_t)(s))
#define NONNEG(v) DP_CHECK_ARG((v) > 0, #v "=%d must be positive", (int)(v))
#define NOTNULL(p) DP_CHECK_ARG((p) != nullptr, #p " is NULL")

extern "C" {

int dp_version(void) { return DP_VERSION; }
const char* dp_last_error_string(void) { return dp::last_error(); }
int dp_device_error(int clear) { return device_error_take(clear != 0); }
const char* dp_device_error_describe(int mask) { return device_error_text(mask); }
#define DEVICE_GATE(name)                                     \
    do {                                                      \
        const int gate_rc_ = device_error_gate(name);         \
        if (gate_rc_ != DP_OK) return gate_rc_;               \
    } while (0)

int dp_bgemm_f32(const float* A, const float* B, float* C, const float* bias, int batch, int M, int N, int K,
                 int lda, int ldb, int ldc, long strideA, long strideB, long strideC, int transA, int transB,
                 float alpha, float beta, int act, void* stream) {
    NOTNULL(A); NOTNULL(B); NOTNULL(C);
    NONNEG(batch); NONNEG(M); NONNEG(N);
    DP_CHECK_ARG(K >= 0, "K=%d must be >= 0", K);
    DP_CHECK_ARG(lda >= (transA ? M : K), "lda=%d too small", lda);
    DP_CHECK_ARG(ldb >= (transB ? K : N), "ldb=%d too small", ldb);
    DP_CHECK_ARG(ldc >= N, "ldc=%d < N=%d", ldc, N);
    Seq q(STREAM(stream), nullptr, 0);
    bgemm(q, A, B, C, bias, batch, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, transA != 0, transB != 0, alpha,
          beta, act);
    return q.err;
}

int dp_adj_aggregate(const float* adj, const float* V, int ldv, float* U, int ldu, int B, int n, int C, int trans,
                     float beta, void* stream) {
    NOTNULL(adj); NOTNULL(V); NOTNULL(U);
    NONNEG(B); NONNEG(n); NONNEG(C);
    DP_CHECK_ARG(ldv >= C && ldu >= C, "ldv=%d/ldu=%d smaller than C=%d", ldv, ldu, C);
    Seq q(STREAM(stream), nullptr, 0);
    aggregate(q, adj, V, ldv, U, ldu, B, n, C, trans != 0, beta);
    return q.err;
}

int dp_adj_pack_ld(int n) { return adj_pack_ld(n); }
size_t dp_adj_pack_bytes(int B, int n) { return (size_t)B * n * adj_pack_ld(n) * sizeof(unsigned short); }
int dp_adj_pack(const float* adj, void* packed, void* packed_t, int* flag, int B, int n, void* stream) {
    NOTNULL(adj); NOTNULL(packed); NOTNULL(packed_t); NOTNULL(flag);
    NONNEG(B); NONNEG(n);
    Seq q(STREAM(stream), nullptr, 0);
    adj_pack(q, adj, (unsigned short*)packed, (unsigned short*)packed_t, flag, B, n, adj_pack_ld(n));
    return q.err;
}
size_t dp_adj_aggregate_packed_workspace_bytes(int B, int n, int C) {
    return split3_elems(B, n, C) * sizeof(unsigned short) + 512;
}
int dp_adj_aggregate_packed(const float* adj, const void* packed, const void* packed_t, const int* flag,
                            const float* V, int ldv, float* U, int ldu, int B, int n, int C, int trans, float beta,
                            int presplit, void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(adj); NOTNULL(packed); NOTNULL(packed_t); NOTNULL(flag); NOTNULL(V); NOTNULL(U);
    NONNEG(B); NONNEG(n); NONNEG(C);
    DP_CHECK_ARG(ldv >= C && ldu >= C, "ldv=%d/ldu=%d smaller than C=%d", ldv, ldu, C);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    unsigned short* vs = q.alloc<unsigned short>(split3_elems(B, n, C));
    if (q.err) return q.err;
    PackedAdj pk{(const unsigned short*)packed, (const unsigned short*)packed_t, adj_pack_ld(n), flag};
    aggregate(q, adj, V, ldv, U, ldu, B, n, C, trans != 0, beta, &pk, vs, presplit != 0);
    return q.err;
}

size_t dp_gcn_layer_workspace_bytes(int B, int n, int Fin, int Fout) {
    size_t f = sized([&](Seq& q) { gcn_layer_fwd_seq(q, 0, Fin, 0, 0, 0, 0, Fout, 0, B, n, Fin, Fout, 0); });
    size_t b = sized([&](Seq& q) {
        gcn_layer_bwd_seq(q, 0, Fin, 0, 0, 0, Fout, 0, 0, Fout, 0, Fin, 0, 0, (float*)1, B, n, Fin, Fout, 0);
    });
    return f > b ? f : b;
}
int dp_gcn_layer_fwd(const float* x, int ldx, const float* adj, const float* W, const float* bias, float* y, int ldy,
                     float* invnorm, int B, int n, int Fin, int Fout, int flags, void* workspace,
                     size_t workspace_bytes, void* stream) {
    NOTNULL(x); NOTNULL(adj); NOTNULL(W); NOTNULL(y);
    NONNEG(B); NONNEG(n); NONNEG(Fin); NONNEG(Fout);
    DP_CHECK_ARG(ldx >= Fin && ldy >= Fout, "ldx=%d/ldy=%d smaller than the row width", ldx, ldy);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    gcn_layer_fwd_seq(q, x, ldx, adj, W, bias, y, ldy, invnorm, B, n, Fin, Fout, flags);
    return q.err;
}
int dp_gcn_layer_bwd(const float* x, int ldx, const float* adj, const float* W, const float* y, int ldy,
                     const float* invnorm, const float* dy, int lddy, float* dx, int lddx, float* dW, float* db,
                     float* dadj, int B, int n, int Fin, int Fout, int flags, void* workspace, size_t workspace_bytes,
                     void* stream) {
    NOTNULL(x); NOTNULL(adj); NOTNULL(W); NOTNULL(y); NOTNULL(dy); NOTNULL(dW);
    NONNEG(B); NONNEG(n); NONNEG(Fin); NONNEG(Fout);
    DP_CHECK_ARG(!(flags & DP_F_NORMALIZE) || invnorm, "invnorm is NULL but DP_F_NORMALIZE is set");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    gcn_layer_bwd_seq(q, x, ldx, adj, W, y, ldy, invnorm, dy, lddy, dx, lddx, dW, db, dadj, B, n, Fin, Fout, flags);
    return q.err;
}

size_t dp_bn_node_workspace_bytes(int B, int n, int F) {
    size_t f = sized([&](Seq& q) { bn_fwd_seq(q, 0, F, 0, F, 0, B, n, F, 1); });
    size_t b = sized([&](Seq& q) { bn_bwd_seq(q, 0, F, 0, F, 0, 0, F, 0, F, B, n, F, 1); });
    return f > b ? f : b;
}
int dp_bn_node_fwd(const float* x, int ldx, float* y, int ldy, float* stats, int B, int n, int F, int relu,
                   void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(x); NOTNULL(y); NOTNULL(stats);
    NONNEG(B); NONNEG(n); NONNEG(F);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    bn_fwd_seq(q, x, ldx, y, ldy, stats, B, n, F, relu);
    return q.err;
}
int dp_bn_node_bwd(const float* x, int ldx, const float* y, int ldy, const float* stats, const float* dy, int lddy,
                   float* dx, int lddx, int B, int n, int F, int relu, void* workspace, size_t workspace_bytes,
                   void* stream) {
    NOTNULL(y); NOTNULL(stats); NOTNULL(dy); NOTNULL(dx);
    DP_CHECK_ARG(!relu || x, "x is NULL but relu != 0");
    NONNEG(B); NONNEG(n); NONNEG(F);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    bn_bwd_seq(q, x, ldx, y, ldy, stats, dy, lddy, dx, lddx, B, n, F, relu);
    return q.err;
}

size_t dp_assign_workspace_bytes(int B, int n, int Din, int K) {
    return sized([&](Seq& q) { assign_fwd_seq(q, 0, Din, 0, 0, 0, 0, B, n, Din, K); });
}
int dp_assign_softmax_mask_fwd(const float* z, int ldz, const float* Wp, const float* bp, const int* num_nodes,
                               float* S, int B, int n, int Din, int K, void* workspace, size_t workspace_bytes,
                               void* stream) {
    NOTNULL(z); NOTNULL(Wp); NOTNULL(S);
    NONNEG(B); NONNEG(n); NONNEG(Din); NONNEG(K);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    assign_fwd_seq(q, z, ldz, Wp, bp, num_nodes, S, B, n, Din, K);
    return q.err;
}
int dp_assign_softmax_mask_bwd(const float* z, int ldz, const float* Wp, const float* S, const float* dS,
                               const int* num_nodes, float* dz, int lddz, float* dWp, float* dbp, int B, int n, int Din,
                               int K, void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(z); NOTNULL(Wp); NOTNULL(S); NOTNULL(dS); NOTNULL(dWp);
    NONNEG(B); NONNEG(n); NONNEG(Din); NONNEG(K);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    assign_bwd_seq(q, z, ldz, Wp, S, dS, num_nodes, dz, lddz, dWp, dbp, B, n, Din, K);
    return q.err;
}

int dp_pool_fwd(const float* S, const float* Z, int ldz, const float* adj, float* Xp, float* Ap, float* T, int B, int n,
                int K, int D, void* stream) {
    NOTNULL(S); NOTNULL(Z); NOTNULL(adj); NOTNULL(Xp); NOTNULL(Ap); NOTNULL(T);
    NONNEG(B); NONNEG(n); NONNEG(K); NONNEG(D);
    Seq q(STREAM(stream), nullptr, 0);
    pool_fwd_seq(q, S, Z, ldz, adj, Xp, Ap, T, B, n, K, D);
    return q.err;
}
size_t dp_pool_bwd_workspace_bytes(int B, int n, int K, int D) {
    return sized([&](Seq& q) { pool_bwd_seq(q, 0, 0, D, 0, 0, 0, 0, 0, 0, D, 0, B, n, K, D); });
}
int dp_pool_bwd(const float* S, const float* Z, int ldz, const float* adj, const float* T, const float* dXp,
                const float* dAp, float* dS, float* dZ, int lddz, float* dadj, int B, int n, int K, int D,
                void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(S); NOTNULL(Z); NOTNULL(adj); NOTNULL(T); NOTNULL(dXp); NOTNULL(dAp); NOTNULL(dS); NOTNULL(dZ);
    NONNEG(B); NONNEG(n); NONNEG(K); NONNEG(D);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    pool_bwd_seq(q, S, Z, ldz, adj, T, dXp, dAp, dS, dZ, lddz, dadj, B, n, K, D);
    return q.err;
}

int dp_masked_max_fwd(const float* Z, int ldz, const int* num_nodes, float* out, int ldo, int* argmax, int B, int n,
                      int F, void* stream) {
    NOTNULL(Z); NOTNULL(out); NOTNULL(argmax);
    NONNEG(B); NONNEG(n); NONNEG(F);
    Seq q(STREAM(stream), nullptr, 0);
    masked_max_fwd(q, Z, ldz, num_nodes, out, ldo, argmax, F, B, n, F);
    return q.err;
}
int dp_masked_max_bwd(const float* dout, int ldo, const int* argmax, float* dZ, int lddz, int B, int n, int F,
                      void* stream) {
    NOTNULL(dout); NOTNULL(argmax); NOTNULL(dZ);
    NONNEG(B); NONNEG(n); NONNEG(F);
    Seq q(STREAM(stream), nullptr, 0);
    masked_max_bwd(q, dout, ldo, argmax, F, dZ, lddz, B, n, F);
    return q.err;
}

size_t dp_linkpred_workspace_bytes(int B, int n, int K) {
    size_t f = sized([&](Seq& q) { linkpred_fwd(q, 0, K, 0, 0, 0, B, n, K); });
    size_t b = sized([&](Seq& q) { linkpred_bwd(q, 0, K, 0, 0, 0, 0, K, B, n, K, 0); });
    return f > b ? f : b;
}
int dp_linkpred_loss_fwd(const float* S, const float* adj, const int* num_nodes, float* loss_out, int B, int n, int K,
                         void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(S); NOTNULL(adj); NOTNULL(loss_out);
    NONNEG(B); NONNEG(n); NONNEG(K);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    linkpred_fwd(q, S, K, adj, num_nodes, loss_out, B, n, K);
    return q.err;
}
int dp_linkpred_loss_bwd(const float* S, const float* adj, const int* num_nodes, const float* dloss, float* dS,
                         int accumulate, int B, int n, int K, void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(S); NOTNULL(adj); NOTNULL(dS);
    NONNEG(B); NONNEG(n); NONNEG(K);
    Seq q(STREAM(stream), workspace, workspace_bytes);
    linkpred_bwd(q, S, K, adj, num_nodes, dloss, dS, K, B, n, K, accumulate);
    return q.err;
}

int dp_cross_entropy_fwd(const float* logits, const long long* label, float* loss_out, float* prob, int B, int C,
                         void* stream) {
    NOTNULL(logits); NOTNULL(label); NOTNULL(loss_out);
    NONNEG(B); NONNEG(C);
    Seq q(STREAM(stream), nullptr, 0);
    ce_fwd(q, logits, label, loss_out, prob, B, C);
    return q.err;
}
int dp_cross_entropy_bwd(const float* prob, const long long* label, const float* dloss, float* dlogits, int B, int C,
                         void* stream) {
    NOTNULL(prob); NOTNULL(label); NOTNULL(dlogits);
    NONNEG(B); NONNEG(C);
    Seq q(STREAM(stream), nullptr, 0);
    ce_bwd(q, prob, label, dloss, 1.f, dlogits, B, C);
    return q.err;
}

size_t dp_set2set_save_bytes(int B, int n, int d) { return set2set_save_bytes(B, n, d); }
int dp_set2set_fwd(const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                   const float* b_hh, const float* Wp, const float* bp, float* out, int B, int n, int d, void* save,
                   size_t save_bytes, void* stream) {
    NOTNULL(emb); NOTNULL(w_ih); NOTNULL(w_hh); NOTNULL(b_ih); NOTNULL(b_hh); NOTNULL(Wp); NOTNULL(bp); NOTNULL(out);
    NOTNULL(save);
    NONNEG(B); NONNEG(n); NONNEG(d);
    DP_CHECK_ARG(save_bytes >= set2set_save_bytes(B, n, d), "set2set save buffer too small: %zu < %zu",
                 save_bytes, set2set_save_bytes(B, n, d));
    Seq q(STREAM(stream), nullptr, 0);
    set2set_fwd(q, emb, lde, w_ih, w_hh, b_ih, b_hh, Wp, bp, out, B, n, d, save);
    return q.err;
}
size_t dp_set2set_bwd_workspace_bytes(int B, int n, int d) {
    return sized([&](Seq& q) {
        set2set_bwd(q, 0, d, 0, 0, 0, 0, 0, 0, 0, 0, 0, d, 0, 0, 0, 0, 0, 0, B, n, d, 0);
    });
}
int dp_set2set_bwd(const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                   const float* b_hh, const float* Wp, const float* bp, const float* out, const float* dout,
                   float* demb, int ldde, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dWp,
                   float* dbp, int B, int n, int d, const void* save, size_t save_bytes, void* workspace,
                   size_t workspace_bytes, void* stream) {
    NOTNULL(emb); NOTNULL(w_ih); NOTNULL(w_hh); NOTNULL(Wp); NOTNULL(out); NOTNULL(dout); NOTNULL(demb);
    NOTNULL(dw_ih); NOTNULL(dw_hh); NOTNULL(db_ih); NOTNULL(db_hh); NOTNULL(dWp); NOTNULL(dbp); NOTNULL(save);
    NONNEG(B); NONNEG(n); NONNEG(d);
    DP_CHECK_ARG(save_bytes >= set2set_save_bytes(B, n, d), "set2set save buffer too small");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    set2set_bwd(q, emb, lde, w_ih, w_hh, b_ih, b_hh, Wp, bp, out, dout, demb, ldde, dw_ih, dw_hh, db_ih, db_hh, dWp,
                dbp, B, n, d, save);
    return q.err;
}

int dp_mean_aggregate_fwd(const float* table, int ldt, const int* indptr, const int* indices, float* out, int ldo,
                          int n_rows, int feat, void* stream) {
    NOTNULL(table); NOTNULL(indptr); NOTNULL(indices); NOTNULL(out);
    NONNEG(n_rows); NONNEG(feat);
    Seq q(STREAM(stream), nullptr, 0);
    mean_aggregate_fwd(q, table, ldt, indptr, indices, out, ldo, n_rows, feat);
    return q.err;
}
int dp_mean_aggregate_bwd(const float* dout, int ldo, const int* indptr, const int* indices, float* dtable, int ldt,
                          int n_rows, int feat, void* stream) {
    NOTNULL(dout); NOTNULL(indptr); NOTNULL(indices); NOTNULL(dtable);
    NONNEG(n_rows); NONNEG(feat);
    Seq q(STREAM(stream), nullptr, 0);
    mean_aggregate_bwd(q, dout, ldo, indptr, indices, dtable, ldt, n_rows, feat);
    return q.err;
}

int dp_bgemm_split_bf16(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb,
                        int ldc, long strideA, long strideB, long strideC, int transA, int transB, float beta,
                        void* stream) {
    NOTNULL(A); NOTNULL(B); NOTNULL(C);
    NONNEG(batch); NONNEG(M); NONNEG(N); NONNEG(K);
    DP_CHECK_ARG(beta == 0.f || beta == 1.f, "beta must be 0 or 1");
    Seq q(STREAM(stream), nullptr, 0);
    GemmDesc d{A, B, C, nullptr, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, transA != 0, transB != 0, 1.f, beta,
               0, 0, 0, nullptr, 0, 0, 0};
    gemm_split_bf16(q, d, batch);
    return q.err;
}

// ------------------------------------------------------------------ N4: CSR GraphConv
int dp_csr_aggregate(const float* table, int ldt, const int* indptr, const int* indices, float* out, int ldo, int n_rows,
                     int feat, int mean, float beta, void* stream) {
    NOTNULL(table); NOTNULL(indptr); NOTNULL(indices); NOTNULL(out);
    NONNEG(n_rows); NONNEG(feat);
    Seq q(STREAM(stream), nullptr, 0);
    csr_aggregate_fwd(q, table, ldt, indptr, indices, out, ldo, n_rows, feat, mean, beta);
    return q.err;
}
namespace {
void sparse_gcn_fwd_seq(Seq& q, const float* x, int ldx, const int* indptr, const int* indices, const float* W,
                        const float* bias, float* y, int ldy, float* ax, float* invn, int n, int Fin, int Fout,
                        int flags) {
    float* U = q.alloc<float>((size_t)n * Fout);
    if (q.err) return;
    csr_aggregate_fwd(q, x, ldx, indptr, indices, ax, Fin, n, Fin, 0, 0.f);
    if (flags & DP_F_ADD_SELF) axpy(q, ax, x, 1.f, (long)n * Fin);          // (x is contiguous: checked at the ABI)
    bgemm(q, ax, W, U, nullptr, 1, n, Fout, Fin, Fin, Fout, Fout, 0, 0, 0, false, false, 1.f, 0.f, 0);
    rownorm_fwd(q, U, Fout, nullptr, gcp(bias, 0), one_group(Fout), gp(y, ldy), invn, nullptr, n,
                (flags & DP_F_NORMALIZE) ? 1 : 0, 0);
}
void sparse_gcn_bwd_seq(Seq& q, const float* ax, const int* indptr, const int* indices, const int* indptr_t,
                        const int* indices_t, const float* W, const float* y, int ldy, const float* invn,
                        const float* dy, int lddy, float* dx, int lddx, float* dW, float* db, int n, int Fin, int Fout,
                        int flags) {
    float* dU = q.alloc<float>((size_t)n * Fout);
    float* dax = dx ? q.alloc<float>((size_t)n * Fin) : nullptr;
    if (q.err) return;
    rownorm_bwd(q, gcp(dy, lddy), gcp(nullptr, 0), gcp(y, ldy), invn, nullptr, nullptr, one_group(Fout), dU, Fout,
                nullptr, 1, n, 0, 0, (flags & DP_F_NORMALIZE) ? 1 : 0);
    if (db) colsum_batched(q, dU, Fout, 0, n, Fout, db, 0, 1);
    bgemm(q, ax, dU, dW, nullptr, 1, Fin, Fout, n, Fin, Fout, Fout, 0, 0, 0, true, false, 1.f, 0.f, 0);
    if (!dx) return;
    bgemm(q, dU, W, dax, nullptr, 1, n, Fin, Fout, Fout, Fout, Fin, 0, 0, 0, false, true, 1.f, 0.f, 0);
    if (indptr_t && indices_t) {
        csr_aggregate_fwd(q, dax, Fin, indptr_t, indices_t, dx, lddx, n, Fin, 0, 0.f);      // dx = A^T dax, a gather
    } else {
        zero_fill(q, dx, (size_t)n * Fin * sizeof(float));
        csr_aggregate_bwd_scatter(q, dax, Fin, indptr, indices, dx, lddx, n, Fin, 0);
    }
    if (flags & DP_F_ADD_SELF) axpy(q, dx, dax, 1.f, (long)n * Fin);
}
}  // namespace
size_t dp_sparse_gcn_layer_workspace_bytes(int n, int Fin, int Fout) {
    size_t f = sized([&](Seq& q) { sparse_gcn_fwd_seq(q, 0, Fin, 0, 0, 0, 0, 0, Fout, 0, 0, n, Fin, Fout, 0); });
    size_t b = sized([&](Seq& q) {
        sparse_gcn_bwd_seq(q, 0, 0, 0, 0, 0, 0, 0, Fout, 0, 0, Fout, (float*)1, Fin, 0, 0, n, Fin, Fout, 0);
    });
    return f > b ? f : b;
}
int dp_sparse_gcn_layer_fwd(const float* x, int ldx, const int* indptr, const int* indices, const float* W,
                            const float* bias, float* y, int ldy, float* ax, float* invnorm, int n, int Fin, int Fout,
                            int flags, void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(x); NOTNULL(indptr); NOTNULL(indices); NOTNULL(W); NOTNULL(y); NOTNULL(ax); NOTNULL(invnorm);
    NONNEG(n); NONNEG(Fin); NONNEG(Fout);
    DP_CHECK_ARG(!(flags & DP_F_ADD_SELF) || ldx == Fin, "add_self needs a contiguous x (ldx == Fin)");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    sparse_gcn_fwd_seq(q, x, ldx, indptr, indices, W, bias, y, ldy, ax, invnorm, n, Fin, Fout, flags);
    return q.err;
}
int dp_sparse_gcn_layer_bwd(const float* ax, const int* indptr, const int* indices, const int* indptr_t,
                            const int* indices_t, const float* W, const float* y, int ldy, const float* invnorm,
                            const float* dy, int lddy, float* dx, int lddx, float* dW, float* db, int n, int Fin,
                            int Fout, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(ax); NOTNULL(indptr); NOTNULL(indices); NOTNULL(W); NOTNULL(y); NOTNULL(invnorm); NOTNULL(dy); NOTNULL(dW);
    NONNEG(n); NONNEG(Fin); NONNEG(Fout);
    DP_CHECK_ARG((indptr_t == nullptr) == (indices_t == nullptr), "indptr_t and indices_t come together");
    DP_CHECK_ARG(!dx || lddx == Fin, "dx must be contiguous (lddx == Fin)");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    sparse_gcn_bwd_seq(q, ax, indptr, indices, indptr_t, indices_t, W, y, ldy, invnorm, dy, lddy, dx, lddx, dW, db, n,
                       Fin, Fout, flags);
    return q.err;
}

// ------------------------------------------------------------------ model level
size_t dp_sizeof_encoder_cfg(void) { return sizeof(dp_encoder_cfg); }
size_t dp_encoder_save_bytes(const dp_encoder_cfg* cfg) {
    if (encoder_validate(cfg) != DP_OK) return 0;
    return encoder_save_bytes(*cfg) + 256;
}
size_t dp_encoder_workspace_bytes(const dp_encoder_cfg* cfg) {
    if (encoder_validate(cfg) != DP_OK) return 0;
    size_t f = sized([&](Seq& q) { encoder_forward(q, *cfg, 0, 0, 0, 0, 0, 0, 0, 0, 0, DP_MODE_TRAIN, 0); });
    size_t b = sized([&](Seq& q) { encoder_backward(q, *cfg, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); });
    return f > b ? f : b;
}
int dp_encoder_save_locate(const dp_encoder_cfg* cfg, int level, int field, size_t* offset, size_t* count) {
    int rc = encoder_validate(cfg);
    if (rc != DP_OK) return rc;
    NOTNULL(offset); NOTNULL(count);
    DP_CHECK_ARG(level >= 0 && level <= cfg->num_pooling, "level=%d out of range [0,%d]", level, cfg->num_pooling);
    return encoder_save_locate(*cfg, level, field, offset, count);
}
int dp_encoder_forward(const dp_encoder_cfg* cfg, const float* params, const float* x, const float* adj,
                       const float* assign_x, const int* num_nodes, const float* dropout, float* ypred,
                       float* assign_out, long long* labels_out, void* save, size_t save_bytes, void* workspace,
                       size_t workspace_bytes, int mode, void* stream) {
    int rc = encoder_validate(cfg);
    if (rc != DP_OK) return rc;
    NOTNULL(params); NOTNULL(x); NOTNULL(adj); NOTNULL(ypred); NOTNULL(save);
    DP_CHECK_ARG(cfg->num_pooling == 0 || assign_x, "assign_x is NULL");
    DP_CHECK_ARG(save_bytes >= encoder_save_bytes(*cfg), "save buffer too small: %zu < %zu", save_bytes,
                 encoder_save_bytes(*cfg));
    DEVICE_GATE("dp_encoder_forward");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    DP_CHECK_ARG((mode & ~(DP_MODE_TRAIN)) == 0, "mode=%d: DP_MODE_EVAL or DP_MODE_TRAIN", mode);
    return encoder_forward(q, *cfg, params, x, adj, assign_x, num_nodes, dropout, ypred, assign_out, save, mode,
                           labels_out);
}
int dp_encoder_backward(const dp_encoder_cfg* cfg, const float* params, const float* x, const float* adj,
                        const float* assign_x, const int* num_nodes, const float* dropout, const float* d_ypred,
                        const float* d_assign, float* grads, const void* save, size_t save_bytes, void* workspace,
                        size_t workspace_bytes, int prezeroed, void* stream) {
    int rc = encoder_validate(cfg);
    if (rc != DP_OK) return rc;
    NOTNULL(params); NOTNULL(x); NOTNULL(adj); NOTNULL(d_ypred); NOTNULL(grads); NOTNULL(save);
    DP_CHECK_ARG(save_bytes >= encoder_save_bytes(*cfg), "save buffer too small: %zu < %zu", save_bytes,
                 encoder_save_bytes(*cfg));
    DEVICE_GATE("dp_encoder_backward");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    return encoder_backward(q, *cfg, params, x, adj, assign_x, num_nodes, dropout, d_ypred, d_assign, grads, save,
                            prezeroed);
}

int dp_encoder_forward_packed(const dp_encoder_cfg* cfg, const float* params, const float* x, const void* adj_pk,
                              const void* adj_pkt,
                       const float* assign_x, const int* num_nodes, const float* dropout, float* ypred,
                       float* assign_out, long long* labels_out, void* save, size_t save_bytes, void* workspace,
                       size_t workspace_bytes, int mode, void* stream) {
    int rc = encoder_validate(cfg);
    if (rc != DP_OK) return rc;
    NOTNULL(params); NOTNULL(x); NOTNULL(adj_pk); NOTNULL(adj_pkt); NOTNULL(ypred); NOTNULL(save);
    DP_CHECK_ARG(cfg->num_pooling == 0 || assign_x, "assign_x is NULL");
    DP_CHECK_ARG(save_bytes >= encoder_save_bytes(*cfg), "save buffer too small: %zu < %zu", save_bytes,
                 encoder_save_bytes(*cfg));
    DEVICE_GATE("dp_encoder_forward_packed");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    DP_CHECK_ARG((mode & ~(DP_MODE_TRAIN)) == 0, "mode=%d: DP_MODE_EVAL or DP_MODE_TRAIN", mode);
    const PackedAdj given{static_cast<const unsigned short*>(adj_pk), static_cast<const unsigned short*>(adj_pkt),
                          adj_pack_ld(cfg->N), nullptr};
    return encoder_forward(q, *cfg, params, x, nullptr, assign_x, num_nodes, dropout, ypred, assign_out, save, mode,
                           labels_out, &given);
}

int dp_encoder_backward_packed(const dp_encoder_cfg* cfg, const float* params, const float* x, const void* adj_pk,
                               const void* adj_pkt,
                        const float* assign_x, const int* num_nodes, const float* dropout, const float* d_ypred,
                        const float* d_assign, float* grads, const void* save, size_t save_bytes, void* workspace,
                        size_t workspace_bytes, int prezeroed, void* stream) {
    int rc = encoder_validate(cfg);
    if (rc != DP_OK) return rc;
    NOTNULL(params); NOTNULL(x); NOTNULL(adj_pk); NOTNULL(adj_pkt); NOTNULL(d_ypred); NOTNULL(grads); NOTNULL(save);
    DP_CHECK_ARG(save_bytes >= encoder_save_bytes(*cfg), "save buffer too small: %zu < %zu", save_bytes,
                 encoder_save_bytes(*cfg));
    DEVICE_GATE("dp_encoder_backward_packed");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    const PackedAdj given{static_cast<const unsigned short*>(adj_pk), static_cast<const unsigned short*>(adj_pkt),
                          adj_pack_ld(cfg->N), nullptr};
    return encoder_backward(q, *cfg, params, x, nullptr, assign_x, num_nodes, dropout, d_ypred, d_assign, grads, save,
                            prezeroed, &given);
}

size_t dp_loss_workspace_bytes(int B, int N, int K, int linkpred) {
    size_t f = sized([&](Seq& q) { loss_fwd_seq(q, 0, 0, 0, 0, 0, 0, 0, 0, 0, B, 1, N, K, linkpred); });
    size_t b = sized([&](Seq& q) { loss_bwd_seq(q, 0, 0, 0, 0, 0, 0, 0, 0, 0, B, 1, N, K, linkpred); });
    return f > b ? f : b;
}
int dp_loss_forward(const float* ypred, const long long* label, const float* S, const float* adj, const int* num_nodes,
                    const float* link_norm, float* loss_out, float* prob, float* d_ypred_unit, int B, int C, int N, int K,
                    int linkpred, void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(ypred); NOTNULL(label); NOTNULL(loss_out); NOTNULL(prob);
    NONNEG(B); NONNEG(C);
    DP_CHECK_ARG(!linkpred || (S && adj && N > 0 && K > 0), "linkpred needs S, adj, N, K");
    DEVICE_GATE("dp_loss_forward");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    loss_fwd_seq(q, ypred, label, S, adj, num_nodes, link_norm, loss_out, prob, d_ypred_unit, B, C, N, K, linkpred);
    return q.err;
}
int dp_loss_backward(const float* prob, const long long* label, const float* S, const float* adj, const int* num_nodes,
                     const float* link_norm, const float* dloss, float* d_ypred, float* dS, int B, int C, int N, int K,
                     int linkpred, void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(prob); NOTNULL(label);
    NONNEG(B); NONNEG(C);
    DP_CHECK_ARG(!linkpred || (S && adj && dS && N > 0 && K > 0), "linkpred needs S, adj, dS, N, K");
    DEVICE_GATE("dp_loss_backward");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    loss_bwd_seq(q, prob, label, S, adj, num_nodes, link_norm, dloss, d_ypred, dS, B, C, N, K, linkpred);
    return q.err;
}

int dp_build_batch(const int* edge_src, const int* edge_dst, const int* edge_ptr, const int* node_label,
                   const int* node_ptr, float* adj, float* feats, float* assign_feats, int* num_nodes, int* errors,
                   int* degree, int B, int N, int F, int feature_mode, int symmetric, int max_edges_per_graph,
                   void* stream) {
    NOTNULL(edge_src); NOTNULL(edge_dst); NOTNULL(edge_ptr); NOTNULL(node_ptr); NOTNULL(adj); NOTNULL(num_nodes);
    NOTNULL(errors);
    NONNEG(B); NONNEG(N);
    DP_CHECK_ARG(feature_mode >= 0 && feature_mode <= 3, "feature_mode=%d (0 default, 1 id, 2 deg-num, 3 deg)",
                 feature_mode);
    const bool labels = feature_mode == 0 || feature_mode == 3;
    DP_CHECK_ARG(!(feats || assign_feats) || !labels || (node_label && F > 0),
                 "label-based features need node labels and F > 0");
    DP_CHECK_ARG(feature_mode < 2 || degree, "deg / deg-num features need the degree workspace (B*N ints)");
    Seq q(STREAM(stream), nullptr, 0);
    build_batch(q, edge_src, edge_dst, edge_ptr, node_label, node_ptr, adj, nullptr, nullptr, feats, assign_feats,
                num_nodes, errors, degree, B, N, F, feature_mode, symmetric, max_edges_per_graph);
    return q.err;
}

int dp_build_batch_packed(const int* edge_src, const int* edge_dst, const int* edge_ptr, const int* node_label,
                          const int* node_ptr, void* adj_pk, void* adj_pkt, float* feats, float* assign_feats,
                          int* num_nodes, int* errors, int* degree, int B, int N, int F, int feature_mode, int symmetric,
                          int max_edges_per_graph, void* stream) {
    NOTNULL(edge_src); NOTNULL(edge_dst); NOTNULL(edge_ptr); NOTNULL(node_ptr); NOTNULL(adj_pk); NOTNULL(adj_pkt);
    NOTNULL(num_nodes); NOTNULL(errors);
    NONNEG(B); NONNEG(N);
    DP_CHECK_ARG(feature_mode >= 0 && feature_mode <= 3, "feature_mode=%d (0 default, 1 id, 2 deg-num, 3 deg)",
                 feature_mode);
    DP_CHECK_ARG(adj_pk != adj_pkt || symmetric, "A and A^T may share one buffer only for a symmetric edge list");
    const bool labels = feature_mode == 0 || feature_mode == 3;
    DP_CHECK_ARG(!(feats || assign_feats) || !labels || (node_label && F > 0),
                 "label-based features need node labels and F > 0");
    DP_CHECK_ARG(feature_mode < 2 || degree, "deg / deg-num features need the degree workspace (B*N ints)");
    Seq q(STREAM(stream), nullptr, 0);
    build_batch(q, edge_src, edge_dst, edge_ptr, node_label, node_ptr, nullptr, static_cast<unsigned short*>(adj_pk),
                static_cast<unsigned short*>(adj_pkt), feats, assign_feats, num_nodes, errors, degree, B, N, F,
                feature_mode, symmetric, max_edges_per_graph);
    return q.err;
}

int dp_gather_labels(const long long* graph_label, long long* label_out, int B, void* stream) {
    NOTNULL(graph_label); NOTNULL(label_out);
    NONNEG(B);
    Seq q(STREAM(stream), nullptr, 0);
    gather_labels(q, graph_label, label_out, B);
    return q.err;
}

size_t dp_clip_adam_workspace_bytes(void) { return 4096; }
int dp_clip_adam_step_counted(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, int* step_counter,
                              float lr, float beta1, float beta2, float eps, float max_norm, float* total_norm_out,
                              void* workspace, size_t workspace_bytes, void* stream) {
    NOTNULL(params); NOTNULL(grads); NOTNULL(exp_avg); NOTNULL(exp_avg_sq); NOTNULL(step_counter);
    DP_CHECK_ARG(n >= 0, "n=%ld must be >= 0", n);
    DP_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "betas (%g, %g) must lie in [0, 1)",
                 (double)beta1, (double)beta2);
    DEVICE_GATE("dp_clip_adam_step_counted");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    clip_adam_step(q, params, grads, exp_avg, exp_avg_sq, n, max_norm, beta1, beta2, eps, 0.f, 0.f, total_norm_out,
                   step_counter, lr);
    return q.err;
}
int dp_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, int step, float lr,
                      float beta1, float beta2, float eps, float max_norm, float* total_norm_out, void* workspace,
                      size_t workspace_bytes, void* stream) {
    NOTNULL(params); NOTNULL(grads); NOTNULL(exp_avg); NOTNULL(exp_avg_sq);
    DP_CHECK_ARG(n >= 0, "n=%ld must be >= 0", n);
    DP_CHECK_ARG(step >= 1, "step=%d is the 1-based count of this update", step);
    DP_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "betas (%g, %g) must lie in [0, 1)",
                 (double)beta1, (double)beta2);
    DEVICE_GATE("dp_clip_adam_step");
    Seq q(STREAM(stream), workspace, workspace_bytes);
    // bias corrections in double on the host, exactly as torch.optim.Adam's single-tensor path computes them
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    clip_adam_step(q, params, grads, exp_avg, exp_avg_sq, n, max_norm, beta1, beta2, eps, (float)((double)lr / bc1),
                   (float)(1.0 / std::sqrt(bc2)), total_norm_out);
    return q.err;
}

}  // extern "C"
