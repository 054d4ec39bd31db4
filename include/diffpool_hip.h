/* diffpool_hip.h — C ABI of libdiffpool_hip.so (MI355X / gfx950).
 *
 * The reference (JiaxuanYou/graph-pooling) has no FFI of its own: its DiffPool path is stock
 * torch calls inside encoders.py / set2set.py / aggregators.py.  Each entry point below replaces
 * the torch call sites named in its comment (file:line relative to the reference root); the
 * Python modules in graph_pooling_amd/ bind them with ctypes.
 *
 * Conventions
 *   - every tensor is fp32, row-major; `ld*` is a row stride in ELEMENTS; batch-major [B, n, ·]
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every
 *     buffer including the workspace; the library never allocates, frees or retains memory
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises the
 *     device, so every entry point can be captured into a hipGraph
 *   - return value: 0 = ok; < 0 = argument / workspace error detected before any launch
 *     (DP_ERR_*); > 0 = the hipError_t of a failed launch.  dp_last_error_string() describes the
 *     last failure of the calling thread.  No C++ exception crosses this boundary.
 *   - `num_nodes` is int32[B] on the device (it replaces the host-built mask of
 *     construct_mask, encoders.py:1035-1046); NULL means "no masking"
 */
#ifndef DIFFPOOL_HIP_H
#define DIFFPOOL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden */
#endif

#define DP_VERSION 100 /* 0.1.0 */

#define DP_OK 0
#define DP_ERR_INVALID_ARG (-1)
#define DP_ERR_WORKSPACE (-2)
#define DP_ERR_UNSUPPORTED (-3)
#define DP_ERR_DEVICE (-4)      /* a kernel enqueued by an EARLIER call on this device reported a failure */

/* layer flags */
#define DP_F_ADD_SELF 1   /* y += x before the weight (GraphConv add_self, encoders.py:966-967) */
#define DP_F_NORMALIZE 2  /* F.normalize(p=2, dim=2), encoders.py:971-972 */
#define DP_F_RELU 4
#define DP_F_BN 8         /* apply_bn after ReLU, encoders.py:1062-1064 */
#define DP_F_LAST_ONLY 16 /* readout of the last layer only (concat=False, encoders.py:1118-1119) */

#define DP_MAX_LAYERS 8
#define DP_MAX_LEVELS 4   /* pooling levels */
#define DP_MAX_PRED 4     /* hidden layers of pred_model */

int dp_version(void);
const char* dp_last_error_string(void);

/* Device-side failures.  Calls only ENQUEUE work, so a failure inside a kernel cannot be the return value of the call
 * that launched it.  Such a kernel raises a bit in a per-device word that lives in pinned, device-mapped HOST memory
 * (64 bytes per device, allocated on first use; the one allocation this library makes — no device memory, no sync):
 *   DP_DEVERR_BARRIER         a whole-level kernel's grid barrier gave up (its workgroups were not co-resident: the
 *                             device is shared with another process / stream).  The kernel also poisons its BatchNorm
 *                             statistics with NaN, so the step's outputs and gradients are NaN, never plausible.
 *   DP_DEVERR_NONFINITE_GRAD  dp_clip_adam_step met a non-finite gradient norm and SKIPPED the update (parameters and
 *                             moments untouched) — which is also what keeps a poisoned backward from ruining them.
 * The NEXT model-level entry on that device (dp_encoder_forward / _backward, dp_loss_forward / _backward,
 * dp_clip_adam_step) finds the word set, clears it, returns DP_ERR_DEVICE before launching anything and describes it
 * in dp_last_error_string().  dp_device_error(clear) reads the current device's word directly (after a stream or
 * device synchronisation it is up to date); dp_device_error_describe(mask) is the text for a mask.  Escape hatch for
 * shared devices: DP_NO_LEVEL_FUSION=1 (no kernel with a grid barrier is used). */
#define DP_DEVERR_BARRIER 1
#define DP_DEVERR_NONFINITE_GRAD 2
int dp_device_error(int clear);
const char* dp_device_error_describe(int mask);

/* Launch timing of the two persistent level-0 kernels (k_level0_fwd / k_level0_bwd, dp_level0.hip) for bench.py's
 * roofline object: dp_profile_level0(1) makes every EAGER launch of them (a capturing stream is left alone) record a
 * pair of HIP events on the launch stream; dp_profile_level0_read(which, &us, &n) waits for the pairs recorded so far
 * and returns their summed elapsed time in microseconds and their number (which: 0 forward, 1 backward);
 * dp_profile_level0(0) switches it off and drops the pairs.  No reference counterpart (measurement only). */
int dp_profile_level0(int enable);
int dp_profile_level0_read(int which, double* total_us, int* launches);

/* ------------------------------------------------------------------ generic contraction
 * C[b] = act(alpha * op(A[b]) op(B[b]) + beta * C[b] + bias), fp32 MFMA (exact f32).
 * Replaces torch.matmul / @ at encoders.py:965,968,1278,1279,1311.  act: 0 none, 1 relu. */
int dp_bgemm_f32(const float* A, const float* B, float* C, const float* bias, int batch, int M, int N,
                 int K, int lda, int ldb, int ldc, long strideA, long strideB, long strideC, int transA,
                 int transB, float alpha, float beta, int act, void* stream);

/* The same contraction (alpha = 1, beta in {0, 1}, no bias / activation) with BOTH fp32 operands split exactly into
 * three bf16 planes on the way to LDS and multiplied on the bf16 matrix cores — six plane products, fp32 accumulation;
 * the dropped cross terms are <= 2^-23 of a product, so the result is fp32-grade (not bit-identical to dp_bgemm_f32).
 * dp_bgemm_f32 and the encoder plans take this kernel by themselves for large shapes (M, N >= 96, K >= 64, >= 256
 * output tiles of 128 x 128); this entry runs it on any shape with every extent >= 4 (smaller ones: dp_bgemm_f32). */
int dp_bgemm_split_bf16(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb,
                        int ldc, long strideA, long strideB, long strideC, int transA, int transB, float beta,
                        void* stream);

/* ------------------------------------------------------------------ adjacency aggregation
 * U[b] = op(adj[b]) · V[b] (+ beta U[b]):  adj [B,n,n], V [B,n,C] (ldv), U [B,n,C] (ldu); trans != 0 uses
 * adj^T.  The HBM-bound pass over the padded dense adjacency — torch.matmul(adj, x), encoders.py:965,
 * and its transpose in backward — as the LDS-panel kernel (any shape; shapes the panel kernel does not
 * take run on dp_bgemm_f32's kernel). */
int dp_adj_aggregate(const float* adj, const float* V, int ldv, float* U, int ldu, int B, int n, int C,
                     int trans, float beta, void* stream);

/* Packed adjacency (the reference multiplies the same fp32 adjacency 12 times per step, encoders.py:965,1279,
 * 1311).  dp_adj_pack makes ONE pass over adj [B,n,n] and writes bf16 copies of A and A^T (rows padded to
 * dp_adj_pack_ld(n) elements) plus a device flag that is 0 iff every entry is exactly representable in bf16
 * (always true for the 0/1 adjacency of graph_sampler.py:26).  dp_adj_aggregate_packed then computes the same
 * U = op(adj)·V with V split exactly into three bf16 planes (fp32-grade result, bf16 MFMA rate, half the
 * adjacency bytes) when the flag is 0, and with the fp32 loop otherwise — decided on the device, no sync.
 * `flag` points at a 256-byte, 16-byte-aligned device block: dp_adj_pack clears all of it and sets only word 0. */
int dp_adj_pack_ld(int n);
size_t dp_adj_pack_bytes(int B, int n);            /* bytes of ONE packed copy */
int dp_adj_pack(const float* adj, void* packed, void* packed_t, int* flag, int B, int n, void* stream);
size_t dp_adj_aggregate_packed_workspace_bytes(int B, int n, int C);
/* presplit != 0: the workspace already holds the 3-plane split of this V from an earlier call (skips the split
 * pass — the encoder plan gets the split from V's producer kernel the same way). */
int dp_adj_aggregate_packed(const float* adj, const void* packed, const void* packed_t, const int* flag,
                            const float* V, int ldv, float* U, int ldu, int B, int n, int C, int trans, float beta,
                            int presplit, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ A1  GraphConv
 * y = l2norm((adj @ x [+ x]) @ W + b)   — GraphConv.forward, encoders.py:962-974.
 * x [B,n,Fin] (ldx), adj [B,n,n], W [Fin,Fout], bias [Fout] or NULL, y [B,n,Fout] (ldy),
 * invnorm [B,n] (saved for backward, may be NULL).  flags: DP_F_ADD_SELF | DP_F_NORMALIZE.
 * Workspace: dp_gcn_layer_workspace_bytes(). */
size_t dp_gcn_layer_workspace_bytes(int B, int n, int Fin, int Fout);
int dp_gcn_layer_fwd(const float* x, int ldx, const float* adj, const float* W, const float* bias,
                     float* y, int ldy, float* invnorm, int B, int n, int Fin, int Fout, int flags,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Backward of the above. dx / dadj may be NULL (not needed). dW [Fin,Fout] and db [Fout] are
 * OVERWRITTEN. */
int dp_gcn_layer_bwd(const float* x, int ldx, const float* adj, const float* W, const float* y, int ldy,
                     const float* invnorm, const float* dy, int lddy, float* dx, int lddx, float* dW,
                     float* db, float* dadj, int B, int n, int Fin, int Fout, int flags, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ A3  apply_bn
 * Batch-norm over the node index with batch statistics (a fresh BatchNorm1d(n) per call,
 * encoders.py:1048-1052): per node n, mean / biased variance over (batch, feature), eps 1e-5.
 * stats [n,2] receives (mean, rstd).  relu != 0 applies ReLU first (encoders.py:1062). */
size_t dp_bn_node_workspace_bytes(int B, int n, int F);
int dp_bn_node_fwd(const float* x, int ldx, float* y, int ldy, float* stats, int B, int n, int F,
                   int relu, void* workspace, size_t workspace_bytes, void* stream);
/* dx from dy; y is the forward OUTPUT (xhat), x the forward input (used for the ReLU mask when
 * relu != 0, may be NULL otherwise). */
int dp_bn_node_bwd(const float* x, int ldx, const float* y, int ldy, const float* stats, const float* dy,
                   int lddy, float* dx, int lddx, int B, int n, int F, int relu, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ A5  assignment head
 * S = softmax_K(z @ Wp^T + bp) * mask — encoders.py:1273-1275. z [B,n,Din] (ldz), Wp [K,Din]
 * (nn.Linear layout), S [B,n,K]. */
size_t dp_assign_workspace_bytes(int B, int n, int Din, int K);
int dp_assign_softmax_mask_fwd(const float* z, int ldz, const float* Wp, const float* bp,
                               const int* num_nodes, float* S, int B, int n, int Din, int K,
                               void* workspace, size_t workspace_bytes, void* stream);
int dp_assign_softmax_mask_bwd(const float* z, int ldz, const float* Wp, const float* S, const float* dS,
                               const int* num_nodes, float* dz, int lddz, float* dWp, float* dbp, int B,
                               int n, int Din, int K, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ------------------------------------------------------------------ A6  pooling
 * Xp = S^T Z [B,K,D], Ap = S^T A S [B,K,K] — encoders.py:1278-1279.
 * T [B,K,n] receives S^T A (saved for backward). */
int dp_pool_fwd(const float* S, const float* Z, int ldz, const float* adj, float* Xp, float* Ap, float* T,
                int B, int n, int K, int D, void* stream);
/* dS [B,n,K] (overwritten), dZ [B,n,D] (lddz; ACCUMULATED INTO), dadj [B,n,n] or NULL
 * (accumulated into). */
size_t dp_pool_bwd_workspace_bytes(int B, int n, int K, int D);
int dp_pool_bwd(const float* S, const float* Z, int ldz, const float* adj, const float* T, const float* dXp,
                const float* dAp, float* dS, float* dZ, int lddz, float* dadj, int B, int n, int K, int D,
                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ A7  max readout
 * out[b,f] = max_n (Z * mask)[b,n,f] — encoders.py:1079-1080,1257,1287.  argmax [B,F] int32
 * (-1 where a masked zero row wins: no gradient). */
int dp_masked_max_fwd(const float* Z, int ldz, const int* num_nodes, float* out, int ldo, int* argmax,
                      int B, int n, int F, void* stream);
/* dZ is ACCUMULATED INTO. */
int dp_masked_max_bwd(const float* dout, int ldo, const int* argmax, float* dZ, int lddz, int B, int n,
                      int F, void* stream);

/* ------------------------------------------------------------------ A8  link-prediction loss
 * loss = sum_{n,m < n_b} [-A log(P+1e-7) - (1-A) log(1-P+1e-7)] / sum_b n_b^2,
 * P = min(S S^T, 1) — encoders.py:1309-1331 (adj_hop = 1).  loss_out: 1 float. */
size_t dp_linkpred_workspace_bytes(int B, int n, int K);
int dp_linkpred_loss_fwd(const float* S, const float* adj, const int* num_nodes, float* loss_out, int B,
                         int n, int K, void* workspace, size_t workspace_bytes, void* stream);
/* dS (+)= dloss * d loss / dS.  dloss: device pointer to 1 float (NULL = 1.0). */
int dp_linkpred_loss_bwd(const float* S, const float* adj, const int* num_nodes, const float* dloss,
                         float* dS, int accumulate, int B, int n, int K, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ softmax cross entropy
 * loss = mean_b CE(logits_b, label_b) — F.cross_entropy, encoders.py:1127.  prob [B,C] is saved
 * for backward.  label: int64[B]. */
int dp_cross_entropy_fwd(const float* logits, const long long* label, float* loss_out, float* prob, int B,
                         int C, void* stream);
int dp_cross_entropy_bwd(const float* prob, const long long* label, const float* dloss, float* dlogits,
                         int B, int C, void* stream);

/* ------------------------------------------------------------------ A10  Set2Set
 * Set2Set.forward, set2set.py:32-57: n LSTM-attention steps over emb [B,n,d] -> out [B,d].
 * Weights in nn.LSTM layout: w_ih [4d,2d], w_hh [4d,d], b_ih [4d], b_hh [4d] (gates i,f,g,o);
 * pred: Wp [d,2d], bp [d].  `save` (dp_set2set_save_bytes) keeps per-step state for backward. */
size_t dp_set2set_save_bytes(int B, int n, int d);
int dp_set2set_fwd(const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                   const float* b_hh, const float* Wp, const float* bp, float* out, int B, int n, int d,
                   void* save, size_t save_bytes, void* stream);
size_t dp_set2set_bwd_workspace_bytes(int B, int n, int d);
int dp_set2set_bwd(const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                   const float* b_hh, const float* Wp, const float* bp, const float* out, const float* dout,
                   float* demb, int ldde, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dWp,
                   float* dbp, int B, int n, int d, const void* save, size_t save_bytes, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ A9  mean aggregator
 * out[i,:] = mean_{j in indices[indptr[i]:indptr[i+1]]} table[j,:] — MeanAggregator.forward with
 * num_sample=None, aggregators.py:50-62 (the row-normalised mask times the embedding matrix, as a
 * CSR gather-mean).  Rows with no neighbour give 0/0 = NaN in the reference; here too. */
int dp_mean_aggregate_fwd(const float* table, int ldt, const int* indptr, const int* indices, float* out,
                          int ldo, int n_rows, int feat, void* stream);
/* dtable [n_table, feat] is ACCUMULATED INTO (atomic adds). */
int dp_mean_aggregate_bwd(const float* dout, int ldo, const int* indptr, const int* indices, float* dtable,
                          int ldt, int n_rows, int feat, void* stream);

/* ------------------------------------------------------------------ N4  GraphConv on a CSR graph
 * The padded dense path stores a graph as an [N,N] block (load_data.py:79 drops graphs above max_nodes: DD's largest
 * has 5 748 nodes = 132 MB dense).  For such graphs the same layer runs on CSR:
 *     y = l2norm((A x [+ x]) W + b),   A x = sum over the neighbours listed in indices[indptr[i]:indptr[i+1]]
 * — GraphConv.forward (encoders.py:962-974) on ONE graph.  x [n,Fin] (ldx), y [n,Fout] (ldy); ax [n,Fin] receives
 * A x (+ x) and invnorm [n] the row norms' reciprocals (both saved for backward).  flags: DP_F_ADD_SELF | DP_F_NORMALIZE.
 * dp_csr_aggregate is the aggregation alone (mean != 0: the MeanAggregator's mean; beta: out = agg + beta * out). */
int dp_csr_aggregate(const float* table, int ldt, const int* indptr, const int* indices, float* out, int ldo,
                     int n_rows, int feat, int mean, float beta, void* stream);
size_t dp_sparse_gcn_layer_workspace_bytes(int n, int Fin, int Fout);
int dp_sparse_gcn_layer_fwd(const float* x, int ldx, const int* indptr, const int* indices, const float* W,
                            const float* bias, float* y, int ldy, float* ax, float* invnorm, int n, int Fin, int Fout,
                            int flags, void* workspace, size_t workspace_bytes, void* stream);
/* indptr_t / indices_t: CSR of A^T (the same arrays as the forward's for an undirected graph) — dx is then a gather,
 * deterministic; NULL: dx is accumulated with float atomics from the forward CSR.  dx may be NULL.  dW, db OVERWRITTEN. */
int dp_sparse_gcn_layer_bwd(const float* ax, const int* indptr, const int* indices, const int* indptr_t,
                            const int* indices_t, const float* W, const float* y, int ldy, const float* invnorm,
                            const float* dy, int lddy, float* dx, int lddx, float* dW, float* db, int n, int Fin,
                            int Fout, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* ================================================================== model-level entry points
 * One call enqueues the whole forward (or backward) of an encoder, so the Python host pays one
 * FFI crossing per pass instead of ~100.  Parameters live in ONE flat fp32 buffer; the cfg gives
 * the offset (in floats) of every tensor.  Gradients are written to a flat buffer of the same
 * layout (ready for a single RCCL all-reduce). */
typedef struct {
    int n_layers;                    /* L */
    int dims[DP_MAX_LAYERS + 1];     /* dims[0] = input width, dims[l] = output width of layer l */
    long w_off[DP_MAX_LAYERS];       /* weight [dims[l], dims[l+1]] offset in the flat buffer */
    long b_off[DP_MAX_LAYERS];       /* bias [dims[l+1]] offset, or -1 */
    long drop_off[DP_MAX_LAYERS];    /* layer-input dropout (GraphConv.forward, encoders.py:962-964): offset in the
                                        `dropout` buffer of a mask [B, n, dims[l]] holding 0 or 1/(1-p), or -1.
                                        Layer 0 (conv_first) has none in the reference; must be -1. */
} dp_stack_cfg;

/* Sync-BN exchange (data parallelism, SURVEY 8(e) mode ii).  apply_bn couples the whole batch (encoders.py:1048-1052);
 * when a batch is sharded over `bn_world` ranks the per-row partials every BatchNorm site produces must be gathered
 * from all ranks before they are combined.  The library calls `exchange` from inside dp_encoder_forward / backward,
 * between the launch that writes the local block and the launch that reads the gathered one; the callback must
 * ENQUEUE, on `stream`, an all-gather of `bytes_per_rank` bytes at `local` into `gathered` (rank-major,
 * bn_world * bytes_per_rank bytes; both are device pointers inside the call's workspace) and return 0. */
typedef int (*dp_exchange_fn)(void* user, const void* local, void* gathered, size_t bytes_per_rank, void* stream);

typedef struct {
    int B, N;                        /* batch, padded node count */
    int num_pooling;                 /* P */
    int n_nodes[DP_MAX_LEVELS + 1];  /* n_nodes[0] = N, n_nodes[j+1] = K_j (encoders.py:1203,1215) */
    dp_stack_cfg embed[DP_MAX_LEVELS + 1]; /* embed[0]: conv_first/.. ; embed[j+1]: after pool j */
    dp_stack_cfg assign[DP_MAX_LEVELS];    /* assign GCN of level j (input: assign_x / pooled X) */
    long assign_pred_w_off[DP_MAX_LEVELS]; /* Linear [K_j, Da_j] (encoders.py:1210) */
    long assign_pred_b_off[DP_MAX_LEVELS];
    int n_pred;                      /* number of Linear layers in pred_model (hidden + 1) */
    int pred_dims[DP_MAX_PRED + 2];  /* pred_dims[0] = D*(P+1) ... pred_dims[n_pred] = label_dim */
    long pred_w_off[DP_MAX_PRED + 1];
    long pred_b_off[DP_MAX_PRED + 1];
    int flags;                       /* DP_F_BN | DP_F_ADD_SELF */
    int readout;                     /* 0: max over nodes of the concat (DiffPool / base encoder),
                                        1: Set2Set (GcnSet2SetEncoder) */
    long s2s_off[6];                 /* w_ih, w_hh, b_ih, b_hh, pred.weight, pred.bias (readout 1) */
    int mask_readout;                /* 1: level-0 embedding is masked before readout (DiffPool,
                                        Set2Set encoder); 0: base encoder (encoders.py:1083-1122) */
    long n_params;                   /* total floats in the flat buffer */
    long n_graph_params;             /* params [0, n_graph_params) are the GCN stacks + assign heads
                                        (their gradients are reduced from per-graph slabs); the
                                        pred_model / Set2Set parameters follow */
    int bn_world;                    /* 0 / 1: apply_bn over the local batch; W > 1: over the W equal shards of a
                                        data-parallel batch (every rank passes the same B) through `exchange`.
                                        bn_world == 1 WITH `exchange` set runs the sync-BN launch sequence and every
                                        callback over one rank (a one-GPU rehearsal of the collective path) */
    dp_exchange_fn exchange;
    void* exchange_user;
} dp_encoder_cfg;

size_t dp_sizeof_encoder_cfg(void); /* sizeof(dp_encoder_cfg): lets a binding check its struct layout */
size_t dp_encoder_save_bytes(const dp_encoder_cfg* cfg);
/* The workspace of the model-level calls must be ZERO-FILLED ONCE after it is allocated (before its first use with
 * this cfg) and then left alone between calls: its first block holds the barrier state of the persistent level-0
 * kernel, which every launch leaves clean (also a launch whose barrier gave up).  A workspace with garbage there makes
 * the first forward time out: DP_DEVERR_BARRIER on the next entry, NaN logits, never a hang. */
size_t dp_encoder_workspace_bytes(const dp_encoder_cfg* cfg);

/* Where one saved activation of pooling level `level` sits inside the `save` buffer after dp_encoder_forward: the
 * reference keeps only the LAST level's S as `self.assign_tensor` (encoders.py:1276) for train.py:218-219's logging;
 * this gives a caller (and the parity tests) every level's S_j [B,n_j,K_j], X'_j [B,K_j,D], A'_j [B,K_j,K_j]
 * (encoders.py:1278-1279) and the embeddings Z_j [B,n_j,D] / assign-stack outputs [B,n_j,Da_j] without a second pass.
 * offset is in BYTES from `save`, count in 4-byte elements.  Returns DP_ERR_INVALID_ARG for a level without that tensor. */
#define DP_SAVE_S 0
#define DP_SAVE_XPOOL 1
#define DP_SAVE_ADJPOOL 2
#define DP_SAVE_Z 3
#define DP_SAVE_ZASSIGN 4
#define DP_SAVE_ARGMAX 5   /* int32 [B, readout width]: winning row of the max readout (-1: a masked zero row) */
int dp_encoder_save_locate(const dp_encoder_cfg* cfg, int level, int field, size_t* offset, size_t* count);

/* SoftPoolingGcnEncoder.forward (encoders.py:1231-1300), GcnEncoderGraph.forward (:1083-1122,
 * num_pooling = 0) and GcnSet2SetEncoder.forward (:1144-1157, readout = 1).
 * x [B,N,F], adj [B,N,N], assign_x [B,N,Fa] (may alias x), num_nodes int32[B] or NULL.
 * ypred [B,label_dim]; assign_out [B,N,K_0] (level-0 S, = assign_tensor when P = 1) or NULL when
 * P = 0; `save` keeps activations for backward.  dropout: the mask buffer addressed by the stacks' drop_off
 * (training with GraphConv dropout > 0; the caller draws the masks and passes the SAME buffer to backward), or
 * NULL (evaluation, or dropout 0: every drop_off is ignored). */
#define DP_MODE_EVAL 0   /* forward only (train.py:30-58 evaluate): nothing is prepared for a backward pass */
#define DP_MODE_TRAIN 1  /* a dp_encoder_backward with the SAME save and workspace buffers will follow: the forward also
                            clears the backward pass's accumulators (they live at the start of the workspace) on the side
                            of its adjacency-pack kernel, so the backward may be called with prezeroed = 1 — provided
                            nothing else used that workspace in between and it is the first backward of this forward */
/* labels_out: int64 [B] or NULL — the arg-max class of every graph (first index on ties), written by the prediction
 * head's own launch: evaluate() (train.py:42-44) moves B integers to the host instead of B x C logits. */
int dp_encoder_forward(const dp_encoder_cfg* cfg, const float* params, const float* x, const float* adj,
                       const float* assign_x, const int* num_nodes, const float* dropout, float* ypred,
                       float* assign_out, long long* labels_out, void* save, size_t save_bytes, void* workspace,
                       size_t workspace_bytes, int mode, void* stream);
/* d_ypred [B,label_dim]; d_assign [B,N,K_0] or NULL (gradient arriving at the level-0 assignment
 * from the link-prediction loss); grads: flat, same layout as params, OVERWRITTEN.
 * prezeroed: 1 iff the accumulators were cleared by a DP_MODE_TRAIN forward (see above); 0: this call clears them. */
int dp_encoder_backward(const dp_encoder_cfg* cfg, const float* params, const float* x, const float* adj,
                        const float* assign_x, const int* num_nodes, const float* dropout, const float* d_ypred,
                        const float* d_assign, float* grads, const void* save, size_t save_bytes,
                        void* workspace, size_t workspace_bytes, int prezeroed, void* stream);

/* The same two calls for a level-0 adjacency that is ALREADY in the packed form the kernels multiply from — bf16 rows
 * [B, N, dp_adj_pack_ld(N)] of A (adj_pk) and of A^T (adj_pkt; the same buffer for a symmetric adjacency), as written
 * by dp_build_batch_packed or dp_adj_pack: no fp32 [B,N,N] batch is written, read or converted (the
 * end-to-end training step).  The bf16 values ARE the adjacency (nothing to round).  Only configurations that take the
 * persistent level-0 plan accept it (N >= 64, N % 4 == 0, B * ceil(N / RB) <= CUs, no sync-BN, no add_self):
 * DP_ERR_UNSUPPORTED otherwise — the fp32 entries above serve every configuration and train.py.  The link-prediction
 * loss (dp_loss_forward with linkpred) reads the fp32 adjacency and has no packed form. */
int dp_encoder_forward_packed(const dp_encoder_cfg* cfg, const float* params, const float* x, const void* adj_pk,
                              const void* adj_pkt, const float* assign_x, const int* num_nodes, const float* dropout,
                              float* ypred, float* assign_out, long long* labels_out, void* save, size_t save_bytes,
                              void* workspace, size_t workspace_bytes, int mode, void* stream);
int dp_encoder_backward_packed(const dp_encoder_cfg* cfg, const float* params, const float* x, const void* adj_pk,
                               const void* adj_pkt, const float* assign_x, const int* num_nodes, const float* dropout,
                               const float* d_ypred, const float* d_assign, float* grads, const void* save,
                               size_t save_bytes, void* workspace, size_t workspace_bytes, int prezeroed, void* stream);

/* SoftPoolingGcnEncoder.loss (encoders.py:1302-1334): loss_out[0] = CE (+ link), loss_out[1] = link.
 * prob [B,C] saved for backward.  S / adj may be NULL when linkpred == 0.
 * d_ypred_unit [B,C] (may be NULL): d loss / d ypred for an upstream gradient of 1, (softmax - onehot) / B, written by
 * the same launch — `loss.backward()` (train.py:208) sends exactly that down, so the caller can hand it to
 * dp_encoder_backward without a gradient launch of its own. */
size_t dp_loss_workspace_bytes(int B, int N, int K, int linkpred);
/* link_norm (device scalar or NULL): replaces the link loss's normaliser sum_b n_b^2 (encoders.py:1326,1331).  Under
 * data parallelism every rank passes (sum over ALL ranks' graphs) / world_size, so that the mean over ranks of the
 * per-rank losses and gradients is the loss of the concatenated batch. */
int dp_loss_forward(const float* ypred, const long long* label, const float* S, const float* adj,
                    const int* num_nodes, const float* link_norm, float* loss_out, float* prob, float* d_ypred_unit,
                    int B, int C, int N, int K, int linkpred, void* workspace, size_t workspace_bytes, void* stream);
/* dloss: device scalar (NULL = 1).  d_ypred [B,C] (NULL: not wanted — the caller uses d_ypred_unit), dS [B,N,K]
 * (only when linkpred) overwritten. */
int dp_loss_backward(const float* prob, const long long* label, const float* S, const float* adj,
                     const int* num_nodes, const float* link_norm, const float* dloss, float* d_ypred, float* dS,
                     int B, int C, int N, int K, int linkpred, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ N1  on-device batch builder
 * Replaces GraphSampler.__getitem__ + collate + H2D of the dense batch (graph_sampler.py:97-109, train.py:197-201):
 * the host ships the batch's edge lists, the device writes adj [B,N,N] (0/1, zero padded), the node features
 * feats [B,N,Fout] (NULL: skip) and num_nodes [B].  Graph b owns edges [edge_ptr[b], edge_ptr[b+1]) with node ids
 * local to the graph, and node labels label[node_ptr[b] .. node_ptr[b+1]).  symmetric != 0 also sets adj[d][s]
 * (undirected edge lists that store each edge once).
 * feature_mode (graph_sampler.py:33-59): 0 one-hot node label (Fout = F), 1 identity (Fout = N), 2 degree as one
 * column (Fout = 1), 3 one-hot degree capped at 10 then the one-hot label (Fout = 11 + F); modes 2 and 3 need the
 * `degree` workspace (B*N ints).  assign_feats [B,N,N+Fout] (NULL: skip) = [identity | feats], the sampler's
 * assign_feat='id' (graph_sampler.py:85-87).
 * errors: device int, receives the number of skipped out-of-range entries (0 = clean batch).
 * max_edges_per_graph only sizes the launch. */
int dp_build_batch(const int* edge_src, const int* edge_dst, const int* edge_ptr, const int* node_label,
                   const int* node_ptr, float* adj, float* feats, float* assign_feats, int* num_nodes, int* errors,
                   int* degree, int B, int N, int F, int feature_mode, int symmetric, int max_edges_per_graph,
                   void* stream);

/* The same builder writing the adjacency straight into the packed form of dp_adj_pack (bf16 rows, ld =
 * dp_adj_pack_ld(N); 1.0 = 0x3F80): adj_pk receives A, adj_pkt receives A^T — for symmetric != 0 the two are equal
 * and the caller may pass the SAME buffer for both (dp_adj_pack_bytes(B, N) bytes each).  Feeds
 * dp_encoder_forward_packed / _backward_packed: the fp32 [B,N,N] batch is never materialised. */
int dp_build_batch_packed(const int* edge_src, const int* edge_dst, const int* edge_ptr, const int* node_label,
                          const int* node_ptr, void* adj_pk, void* adj_pkt, float* feats, float* assign_feats,
                          int* num_nodes, int* errors, int* degree, int B, int N, int F, int feature_mode, int symmetric,
                          int max_edges_per_graph, void* stream);
/* The input arrays of both builders may live in pinned host memory that the device can address (hipHostMalloc): the
 * kernels then fetch the ~190 KB of a DD batch over PCIe themselves and a captured training step needs no copy node.
 * dp_gather_labels copies the B graph labels the same way (int64, host-visible source -> device). */
int dp_gather_labels(const long long* graph_label, long long* label_out, int B, void* stream);

/* ------------------------------------------------------------------ N2  fused gradient clip + Adam step
 * train.py:209-210 on top of the Adam of train.py:173, over the flat fp32 parameter / gradient buffers (n floats):
 *   total = ||grads||_2;  grads *= min(1, max_norm / (total + 1e-6))   (max_norm <= 0: no clipping)
 *   exp_avg = b1 exp_avg + (1-b1) g;  exp_avg_sq = b2 exp_avg_sq + (1-b2) g^2
 *   params -= lr / (1 - b1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - b2^step) + eps)
 * step >= 1 is the 1-based count of this update.  total_norm_out: device float or NULL.  Two launches. */
size_t dp_clip_adam_workspace_bytes(void);
int dp_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, int step, float lr,
                      float beta1, float beta2, float eps, float max_norm, float* total_norm_out, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same update with the step count kept ON THE DEVICE: step_counter (device int, the number of updates done so far;
 * zero it once) is incremented by the call and the bias corrections of the new count are computed by the kernel (in
 * double, as above).  No argument changes from step to step, so the call can sit inside a captured hipGraph together
 * with dp_build_batch_packed, dp_encoder_forward_packed, dp_loss_forward and dp_encoder_backward_packed: one graph
 * launch per training step (train.py:197-210). */
int dp_clip_adam_step_counted(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, int* step_counter,
                              float lr, float beta1, float beta2, float eps, float max_norm, float* total_norm_out,
                              void* workspace, size_t workspace_bytes, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DIFFPOOL_HIP_H */
