// CSR neighbour aggregation: the GraphSAGE mean aggregator (MeanAggregator.forward with num_sample=None,
// aggregators.py:30-63) and, with mean = 0, the plain neighbour SUM  A x  of GraphConv (encoders.py:965) for graphs
// too large for the padded dense adjacency (DD's largest graph has 5 748 nodes: 132 MB as a dense fp32 matrix,
// 0.1 MB as CSR).  The reference builds a dense 0/1 [n_batch, n_unique] mask, row-normalises it and multiplies it by
// the gathered embedding matrix (aggregators.py:50-62); the same numbers come out of a CSR gather.
//
// One wavefront per output row, lanes across the feature columns (a neighbour's row is one coalesced 256-byte
// request per 64 columns).  The row's neighbour ids are fetched ONCE, 64 at a time, as one coalesced load and handed
// round by readlane; four neighbour rows are in flight per lane before the first add (the first version walked
// indices[e] -> table[...] as one dependent chain per neighbour and feature).  HBM/L2-bound: nnz * feat * 4 bytes.
#include "dp_common.h"

namespace dp {

__global__ __launch_bounds__(256) void k_csr_agg_fwd(const float* table, int ldt, const int* indptr,
                                                     const int* indices, float* out, int ldo, int n_rows, int feat,
                                                     int mean, float beta) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int beg = indptr[row], end = indptr[row + 1];
    // 0 neighbours: mean -> 0 * inf = NaN exactly as mask.div(0) gives in torch; sum -> 0
    const float scale = mean ? 1.f / (float)(end - beg) : 1.f;
    for (int f0 = 0; f0 < feat; f0 += 64) {
        const int f = min(f0 + lane, feat - 1);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int e0 = beg; e0 < end; e0 += 64) {
            const int cnt = min(64, end - e0);
            const int mine = indices[e0 + min(lane, cnt - 1)];
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {
                const long r0 = __builtin_amdgcn_readlane(mine, j), r1 = __builtin_amdgcn_readlane(mine, j + 1);
                const long r2 = __builtin_amdgcn_readlane(mine, j + 2), r3 = __builtin_amdgcn_readlane(mine, j + 3);
                const float v0 = table[r0 * ldt + f], v1 = table[r1 * ldt + f];
                const float v2 = table[r2 * ldt + f], v3 = table[r3 * ldt + f];
                s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            }
            for (; j < cnt; ++j) s0 += table[(long)__builtin_amdgcn_readlane(mine, j) * ldt + f];
        }
        if (f0 + lane < feat) {
            float v = ((s0 + s1) + (s2 + s3)) * scale;
            float* o = out + (long)row * ldo + f;
            *o = beta != 0.f ? v + beta * *o : v;
        }
    }
}
// readlane needs a wave-uniform lane index: j is uniform (loop counter), fine.

// scatter form of the transpose (general CSR without a transposed copy): dtable[indices[e]] += g / deg — float atomics
__global__ __launch_bounds__(256) void k_csr_agg_bwd_scatter(const float* dout, int ldo, const int* indptr,
                                                             const int* indices, float* dtable, int ldt, int n_rows,
                                                             int feat, int mean) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int beg = indptr[row], end = indptr[row + 1];
    const float scale = mean ? 1.f / (float)(end - beg) : 1.f;
    for (int f = lane; f < feat; f += 64) {
        const float g = dout[(long)row * ldo + f] * scale;
        for (int e = beg; e < end; ++e) atomicAdd(&dtable[(long)indices[e] * ldt + f], g);
    }
}

void csr_aggregate_fwd(Seq& q, const float* table, int ldt, const int* indptr, const int* indices, float* out,
                       int ldo, int n_rows, int feat, int mean, float beta) {
    if (!q.ok() || n_rows <= 0 || feat <= 0) return;
    hipLaunchKernelGGL(k_csr_agg_fwd, dim3((n_rows + 3) / 4), dim3(256), 0, q.stream, table, ldt, indptr, indices,
                       out, ldo, n_rows, feat, mean, beta);
    q.check_launch("csr_aggregate_fwd");
}
void csr_aggregate_bwd_scatter(Seq& q, const float* dout, int ldo, const int* indptr, const int* indices,
                               float* dtable, int ldt, int n_rows, int feat, int mean) {
    if (!q.ok() || n_rows <= 0 || feat <= 0) return;
    hipLaunchKernelGGL(k_csr_agg_bwd_scatter, dim3((n_rows + 3) / 4), dim3(256), 0, q.stream, dout, ldo, indptr,
                       indices, dtable, ldt, n_rows, feat, mean);
    q.check_launch("csr_aggregate_bwd");
}

void mean_aggregate_fwd(Seq& q, const float* table, int ldt, const int* indptr, const int* indices, float* out,
                        int ldo, int n_rows, int feat) {
    csr_aggregate_fwd(q, table, ldt, indptr, indices, out, ldo, n_rows, feat, 1, 0.f);
}
void mean_aggregate_bwd(Seq& q, const float* dout, int ldo, const int* indptr, const int* indices, float* dtable,
                        int ldt, int n_rows, int feat) {
    csr_aggregate_bwd_scatter(q, dout, ldo, indptr, indices, dtable, ldt, n_rows, feat, 1);
}

}  // namespace dp
