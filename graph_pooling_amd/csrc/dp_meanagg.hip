// GraphSAGE mean aggregator (MeanAggregator.forward with num_sample=None, aggregators.py:30-63).
// The reference builds a dense 0/1 [n_batch, n_unique] mask, row-normalises it and multiplies it
// by the gathered embedding matrix (aggregators.py:50-62); the same numbers come out of a CSR
// gather-mean, which is HBM/L2-bound row gathering: one 16-lane team per 64-byte slice of a row.
#include "dp_common.h"

namespace dp {

__global__ __launch_bounds__(256) void k_mean_agg_fwd(const float* table, int ldt, const int* indptr,
                                                      const int* indices, float* out, int ldo, int n_rows, int feat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int beg = indptr[row], end = indptr[row + 1];
    const float inv = 1.f / (float)(end - beg);   // 0 neighbours -> inf * 0 = NaN, as mask.div(0) gives in torch
    for (int f = lane; f < feat; f += 64) {
        float s = 0.f;
        for (int e = beg; e < end; ++e) s += table[(long)indices[e] * ldt + f];
        out[(long)row * ldo + f] = s * inv;
    }
}

__global__ __launch_bounds__(256) void k_mean_agg_bwd(const float* dout, int ldo, const int* indptr,
                                                      const int* indices, float* dtable, int ldt, int n_rows,
                                                      int feat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int beg = indptr[row], end = indptr[row + 1];
    const float inv = 1.f / (float)(end - beg);
    for (int f = lane; f < feat; f += 64) {
        const float g = dout[(long)row * ldo + f] * inv;
        for (int e = beg; e < end; ++e) atomicAdd(&dtable[(long)indices[e] * ldt + f], g);
    }
}

void mean_aggregate_fwd(Seq& q, const float* table, int ldt, const int* indptr, const int* indices, float* out,
                        int ldo, int n_rows, int feat) {
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_mean_agg_fwd, dim3((n_rows + 3) / 4), dim3(256), 0, q.stream, table, ldt, indptr, indices,
                       out, ldo, n_rows, feat);
    q.check_launch("mean_aggregate_fwd");
}
void mean_aggregate_bwd(Seq& q, const float* dout, int ldo, const int* indptr, const int* indices, float* dtable,
                        int ldt, int n_rows, int feat) {
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_mean_agg_bwd, dim3((n_rows + 3) / 4), dim3(256), 0, q.stream, dout, ldo, indptr, indices,
                       dtable, ldt, n_rows, feat);
    q.check_launch("mean_aggregate_bwd");
}

}  // namespace dp
