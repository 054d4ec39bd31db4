// On-device batch builder: what GraphSampler.__getitem__ + the DataLoader collate + `.cuda()`
// deliver per step (graph_sampler.py:97-109, train.py:197-201) — `adj [B,N,N]` 0/1 zero-padded, one-hot
// `feats [B,N,F]`, `num_nodes [B]` — built on the GPU from the batch's edge lists.  The host ships the edges
// (~10 bytes per edge: 190 KB for a DD batch) instead of the dense fp32 adjacency (20 MB, ~320 us of PCIe).
//   edges of graph b: (src[e], dst[e]) for e in [edge_ptr[b], edge_ptr[b+1]), node ids local to the graph;
//   node labels of graph b: label[i] for i in [node_ptr[b], node_ptr[b+1]).
// Out-of-range entries (node id >= n_b or >= N, label outside [0,F)) are skipped and counted in `errors`.
#include "dp_common.h"

namespace dp {

__global__ __launch_bounds__(256) void k_scatter_edges(const int* src, const int* dst, const int* edge_ptr,
                                                       const int* node_ptr, float* adj, int B, int N, int symmetric,
                                                       int* errors) {
    const int b = blockIdx.y;
    const int e0 = edge_ptr[b], e1 = edge_ptr[b + 1];
    const int nb = min(node_ptr[b + 1] - node_ptr[b], N);
    float* A = adj + (long)b * N * N;
    int bad = 0;
    for (int e = e0 + blockIdx.x * 256 + threadIdx.x; e < e1; e += gridDim.x * 256) {
        const int s = src[e], d = dst[e];
        if (s < 0 || d < 0 || s >= nb || d >= nb) {
            ++bad;
            continue;
        }
        A[(long)s * N + d] = 1.f;                 // duplicates write the same value: no atomics needed
        if (symmetric) A[(long)d * N + s] = 1.f;
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ __launch_bounds__(256) void k_onehot_nodes(const int* label, const int* node_ptr, float* feats,
                                                      int* num_nodes, int B, int N, int F, int* errors) {
    const int b = blockIdx.y;
    const int p0 = node_ptr[b], n = node_ptr[b + 1] - p0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        num_nodes[b] = min(n, N);
        if (n > N) atomicAdd(errors, n - N);
    }
    if (!feats) return;
    float* Fb = feats + (long)b * N * F;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < min(n, N); i += gridDim.x * 256) {
        const int l = label[p0 + i];
        if (l < 0 || l >= F) atomicAdd(errors, 1);
        else Fb[(long)i * F + l] = 1.f;
    }
}

void build_batch(Seq& q, const int* src, const int* dst, const int* edge_ptr, const int* label, const int* node_ptr,
                 float* adj, float* feats, int* num_nodes, int* errors, int B, int N, int F, int symmetric,
                 int max_edges_per_graph) {
    if (!q.ok() || B <= 0) return;
    zero_fill(q, adj, (size_t)B * N * N * sizeof(float));
    if (feats) zero_fill(q, feats, (size_t)B * N * F * sizeof(float));
    zero_small(q, errors, sizeof(int));
    int gx = (max_edges_per_graph + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(k_scatter_edges, dim3(gx, B), dim3(256), 0, q.stream, src, dst, edge_ptr, node_ptr, adj, B, N,
                       symmetric, errors);
    q.check_launch("scatter_edges");
    int gn = (N + 255) / 256;
    hipLaunchKernelGGL(k_onehot_nodes, dim3(gn, B), dim3(256), 0, q.stream, label, node_ptr, feats, num_nodes, B, N, F,
                       errors);
    q.check_launch("onehot_nodes");
}

}  // namespace dp
