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
                                                       int* errors, int* degree) {
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
        if (degree) {                             // node degrees for the deg / deg-num feature modes (the edge list
            atomicAdd(degree + (long)b * N + s, 1);   // holds every undirected edge once, or both directions when
            if (symmetric) atomicAdd(degree + (long)b * N + d, 1);   // symmetric == 0)
        }
    }
    if (bad) atomicAdd(errors, bad);
}

// The same scatter into the PACKED adjacency the encoders multiply from (dp_adj_pack's layout: bf16 rows of ld = N
// rounded up to 8 elements): 1.0 = 0x3F80.  P holds A, Pt holds A^T; for an undirected edge list (symmetric != 0) A is
// its own transpose and the caller may pass Pt == P (one 2-byte store pair per edge instead of two).
__global__ __launch_bounds__(256) void k_scatter_edges_packed(const int* src, const int* dst, const int* edge_ptr,
                                                              const int* node_ptr, unsigned short* P, unsigned short* Pt,
                                                              int B, int N, int ld, int symmetric, int* errors,
                                                              int* degree) {
    const int b = blockIdx.y;
    const int e0 = edge_ptr[b], e1 = edge_ptr[b + 1];
    const int nb = min(node_ptr[b + 1] - node_ptr[b], N);
    unsigned short* A = P + (long)b * N * ld;
    unsigned short* At = Pt + (long)b * N * ld;
    const unsigned short one = 0x3F80;
    int bad = 0;
    for (int e = e0 + blockIdx.x * 256 + threadIdx.x; e < e1; e += gridDim.x * 256) {
        const int s = src[e], d = dst[e];
        if (s < 0 || d < 0 || s >= nb || d >= nb) {
            ++bad;
            continue;
        }
        A[(long)s * ld + d] = one;
        if (symmetric) A[(long)d * ld + s] = one;
        if (At != A) {
            At[(long)d * ld + s] = one;
            if (symmetric) At[(long)s * ld + d] = one;
        }
        if (degree) {
            atomicAdd(degree + (long)b * N + s, 1);
            if (symmetric) atomicAdd(degree + (long)b * N + d, 1);
        }
    }
    if (bad) atomicAdd(errors, bad);
}

// Node features in the sampler's modes (graph_sampler.py:33-59, 85-87); feats / assign are zero-filled beforehand.
//   0 default: one-hot node label            1 id: identity of size N (padded rows too)
//   2 deg-num: the degree, one column        3 deg: one-hot degree capped at 10, then the one-hot node label
// assign != null additionally writes [identity(N) | feats] (assign_feat='id').
__global__ __launch_bounds__(256) void k_node_features(const int* label, const int* node_ptr, const int* degree,
                                                       float* feats, float* assign, int* num_nodes, int B, int N, int F,
                                                       int Fout, int mode, int* errors) {
    const int b = blockIdx.y;
    const int p0 = node_ptr[b], n = node_ptr[b + 1] - p0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        num_nodes[b] = min(n, N);
        if (n > N) atomicAdd(errors, n - N);
    }
    const int nb = min(n, N);
    const int la = N + Fout;                           // row width of the assign features
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        float* f = feats ? feats + ((long)b * N + i) * Fout : nullptr;
        float* as = assign ? assign + ((long)b * N + i) * la : nullptr;
        if (as) as[i] = 1.f;
        auto put = [&](int c, float v) {
            if (f) f[c] = v;
            if (as) as[N + c] = v;
        };
        if (mode == 1) {
            put(i, 1.f);
            continue;
        }
        if (i >= nb) continue;
        int l = -1;
        if (mode == 0 || mode == 3) {
            l = label[p0 + i];
            if (l < 0 || l >= F) {
                atomicAdd(errors, 1);
                l = -1;
            }
        }
        if (mode == 0) {
            if (l >= 0) put(l, 1.f);
        } else if (mode == 2) {
            put(0, (float)degree[(long)b * N + i]);
        } else {
            put(min(degree[(long)b * N + i], 10), 1.f);
            if (l >= 0) put(11 + l, 1.f);
        }
    }
}

// adj (fp32 dense) or pk / pkt (packed bf16, ld = adj_pack_ld(N)): exactly one of the two forms is written
void build_batch(Seq& q, const int* src, const int* dst, const int* edge_ptr, const int* label, const int* node_ptr,
                 float* adj, unsigned short* pk, unsigned short* pkt, float* feats, float* assign, int* num_nodes,
                 int* errors, int* degree, int B, int N, int F, int mode, int symmetric, int max_edges_per_graph) {
    if (!q.ok() || B <= 0) return;
    const int Fout = mode == 0 ? F : (mode == 1 ? N : (mode == 2 ? 1 : 11 + F));
    const int ld = adj_pack_ld(N);
    if (adj) zero_fill(q, adj, (size_t)B * N * N * sizeof(float));
    else {
        zero_fill(q, pk, (size_t)B * N * ld * sizeof(unsigned short));
        if (pkt != pk) zero_fill(q, pkt, (size_t)B * N * ld * sizeof(unsigned short));
    }
    if (feats) zero_fill(q, feats, (size_t)B * N * Fout * sizeof(float));
    if (assign) zero_fill(q, assign, (size_t)B * N * (N + Fout) * sizeof(float));
    const bool need_deg = mode == 2 || mode == 3;
    if (need_deg) zero_fill(q, degree, (size_t)B * N * sizeof(int));
    zero_small(q, errors, sizeof(int));
    int gx = (max_edges_per_graph + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    if (adj)
        hipLaunchKernelGGL(k_scatter_edges, dim3(gx, B), dim3(256), 0, q.stream, src, dst, edge_ptr, node_ptr, adj, B, N,
                           symmetric, errors, need_deg ? degree : (int*)nullptr);
    else
        hipLaunchKernelGGL(k_scatter_edges_packed, dim3(gx, B), dim3(256), 0, q.stream, src, dst, edge_ptr, node_ptr, pk,
                           pkt, B, N, ld, symmetric, errors, need_deg ? degree : (int*)nullptr);
    q.check_launch("scatter_edges");
    int gn = (N + 255) / 256;
    hipLaunchKernelGGL(k_node_features, dim3(gn, B), dim3(256), 0, q.stream, label, node_ptr, degree, feats, assign,
                       num_nodes, B, N, F, Fout, mode, errors);
    q.check_launch("node_features");
}

__global__ __launch_bounds__(256) void k_gather_labels(const long long* src, long long* dst, int B) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) dst[i] = src[i];
}
void gather_labels(Seq& q, const long long* src, long long* dst, int B) {
    if (!q.ok() || B <= 0) return;
    hipLaunchKernelGGL(k_gather_labels, dim3((B + 255) / 256), dim3(256), 0, q.stream, src, dst, B);
    q.check_launch("gather_labels");
}

}  // namespace dp
