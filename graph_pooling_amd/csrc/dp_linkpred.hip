// Link-prediction auxiliary loss of DiffPool (SoftPoolingGcnEncoder.loss, encoders.py:1309-1331):
//   P = min(S S^T, 1);  l = -A log(P + eps) - (1 - A) log(1 - P + eps), eps = 1e-7;
//   zero outside the n_b x n_b block; loss = sum(l) / sum_b n_b^2.
// v1: P is formed per graph by the MFMA contraction, then one elementwise pass produces the
// per-block partial sums (forward) or D = d loss / d P in place (backward);
// dS = (D + D^T) S by two more contractions.
#include "dp_common.h"

namespace dp {

#define LINK_EPS 1e-7f

__device__ inline float block_sum_256(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// mode 0: partial loss sums;  mode 1: P <- D (gradient w.r.t. P, already scaled)
__global__ __launch_bounds__(256) void k_link_elem(float* P, const float* adj, const int* num_nodes, int n,
                                                   float* partial, const float* scale_ptr, int mode) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int nb = num_nodes ? min(num_nodes[b], n) : n;
    float* p = P + (long)b * n * n;
    const float* a = adj + (long)b * n * n;
    const float scale = (mode == 1) ? scale_ptr[0] : 0.f;
    float acc = 0.f;
    const long total = (long)n * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / n), c = (int)(i % n);
        const bool valid = r < nb && c < nb;
        if (mode == 0) {
            if (valid) {
                const float pv = fminf(p[i], 1.f);
                const float av = a[i];
                acc += -av * logf(pv + LINK_EPS) - (1.f - av) * logf(1.f - pv + LINK_EPS);
            }
        } else {
            float d = 0.f;
            if (valid) {
                const float raw = p[i];
                const float pv = fminf(raw, 1.f);
                const float av = a[i];
                // torch.min(a, b) backward: full gradient where a < b, half on ties, none above
                const float gate = raw < 1.f ? 1.f : (raw == 1.f ? 0.5f : 0.f);
                d = scale * gate * (-av / (pv + LINK_EPS) + (1.f - av) / (1.f - pv + LINK_EPS));
            }
            p[i] = d;
        }
    }
    if (mode == 0) {
        const float s = block_sum_256(acc, red);
        if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = s;
    }
}

// loss = sum(partials) / sum_b n_b^2  (single block; deterministic)
__global__ __launch_bounds__(256) void k_link_final(const float* partial, int count, const int* num_nodes, int B,
                                                    int n, float* out, float* scale_out, const float* dloss) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial ? partial[i] : 0.f;
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        double nn = 0.0;
        for (int b = 0; b < B; ++b) {
            const double v = num_nodes ? (double)min(num_nodes[b], n) : (double)n;
            nn += v * v;
        }
        if (out) out[0] = (float)((double)s / nn);
        if (scale_out) scale_out[0] = (float)((dloss ? (double)dloss[0] : 1.0) / nn);
    }
}

static int link_blocks(int n) {
    long total = (long)n * n;
    long blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

void linkpred_fwd(Seq& q, const float* S, int lds, const float* adj, const int* num_nodes, float* loss_out, int B,
                  int n, int K) {
    if (q.err) return;
    float* P = q.alloc<float>((size_t)B * n * n);
    const int nblk = link_blocks(n);
    float* partial = q.alloc<float>((size_t)B * nblk);
    if (!q.ok()) return;
    bgemm(q, S, S, P, nullptr, B, n, n, K, lds, lds, n, (long)n * lds, (long)n * lds, (long)n * n, false, true,
          1.f, 0.f, 0);
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_link_elem, dim3(nblk, B), dim3(256), 0, q.stream, P, adj, num_nodes, n, partial,
                       (const float*)nullptr, 0);
    q.check_launch("link_elem");
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_link_final, dim3(1), dim3(256), 0, q.stream, partial, B * nblk, num_nodes, B, n, loss_out,
                       (float*)nullptr, (const float*)nullptr);
    q.check_launch("link_final");
}

void linkpred_bwd(Seq& q, const float* S, int lds, const float* adj, const int* num_nodes, const float* dloss,
                  float* dS, int ldds, int B, int n, int K, int accumulate) {
    if (q.err) return;
    float* P = q.alloc<float>((size_t)B * n * n);
    float* scale = q.alloc<float>(64);
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_link_final, dim3(1), dim3(256), 0, q.stream, (const float*)nullptr, 0, num_nodes, B, n,
                       (float*)nullptr, scale, dloss);
    q.check_launch("link_scale");
    bgemm(q, S, S, P, nullptr, B, n, n, K, lds, lds, n, (long)n * lds, (long)n * lds, (long)n * n, false, true,
          1.f, 0.f, 0);
    if (!q.ok()) return;
    hipLaunchKernelGGL(k_link_elem, dim3(link_blocks(n), B), dim3(256), 0, q.stream, P, adj, num_nodes, n,
                       (float*)nullptr, (const float*)scale, 1);
    q.check_launch("link_elem_bwd");
    // dS = D S + D^T S
    bgemm(q, P, S, dS, nullptr, B, n, K, n, n, lds, ldds, (long)n * n, (long)n * lds, (long)n * ldds, false, false,
          1.f, accumulate ? 1.f : 0.f, 0);
    bgemm(q, P, S, dS, nullptr, B, n, K, n, n, lds, ldds, (long)n * n, (long)n * lds, (long)n * ldds, true, false,
          1.f, 1.f, 0);
}

}  // namespace dp
