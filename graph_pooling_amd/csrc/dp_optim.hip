// One optimiser step over the FLAT parameter buffer: the reference does
//   nn.utils.clip_grad_norm(model.parameters(), args.clip); optimizer.step()      (train.py:209-210, Adam of :173)
// as ~100 small launches over 24 tensors.  Here: one pass for the squared norm (partials per workgroup, summed in
// a fixed order so the result is deterministic), one pass that every workgroup starts by folding those partials
// into the clip coefficient and then applies to its slice:  g *= min(1, max_norm / (||g|| + 1e-6));  Adam with the
// bias corrections folded into step_size = lr / (1 - b1^t) and 1 / sqrt(1 - b2^t) by the host (as torch does) — or, for
// a step captured in a hipGraph, by the update kernel itself from a device-resident step counter.
// A NON-FINITE gradient norm (a poisoned backward pass, diffpool_hip.h "Device-side failures") skips the whole update —
// parameters, moments and gradients stay as they are — and raises DP_DEVERR_NONFINITE_GRAD; torch would write NaN
// into every parameter (clip_grad_norm_ with its default error_if_nonfinite=False, then Adam).
#include "dp_common.h"

namespace dp {

constexpr int OPT_WGS = 256;

// step_counter (device int or null): the number of updates done so far — counted up here, one launch ahead of the
// update kernel that reads it (a captured training step has no host to count for it)
__global__ __launch_bounds__(256) void k_sqnorm_partials(const float* g, long n, float* partial, int* step_counter) {
    __shared__ float red[4];
    if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) step_counter[0] += 1;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
    s = wave64_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_clip_adam(float* p, float* g, float* m, float* v, long n, const float* partial,
                                                   int npartial, float max_norm, float beta1, float beta2, float eps,
                                                   float step_size, float inv_bc2_sqrt, float* total_norm_out,
                                                   int* dev_err, const int* step_counter, float lr) {
    __shared__ float coef_s, step_s[2];
    if (step_counter && threadIdx.x == 64) {
        // the bias corrections of update t in double, as the host entry (and torch.optim.Adam) computes them
        const double t = (double)step_counter[0];
        const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
        step_s[0] = (float)((double)lr / bc1);
        step_s[1] = (float)(1.0 / sqrt(bc2));
    }
    if (threadIdx.x < 64) {
        float s = 0.f;
        if (partial)
            for (int i = threadIdx.x; i < npartial; i += 64) s += partial[i];
        s = wave64_sum(s);
        if (threadIdx.x == 0) {
            const float tn = sqrtf(s);
            float c = 1.f;
            if (partial && max_norm > 0.f) c = fminf(max_norm / (tn + 1e-6f), 1.f);
            if (partial && !isfinite(tn)) {
                c = __builtin_nanf("");
                if (blockIdx.x == 0) dev_err_raise(dev_err, DP_DEVERR_NONFINITE_GRAD);
            }
            coef_s = c;
            if (blockIdx.x == 0 && total_norm_out) total_norm_out[0] = tn;
        }
    }
    __syncthreads();
    const float coef = coef_s;
    if (coef != coef) return;          // non-finite gradient norm: no update
    if (step_counter) {
        step_size = step_s[0];
        inv_bc2_sqrt = step_s[1];
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        g[i] = gi;                     // clip_grad_norm_ scales .grad in place
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
    }
}

void clip_adam_step(Seq& q, float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float max_norm,
                    float beta1, float beta2, float eps, float step_size, float inv_bc2_sqrt, float* total_norm_out,
                    int* step_counter, float lr) {
    float* partial = q.alloc<float>(OPT_WGS);
    if (!q.ok() || n <= 0) return;
    long want = (n + 255) / 256;
    const int wgs = (int)(want < OPT_WGS ? want : OPT_WGS);
    const bool need_norm = max_norm > 0.f || total_norm_out;
    if (need_norm || step_counter) {
        hipLaunchKernelGGL(k_sqnorm_partials, dim3(wgs), dim3(256), 0, q.stream, grads, n, partial, step_counter);
        q.check_launch("sqnorm_partials");
    }
    hipLaunchKernelGGL(k_clip_adam, dim3(wgs), dim3(256), 0, q.stream, params, grads, exp_avg, exp_avg_sq, n,
                       need_norm ? partial : (const float*)nullptr, wgs, max_norm, beta1, beta2, eps, step_size,
                       inv_bc2_sqrt, total_norm_out, device_error_word(), step_counter, lr);
    q.check_launch("clip_adam");
}

}  // namespace dp
