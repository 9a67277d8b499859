// optim.hip -- the optimiser step of the reference's training loop (network/trainer.py:142-150; optimisers built by
// network/builder.py:108-113: torch.optim.Adam / Adamax) as TWO launches over all ~1 060 parameter tensors instead of
// ~5 foreach-kernels x several passes: (1) clip_grad_value_ in place + per-chunk sum of squares, (2) clip_grad_norm_ coefficient
// from the partial sums (fixed order: deterministic) + the Adam / Adamax update.  Element-wise arithmetic follows torch's
// single-tensor implementations operation for operation (lerp for exp_avg, addcmul for exp_avg_sq, addcdiv for the update).
#include <math.h>

#include "common.h"

namespace glowhip {

__global__ void __launch_bounds__(256) k_optim_clip_sumsq(const glowhip_optim_chunk* __restrict__ chunks, float clip_value,
                                                          double* __restrict__ partial) {
    __shared__ double red[4];
    const glowhip_optim_chunk c = chunks[blockIdx.x];
    double ss = 0.0;
    for (int i = threadIdx.x; i < c.n; i += 256) {
        float g = c.grad[i];
        if (clip_value > 0.f) {                       // torch.nn.utils.clip_grad_value_: clamp_(-v, v) (NaN stays NaN)
            g = g < -clip_value ? -clip_value : (g > clip_value ? clip_value : g);
            c.grad[i] = g;
        }
        ss += (double)g * (double)g;
    }
    const double tot = block_sum<256>(ss, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// kind 0: Adam (no amsgrad), 1: Adamax.  step = number of this update (1-based).
__global__ void __launch_bounds__(256) k_optim_update(const glowhip_optim_chunk* __restrict__ chunks, int n_chunks,
                                                      const double* __restrict__ partial, int kind, float lr, float beta1,
                                                      float beta2, float omb1, float omb2, float eps, float weight_decay, double bc1, double bc2,
                                                      float max_norm, float* __restrict__ grad_norm_out, int skip_if_nonfinite) {
    __shared__ float s_coef;
    __shared__ int s_skip;
    if (threadIdx.x < 64) {       // total gradient norm: the chunk sums in a fixed order (wave 0), then clip_grad_norm_'s coefficient
        double t = 0.0;
        for (int i = threadIdx.x; i < n_chunks; i += 64) t += partial[i];
        t = wave_sum(t);
        if (threadIdx.x == 0) {
            const float norm = (float)sqrt(t);
            float coef = 1.f;
            if (max_norm > 0.f) {
                coef = max_norm / (norm + 1e-6f);
                coef = coef > 1.f ? 1.f : coef;
            }
            s_coef = coef;
            s_skip = skip_if_nonfinite && !isfinite(norm);
            if (blockIdx.x == 0 && grad_norm_out) grad_norm_out[0] = norm;
        }
    }
    __syncthreads();
    const float coef = s_coef;
    if (s_skip) return;      // a non-finite gradient somewhere: parameters, state and gradients stay as they are (the caller
                             // reads grad_norm_out later and decides -- no host sync here)
    const glowhip_optim_chunk c = chunks[blockIdx.x];
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    for (int i = threadIdx.x; i < c.n; i += 256) {
        float g = c.grad[i];
        if (max_norm > 0.f) {                         // clip_grad_norm_ multiplies every gradient by the clamped coefficient
            g = g * coef;
            c.grad[i] = g;
        }
        float p = c.param[i];
        if (weight_decay != 0.f) g = g + weight_decay * p;
        float m = c.m[i];
        m = m + (g - m) * omb1;                       // exp_avg.lerp_(grad, 1 - beta1); omb1 = float(1 - beta1 in double), as torch
        c.m[i] = m;
        if (kind == 0) {
            float v = c.v[i];
            v = v * beta2 + omb2 * g * g;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            c.v[i] = v;
            const float denom = sqrtf(v) / bc2_sqrt + eps;
            p = p - step_size * (m / denom);          // param.addcdiv_(exp_avg, denom, value=-step_size)
        } else {
            float u = c.v[i];                         // exp_inf
            const float a = u * beta2, b = fabsf(g) + eps;
            u = a > b ? a : b;
            c.v[i] = u;
            p = p - step_size * (m / u);              // clr = lr / bias_correction1
        }
        c.param[i] = p;
    }
}

}  // namespace glowhip

using namespace glowhip;

extern "C" int glowhip_optim_step(const glowhip_optim_chunk* chunks_dev, int n_chunks, int kind, float lr, double beta1_d,
                                  double beta2_d, float eps, float weight_decay, int step, float clip_value, float max_norm,
                                  double* partial_dev, float* grad_norm_out, int skip_if_nonfinite, glowhip_stream_t stream) {
    GH_REQUIRE(chunks_dev && partial_dev, "optim_step: null argument");
    GH_REQUIRE(kind == 0 || kind == 1, "optim_step: kind %d (0 = adam, 1 = adamax)", kind);
    GH_REQUIRE(step >= 1, "optim_step: step must be >= 1");
    if (n_chunks == 0) return GLOWHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_optim_clip_sumsq, dim3(n_chunks), dim3(256), 0, s, chunks_dev, clip_value, partial_dev);
    GH_LAUNCH_CHECK("k_optim_clip_sumsq");
    const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
    const double bc1 = 1.0 - pow(beta1_d, (double)step), bc2 = 1.0 - pow(beta2_d, (double)step);     // python-double arithmetic, as torch
    hipLaunchKernelGGL(k_optim_update, dim3(n_chunks), dim3(256), 0, s, chunks_dev, n_chunks, partial_dev, kind, lr, beta1, beta2,
                       (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), eps, weight_decay, bc1, bc2, max_norm, grad_norm_out, skip_if_nonfinite);
    GH_LAUNCH_CHECK("k_optim_update");
    return GLOWHIP_OK;
}
