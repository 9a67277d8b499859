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
    // 16-byte accesses where the chunk allows (chunks start on 256-byte slots of the flat gradient buckets); the sum of squares is
    // accumulated per thread in the element order i, i + 1024, ... either way -- the same partial sums as the scalar loop's lanes
    // would NOT come out, so the vector path keeps its own fixed order: deterministic run to run, like the scalar one
    const bool vec = ((size_t)c.grad & 15) == 0 && (c.n & 3) == 0;
    if (vec) {
        float4* g4 = reinterpret_cast<float4*>(c.grad);
        for (int i = threadIdx.x; i < (c.n >> 2); i += 256) {
            float4 g = g4[i];
            if (clip_value > 0.f) {                   // torch.nn.utils.clip_grad_value_: clamp_(-v, v) (NaN stays NaN)
                g.x = g.x < -clip_value ? -clip_value : (g.x > clip_value ? clip_value : g.x);
                g.y = g.y < -clip_value ? -clip_value : (g.y > clip_value ? clip_value : g.y);
                g.z = g.z < -clip_value ? -clip_value : (g.z > clip_value ? clip_value : g.z);
                g.w = g.w < -clip_value ? -clip_value : (g.w > clip_value ? clip_value : g.w);
                g4[i] = g;
            }
            ss += (double)g.x * (double)g.x + (double)g.y * (double)g.y + (double)g.z * (double)g.z + (double)g.w * (double)g.w;
        }
    } else {
        for (int i = threadIdx.x; i < c.n; i += 256) {
            float g = c.grad[i];
            if (clip_value > 0.f) {
                g = g < -clip_value ? -clip_value : (g > clip_value ? clip_value : g);
                c.grad[i] = g;
            }
            ss += (double)g * (double)g;
        }
    }
    const double tot = block_sum<256>(ss, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// kind 0: Adam (no amsgrad), 1: Adamax.  step = number of this update (1-based).
__global__ void __launch_bounds__(256) k_optim_update(const glowhip_optim_chunk* __restrict__ chunks, int n_chunks,
                                                      const double* __restrict__ partial, int kind, float lr, float beta1,
                                                      float beta2, float omb1, float omb2, float eps, float weight_decay, double bc1, double bc2,
                                                      float max_norm, float* __restrict__ grad_norm_out, int skip_if_nonfinite,
                                                      const double* __restrict__ hyper) {
    // hyper != null (glowhip_optim_step_dev: the launch is part of a captured graph, whose kernel arguments are frozen): the values
    // that change from step to step -- {lr as float, 1 - beta1^step, 1 - beta2^step} -- are read from device memory instead
    if (hyper) { lr = (float)hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }
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
    auto elem = [&](float& g, float& p, float& m, float& v) {
        if (max_norm > 0.f) g = g * coef;             // clip_grad_norm_ multiplies every gradient by the clamped coefficient
        float ge = g;
        if (weight_decay != 0.f) ge = ge + weight_decay * p;
        m = m + (ge - m) * omb1;                      // exp_avg.lerp_(grad, 1 - beta1); omb1 = float(1 - beta1 in double), as torch
        if (kind == 0) {
            v = v * beta2 + omb2 * ge * ge;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            const float denom = sqrtf(v) / bc2_sqrt + eps;
            p = p - step_size * (m / denom);          // param.addcdiv_(exp_avg, denom, value=-step_size)
        } else {
            const float a = v * beta2, b = fabsf(ge) + eps;       // exp_inf
            v = a > b ? a : b;
            p = p - step_size * (m / v);              // clr = lr / bias_correction1
        }
    };
    const bool vec = (c.n & 3) == 0 && (((size_t)c.grad | (size_t)c.param | (size_t)c.m | (size_t)c.v) & 15) == 0;
    if (vec) {      // the same arithmetic per element, four elements per 16-byte access
        float4* g4 = reinterpret_cast<float4*>(c.grad); float4* p4 = reinterpret_cast<float4*>(c.param);
        float4* m4 = reinterpret_cast<float4*>(c.m); float4* v4 = reinterpret_cast<float4*>(c.v);
        for (int i = threadIdx.x; i < (c.n >> 2); i += 256) {
            float4 g = g4[i], p = p4[i], m = m4[i], v = v4[i];
            elem(g.x, p.x, m.x, v.x); elem(g.y, p.y, m.y, v.y); elem(g.z, p.z, m.z, v.z); elem(g.w, p.w, m.w, v.w);
            if (max_norm > 0.f) g4[i] = g;
            p4[i] = p; m4[i] = m; v4[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < c.n; i += 256) {
            float g = c.grad[i], p = c.param[i], m = c.m[i], v = c.v[i];
            elem(g, p, m, v);
            if (max_norm > 0.f) c.grad[i] = g;
            c.param[i] = p; c.m[i] = m; c.v[i] = v;
        }
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
                       (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), eps, weight_decay, bc1, bc2, max_norm, grad_norm_out, skip_if_nonfinite,
                       (const double*)nullptr);
    GH_LAUNCH_CHECK("k_optim_update");
    return GLOWHIP_OK;
}

extern "C" int glowhip_optim_step_dev(const glowhip_optim_chunk* chunks_dev, int n_chunks, int kind, const double* hyper_dev, double beta1_d,
                                      double beta2_d, float eps, float weight_decay, float clip_value, float max_norm,
                                      double* partial_dev, float* grad_norm_out, int skip_if_nonfinite, glowhip_stream_t stream) {
    GH_REQUIRE(chunks_dev && partial_dev && hyper_dev, "optim_step_dev: null argument");
    GH_REQUIRE(kind == 0 || kind == 1, "optim_step_dev: kind %d (0 = adam, 1 = adamax)", kind);
    if (n_chunks == 0) return GLOWHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_optim_clip_sumsq, dim3(n_chunks), dim3(256), 0, s, chunks_dev, clip_value, partial_dev);
    GH_LAUNCH_CHECK("k_optim_clip_sumsq");
    hipLaunchKernelGGL(k_optim_update, dim3(n_chunks), dim3(256), 0, s, chunks_dev, n_chunks, partial_dev, kind, 0.f, (float)beta1_d, (float)beta2_d,
                       (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), eps, weight_decay, 1.0, 1.0, max_norm, grad_norm_out, skip_if_nonfinite,
                       hyper_dev);
    GH_LAUNCH_CHECK("k_optim_update");
    return GLOWHIP_OK;
}
