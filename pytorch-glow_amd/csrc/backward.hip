// backward.hip -- gradient kernels of the flow path (training step).
//
// The training forward keeps every layer output and the coupling network's hidden activations on a tape in HBM
// (11 GB at B=64 for the celeba64 model -- the 288 GB part is what makes this the cheap option; recomputing f()
// would cost the whole forward again), so the backward is a straight reverse sweep without inverses.
//
// This file holds the HBM-bound pieces and the shape-generic (direct) convolution gradients:
//   k_coupling_bwd   affine/additive coupling tail + Conv2dZeros scale/bias gradients
//   k_split_bwd      Split2d Gaussian log-density gradients (+ Conv2dZeros scale/bias gradients)
//   k_act_bwd        ReLU mask + ActNorm scale of a hidden layer, with the ActNorm parameter gradients
//   k_chanmix_bwd    ActNorm + invertible 1x1 conv: input gradient, dW (incl. d log|det W|), d bias, d logs
//   k_prior_bwd      top prior: dL/dz of the Gaussian log-density
//   k_wgrad_direct   dW of a convolution, one workgroup per (out, in) channel pair (generic / cross-check)
//   k_weight_flipT   w[o][i][tap] -> wT[i][o][8-tap]: the input gradient of a convolution is a convolution with wT
// Parameter gradients that are reductions over all pixels are accumulated with fp64 atomics (order effects are
// below fp32 resolution after the final rounding) and converted by k_grad_finalize.
#include "kernels.h"
#include "backward.h"
#include "wgrad_reduce.h"
#include <type_traits>
#include "sh.h"
#include "cnet_fin.h"

namespace glowhip {

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { atomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------
// Coupling tail backward (forward: network/model.py:105-113).  Grid (pixel blocks, Ch, N).
//   affine : shift = hout[2c], r = hout[2c+1], s = sigmoid(r+2), z2' = (y2+shift)*s, logdet += sum log s
//            g_y2 = g2*s;  g_shift = g2*s;  g_r = (g2*(y2+shift) + gld/s) * s*(1-s)
//   add    : z2' = y2 + hout[c];  g_y2 = g2;  g_h = g2
//   Conv2dZeros: hout = (conv + b)*e  =>  g_pre = g_h*e (gradient of conv output and of b), g_logs = 3*sum g_h*hout
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_coupling_bwd(CouplingBwdArgs a) {
    __shared__ double red[4];
    const int c = blockIdx.y;
    const long n = blockIdx.z;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool ok = p < a.HW;
    const int nsub = a.affine ? 2 : 1;
    double sb[2] = {0.0, 0.0}, sl[2] = {0.0, 0.0};
    if (ok) {
        const float g2 = a.g2[n * a.g_bs + (long)c * a.HW + p];
        float gh[2];
        int oc[2];
        if (a.affine) {
            oc[0] = 2 * c; oc[1] = 2 * c + 1;
            const float r = a.hout[(n * a.Cout + oc[1]) * a.HW + p];
            const float s = sigmoidf_(r + 2.0f);
            const float z2p = a.z2out[n * a.z_bs + (long)c * a.HW + p];
            const float y2s = z2p / s;                 // = y2 + shift
            a.gy2[n * a.g_bs + (long)c * a.HW + p] = g2 * s;
            gh[0] = g2 * s;
            gh[1] = (g2 * y2s + a.gld[n] / s) * s * (1.0f - s);
        } else {
            oc[0] = c; oc[1] = c;
            a.gy2[n * a.g_bs + (long)c * a.HW + p] = g2;
            gh[0] = g2; gh[1] = 0.f;
        }
        for (int k = 0; k < nsub; ++k) {
            const float e = a.e4[oc[k]];
            const float gpre = gh[k] * e;
            a.gpre[(n * a.Cout + oc[k]) * a.HW + p] = gpre;
            sb[k] = (double)gpre;
            sl[k] = (double)(gh[k] * a.hout[(n * a.Cout + oc[k]) * a.HW + p]) * 3.0;
        }
    }
    for (int k = 0; k < nsub; ++k) {
        const double tb = block_sum<256>(sb[k], red);
        const double tl = block_sum<256>(sl[k], red);
        if (threadIdx.x == 0) {
            const int o = a.affine ? 2 * c + k : c;
            atomic_add_f64(a.acc_b + o, tb);
            atomic_add_f64(a.acc_l + o, tl);
        }
    }
}

// one wave per row: out[r] = 3 (<w[r], dw[r]> + b[r] db[r]) in fp64 (backward.h LogsJob)
__global__ void __launch_bounds__(256) k_logs_from_dw(const LogsJob* __restrict__ jobs) {
    const LogsJob j = jobs[blockIdx.y];
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= j.rows) return;
    const float* w = j.w + (long)r * j.K;
    const float* dw = j.dw + (long)r * j.K;
    double s = 0.0;
    for (int k = lane; k < j.K; k += 64) s += (double)w[k] * (double)dw[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) j.out[r] = (float)(3.0 * (s + (double)j.b[r] * j.db[r]));
}

int launch_logs_from_dw_batched(const LogsJob* jobs_dev, int n_jobs, hipStream_t s) {
    if (n_jobs == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_logs_from_dw, dim3(128, n_jobs), dim3(256), 0, s, jobs_dev);      // (rows <= 512)
    GH_LAUNCH_CHECK("k_logs_from_dw");
    return GLOWHIP_OK;
}

// The same, four consecutive pixels per thread (16-byte accesses) and no LDS: a workgroup takes 1024 consecutive elements of one
// image's (Ch, HW) plane -- HW / 4 threads per channel, a power of two -- sums per channel in fp32 inside the wave (or inside the
// channel's lane segment when a wave spans several channels), one fp64 atomic per wave (segment), channel and quantity.  The
// one-pixel-per-thread kernel above (four fp64 block reductions, eight barriers, 1536 workgroups of one wave's worth of work
// each) took 22 us per launch at any level.
__global__ void __launch_bounds__(256) k_coupling_bwd4(CouplingBwdArgs a) {
    const long n = blockIdx.y;
    const int HW = a.HW, tpc = HW >> 2;
    const long off = (long)blockIdx.x * 1024 + threadIdx.x * 4;          // element of the (Ch, HW) plane
    const bool ok = off < (long)a.Ch * HW;
    const int c = ok ? (int)(off / HW) : 0;
    const int p = ok ? (int)(off - (long)c * HW) : 0;
    float sb[2] = {0.f, 0.f}, sl[2] = {0.f, 0.f};
    const int oc0 = a.affine ? 2 * c : c, oc1 = a.affine ? 2 * c + 1 : c;
    if (ok) {
        const float4 g2 = *reinterpret_cast<const float4*>(a.g2 + n * a.g_bs + (long)c * HW + p);
        const float g2v[4] = {g2.x, g2.y, g2.z, g2.w};
        float gy[4], gp0[4], gp1[4];
        if (a.affine) {
            const float4 h0 = *reinterpret_cast<const float4*>(a.hout + (n * a.Cout + oc0) * HW + p);
            const float4 h1 = *reinterpret_cast<const float4*>(a.hout + (n * a.Cout + oc1) * HW + p);
            const float4 zz = *reinterpret_cast<const float4*>(a.z2out + n * a.z_bs + (long)c * HW + p);
            const float h0v[4] = {h0.x, h0.y, h0.z, h0.w}, h1v[4] = {h1.x, h1.y, h1.z, h1.w}, zv[4] = {zz.x, zz.y, zz.z, zz.w};
            const float e0 = a.e4[oc0], e1 = a.e4[oc1], gld = a.gld[n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sg = sigmoidf_(h1v[j] + 2.0f);
                const float y2s = zv[j] / sg;                 // = y2 + shift
                gy[j] = g2v[j] * sg;
                const float gh0 = g2v[j] * sg;
                const float gh1 = (g2v[j] * y2s + gld / sg) * sg * (1.0f - sg);
                gp0[j] = gh0 * e0; gp1[j] = gh1 * e1;
                sb[0] += gp0[j]; sb[1] += gp1[j];
                sl[0] += gh0 * h0v[j]; sl[1] += gh1 * h1v[j];
            }
            *reinterpret_cast<float4*>(a.gpre + (n * a.Cout + oc1) * HW + p) = make_float4(gp1[0], gp1[1], gp1[2], gp1[3]);
        } else {
            const float4 h0 = *reinterpret_cast<const float4*>(a.hout + (n * a.Cout + oc0) * HW + p);
            const float h0v[4] = {h0.x, h0.y, h0.z, h0.w};
            const float e0 = a.e4[oc0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gy[j] = g2v[j];
                gp0[j] = g2v[j] * e0;
                sb[0] += gp0[j];
                sl[0] += g2v[j] * h0v[j];
            }
        }
        *reinterpret_cast<float4*>(a.gy2 + n * a.g_bs + (long)c * HW + p) = make_float4(gy[0], gy[1], gy[2], gy[3]);
        *reinterpret_cast<float4*>(a.gpre + (n * a.Cout + oc0) * HW + p) = make_float4(gp0[0], gp0[1], gp0[2], gp0[3]);
    }
    // per-channel sums: segments of min(tpc, 64) consecutive lanes share a channel
    const int seg = tpc < 64 ? tpc : 64;
    const int nsub = a.affine ? 2 : 1;
    for (int k = 0; k < nsub; ++k) {
        float vb = sb[k], vl = sl[k];
        for (int o = seg >> 1; o > 0; o >>= 1) { vb += __shfl_down(vb, o, 64); vl += __shfl_down(vl, o, 64); }
        if ((threadIdx.x & (seg - 1)) == 0 && ok) {
            // (6 144 atomics on the two cache lines of a 12-channel level took 30 us: the copies spread them over 16 x as many)
            const long aco = (long)((blockIdx.x + blockIdx.y * gridDim.x + (threadIdx.x >> 6)) % a.acc_copies) * a.acc_stride;
            const int o = k == 0 ? oc0 : oc1;
            atomic_add_f64(a.acc_b + aco + o, (double)vb);
            atomic_add_f64(a.acc_l + aco + o, 3.0 * (double)vl);
        }
    }
}

int launch_coupling_bwd(const CouplingBwdArgs& a, hipStream_t s) {
    if (a.N == 0) return GLOWHIP_OK;
    const bool pow2 = a.HW >= 4 && (a.HW & (a.HW - 1)) == 0;
    const bool aligned = a.g_bs % 4 == 0 && a.z_bs % 4 == 0 && ((size_t)a.g2 & 15) == 0 && ((size_t)a.gy2 & 15) == 0 &&
                         ((size_t)a.z2out & 15) == 0 && ((size_t)a.hout & 15) == 0 && ((size_t)a.gpre & 15) == 0;
    if (pow2 && aligned) {
        hipLaunchKernelGGL(k_coupling_bwd4, dim3((unsigned)(((long)a.Ch * a.HW + 1023) / 1024), a.N), dim3(256), 0, s, a);
        GH_LAUNCH_CHECK("k_coupling_bwd4");
        return GLOWHIP_OK;
    }
    hipLaunchKernelGGL(k_coupling_bwd, dim3(cdiv(a.HW, 256), a.Ch, a.N), dim3(256), 0, s, a);
    GH_LAUNCH_CHECK("k_coupling_bwd");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Split2d backward (forward: network/module.py:526-530): logdet += sum logp(z2 | mean, logs), (mean, logs) =
// even/odd channels of hout = Conv2dZeros(z1).  With d = z2 - mean, q = exp(-2 logs):
//   g_z2 = gld * (-d*q) (+ incoming gradient of z2 is zero: z2 is dropped),  g_mean = gld * d*q,
//   g_logs = gld * (d*d*q - 1).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_split_bwd(SplitBwdArgs a) {
    __shared__ double red[4];
    const int c = blockIdx.y;
    const long n = blockIdx.z;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool ok = p < a.HW;
    double sb[2] = {0.0, 0.0}, sl[2] = {0.0, 0.0};
    if (ok) {
        const int Cout = 2 * a.Ch;
        const float mean = a.hout[(n * Cout + 2 * c) * a.HW + p];
        const float logs = a.hout[(n * Cout + 2 * c + 1) * a.HW + p];
        const float z2 = a.z2[n * a.z_bs + (long)c * a.HW + p];
        const float d = z2 - mean, q = expf(-2.0f * logs);
        const float gld = a.gld[n];
        a.gz2[n * a.g_bs + (long)c * a.HW + p] = -gld * d * q;
        const float gh[2] = {gld * d * q, gld * (d * d * q - 1.0f)};
        for (int k = 0; k < 2; ++k) {
            const int o = 2 * c + k;
            const float gpre = gh[k] * a.e4[o];
            a.gpre[(n * Cout + o) * a.HW + p] = gpre;
            sb[k] = (double)gpre;
            sl[k] = (double)(gh[k] * a.hout[(n * Cout + o) * a.HW + p]) * 3.0;
        }
    }
    for (int k = 0; k < 2; ++k) {
        const double tb = block_sum<256>(sb[k], red);
        const double tl = block_sum<256>(sl[k], red);
        if (threadIdx.x == 0) {
            atomic_add_f64(a.acc_b + 2 * c + k, tb);
            atomic_add_f64(a.acc_l + 2 * c + k, tl);
        }
    }
}

int launch_split_bwd(const SplitBwdArgs& a, hipStream_t s) {
    if (a.N == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_split_bwd, dim3(cdiv(a.HW, 256), a.Ch, a.N), dim3(256), 0, s, a);
    GH_LAUNCH_CHECK("k_split_bwd");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Hidden activation backward: h = relu((u + b)*e).  Given g_h (raw) and the saved h:
//   g_u = g_h * (h > 0) * e;   g_b = sum g_u;   g_logs = 3 * sum g_h*h   (h = (u+b)*e wherever the mask is 1)
// In place on g (g_h -> g_u).  Grid (pixel blocks, channels, N).
// ------------------------------------------------------------------------------------------------
// One workgroup = 1024 consecutive elements of one image (= 1024 / HW whole channels, or a slice of one), four per thread as one
// 16-byte access; per-channel sums in fp32 inside the segment (<= 1024 terms), one fp64 atomic pair per channel segment.
// (The first version had one element per thread and two fp64 block reductions per 256 elements: 2.8 TB/s.)
__global__ void __launch_bounds__(256) k_act_bwd(float* __restrict__ g, const float* __restrict__ h,
                                                 const float* __restrict__ e, int Cm, int HW, long per_img,
                                                 double* __restrict__ acc_b, double* __restrict__ acc_l) {
    __shared__ float red[2][4];
    const long n = blockIdx.y;
    const long off = (long)blockIdx.x * 1024 + threadIdx.x * 4;          // element offset inside the image
    const int tpc = HW >> 2;                                             // threads per channel (HW % 4 == 0)
    float sb = 0.f, sl = 0.f;
    int c = 0;
    if (off < per_img) {
        c = (int)(off / HW);
        const long idx = n * per_img + off;
        const float4 hv = *reinterpret_cast<const float4*>(h + idx);
        float4 gh = *reinterpret_cast<const float4*>(g + idx);
        const float ec = e[c];
        sl = (gh.x * hv.x + gh.y * hv.y) + (gh.z * hv.z + gh.w * hv.w);
        gh.x = hv.x > 0.f ? gh.x * ec : 0.f; gh.y = hv.y > 0.f ? gh.y * ec : 0.f;
        gh.z = hv.z > 0.f ? gh.z * ec : 0.f; gh.w = hv.w > 0.f ? gh.w * ec : 0.f;
        *reinterpret_cast<float4*>(g + idx) = gh;
        sb = (gh.x + gh.y) + (gh.z + gh.w);
    }
    if (tpc >= 256) {            // the whole workgroup sits inside one channel
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sb += __shfl_down(sb, o, 64); sl += __shfl_down(sl, o, 64); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sb; red[1][threadIdx.x >> 6] = sl; }
        __syncthreads();
        if (threadIdx.x == 0 && off < per_img) {
            atomic_add_f64(acc_b + c, (double)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])));
            atomic_add_f64(acc_l + c, 3.0 * (double)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])));
        }
        return;
    }
    // tpc in {1, 2, 4, ..., 128}: segments of tpc consecutive threads
    const int w = tpc < 64 ? tpc : 64;
    for (int o = w >> 1; o > 0; o >>= 1) { sb += __shfl_down(sb, o, 64); sl += __shfl_down(sl, o, 64); }
    if (tpc == 128) {            // two waves per channel
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sb; red[1][threadIdx.x >> 6] = sl; }
        __syncthreads();
        if ((threadIdx.x & 127) == 0 && off < per_img) {
            const int wv = threadIdx.x >> 6;
            atomic_add_f64(acc_b + c, (double)(red[0][wv] + red[0][wv + 1]));
            atomic_add_f64(acc_l + c, 3.0 * (double)(red[1][wv] + red[1][wv + 1]));
        }
        return;
    }
    if ((threadIdx.x & (tpc - 1)) == 0 && off < per_img) {
        atomic_add_f64(acc_b + c, (double)sb);
        atomic_add_f64(acc_l + c, 3.0 * (double)sl);
    }
}

// element-per-thread fall-back for shapes the vector kernel does not take (HW not a power of two >= 4)
__global__ void __launch_bounds__(256) k_act_bwd1(float* __restrict__ g, const float* __restrict__ h,
                                                  const float* __restrict__ e, int Cm, int HW,
                                                  double* __restrict__ acc_b, double* __restrict__ acc_l) {
    __shared__ double red[4];
    const int c = blockIdx.y;
    const long n = blockIdx.z;
    const int p = blockIdx.x * 256 + threadIdx.x;
    double sb = 0.0, sl = 0.0;
    if (p < HW) {
        const long idx = (n * Cm + c) * HW + p;
        const float hv = h[idx], gh = g[idx];
        const float gu = hv > 0.f ? gh * e[c] : 0.f;
        g[idx] = gu;
        sb = (double)gu;
        sl = (double)(gh * hv) * 3.0;
    }
    const double tb = block_sum<256>(sb, red);
    const double tl = block_sum<256>(sl, red);
    if (threadIdx.x == 0) {
        atomic_add_f64(acc_b + c, tb);
        atomic_add_f64(acc_l + c, tl);
    }
}

// The hidden activations as the taping k_cnet stores them -- fp16, pixel-tile-major [pixel / 32][R][pixel % 32] over the batch's
// N * HW pixels (sh.h) -- as the fp32 (N, R, HW) tensor the per-layer backward kernels read.  HW % 32 == 0.
__global__ void __launch_bounds__(256) k_half_to_float(const _Float16* __restrict__ src, float* __restrict__ dst, long n, int R, int HW) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;       // 8 consecutive pixels of one row of one pixel tile
    if (i >= n) return;
    const h8 v = *reinterpret_cast<const h8*>(src + i);
    const long tr = i >> 5;                                           // tile * R + row
    const long tile = tr / R, row = tr - tile * R, pix = tile * 32 + (i & 31);
    const long img = pix / HW, p = pix - img * HW;
    float* d = dst + (img * R + row) * HW + p;
    const f32x4_t a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    *reinterpret_cast<f32x4_t*>(d) = a;
    *reinterpret_cast<f32x4_t*>(d + 4) = b;
}

int launch_half_to_float(const void* src_half, float* dst, int N, int R, int HW, hipStream_t s) {
    const long n = (long)N * R * HW;
    if (n <= 0) return GLOWHIP_OK;
    GH_REQUIRE(HW % 32 == 0, "half_to_float: %d pixels per image are not whole 32-pixel tiles of the fp16 tape", HW);
    hipLaunchKernelGGL(k_half_to_float, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s, (const _Float16*)src_half, dst, n, R, HW);
    GH_LAUNCH_CHECK("k_half_to_float");
    return GLOWHIP_OK;
}

int launch_act_bwd(float* g, const float* h, const float* e, int N, int Cm, int HW, double* acc_b, double* acc_l,
                   hipStream_t s) {
    if (N == 0) return GLOWHIP_OK;
    const bool pow2 = HW >= 4 && (HW & (HW - 1)) == 0;
    if (pow2) {
        const long per_img = (long)Cm * HW;
        hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)((per_img + 1023) / 1024), N), dim3(256), 0, s, g, h, e, Cm, HW, per_img, acc_b, acc_l);
    } else {
        hipLaunchKernelGGL(k_act_bwd1, dim3(cdiv(HW, 256), Cm, N), dim3(256), 0, s, g, h, e, Cm, HW, acc_b, acc_l);
    }
    GH_LAUNCH_CHECK("k_act_bwd");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// ActNorm + invertible 1x1 conv / permutation backward.  Forward: v = (x + b)*e, y = W v (or y[o] = v[idx[o]]).
//   g_v = W^T g_y (or scatter);  g_x = g_v * e;
//   dW[o][i] = sum_px g_y[o] v[i]  (+ G*HW*W^-1[i][o] added by k_grad_finalize_w, G = sum_n gld[n])
//   g_b[c] = sum g_v[c]*e[c];  g_logs[c] = 3*sum g_v[c]*v[c]  (+ 3*HW*G in finalize)
// 64 pixels per workgroup; x->v, g_y and g_v live in LDS columns [c][px]; the C*C outer-product sums are formed
// pair-by-pair over the 64 pixels and added with one fp64 atomic per pair per workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int CB_PX = 64;
constexpr int CB_LD = CB_PX + 1;    // LDS row stride: the reductions below read one COLUMN q of many rows per instruction -- with a stride of
                                    // 64 words every lane hit the same bank (64-way conflicts: 41 us per launch at C = 48)
// (the kernel body as a device function: k_chanmix_bwd launches it alone, k_chanmix_bwd_reduce beside the split-K reductions of the
// FlowStep's weight-gradient GEMMs; `block` / `nblocks`: this launch part's block index / count)
__device__ __forceinline__ void chanmix_bwd_body(const ChanMixBwdArgs& a, const int block, const int nblocks, float* sm) {
    const int C = a.C;
    float* v = sm;                   // [C][64]
    float* gy = sm + C * CB_LD;      // [C][64 (+1)]
    float* gv = gy + C * CB_LD;      // [C][64 (+1)]
    // W in LDS where it fits (launch_chanmix_bwd: a.w_lds): the g_v loop below read it from memory eight elements per trip, C / 8
    // trips per output channel, twelve channels per thread at C = 48 -- 72 round trips to the L2 in a row, most of the launch's 39 us
    float* wl = gv + C * CB_LD;      // [C][C]
    if (a.matrix && a.w_lds) {
        for (int e0 = threadIdx.x * 4; e0 < C * C; e0 += 1024) {
            const f32x4_t w4 = *reinterpret_cast<const f32x4_t*>(a.matrix + e0);       // (C * C is a multiple of 4: C is even)
            *reinterpret_cast<f32x4_t*>(wl + e0) = w4;
        }
    }
    const int px = threadIdx.x & (CB_PX - 1);
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the matrix element W[o][i] of the g_v loop
                                                                           // below is then a scalar load (it was a vector load per
                                                                           // FMA: 41 us per launch at C = 48)
    const long gp = (long)block * CB_PX + px;
    const long total = (long)a.N * a.HW;
    const bool valid = gp < total;
    const long n = valid ? gp / a.HW : 0;
    const int p = valid ? (int)(gp - n * a.HW) : 0;
    FinSrc addf{};
    if (a.add_part) {
        CnetPending pd{};
        pd.scratch = a.add_part; pd.MS = a.add_MS; pd.tiles = a.add_tiles; pd.R = a.add_R; pd.NI = a.add_NI; pd.lpxt = a.add_lpxt;
        pd.mode = TAIL_ADD_FWD; pd.Cout = a.add_C;
        addf = fin_src(pd, a.N, a.add_H, a.add_W, a.HW, __builtin_ctz(a.add_W));
    }
    // Staging, four channels of the thread at a time: every load of the four -- x, g_y, and for the first add_C channels the MS
    // partial sums + 2 MS halo rows of the backward k_cnet launch (cnet_fin.h; the row split as a template argument unrolls them) --
    // is unconditional from a clamped address and issued before the first LDS store.  One channel per iteration with the gather
    // behind a branch was 2 round trips per channel: 24 in a row at C = 48, most of that launch's 37 us.
    auto stage = [&](auto msv, auto halo) {
        constexpr int MSV = decltype(msv)::value, U = 4;
        constexpr bool HALO = decltype(halo)::value;      // (false: whole-image tiles -- no halo rows to gather: 4 instead of 20 loads per channel at MS = 4)
        const long nn = valid ? n : 0;
        const int pp = valid ? p : 0;
        for (int c0 = grp; c0 < C; c0 += 4 * U) {
            float xr[U], gr[U], se[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = min(c0 + 4 * u, C - 1);
                xr[u] = a.x[nn * a.x_bs + (long)c * a.HW + pp];
                gr[u] = a.gy[nn * a.g_bs + (long)c * a.HW + pp];
                se[u] = 0.f;
            }
            if (a.add_part) {      // (uniform)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float so;
                    fin_gather_t<MSV, HALO>(addf, nn, min(c0 + 4 * u, a.add_C - 1), pp, se[u], so);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 4 * u;
                if (c < C) {
                    const float g = gr[u] + ((a.add_part && c < a.add_C) ? se[u] * a.add_scale : 0.f);
                    v[c * CB_LD + px] = valid ? (xr[u] + a.bias[c]) * a.scale[c] : 0.f;
                    gy[c * CB_LD + px] = valid ? g : 0.f;
                }
            }
        }
    };
    const bool halos = !a.add_part || addf.halos;
    if (a.add_MS == 1 && halos) stage(std::integral_constant<int, 1>{}, std::true_type{});
    else if (a.add_MS == 2 && halos) stage(std::integral_constant<int, 2>{}, std::true_type{});
    else if (a.add_MS == 4 && halos) stage(std::integral_constant<int, 4>{}, std::true_type{});
    else if (a.add_MS == 4) stage(std::integral_constant<int, 4>{}, std::false_type{});
    else if (a.add_MS == 2) stage(std::integral_constant<int, 2>{}, std::false_type{});
    else stage(std::integral_constant<int, 0>{}, std::true_type{});
    __syncthreads();
    // g_v = W^T g_y : g_v[i] = sum_o W[o][i] g_y[o]   (gather: g_v[idx[o]] = g_y[o])
    // (W in LDS and C a multiple of 4: four CONSECUTIVE output channels per thread -- one 16-byte broadcast read of W[o][i .. i+3]
    // and one read of g_y[o] for four FMAs instead of two reads per FMA; the phase was 10 of the C = 48 launch's 31 us)
    const bool blk4 = a.matrix && a.w_lds && (C & 3) == 0;
    for (int i0 = 4 * grp; blk4 && i0 < C; i0 += 16) {
        f32x4_t r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int o = 0; o < C; ++o) {
            const f32x4_t w4 = *reinterpret_cast<const f32x4_t*>(wl + o * C + i0);
            const float g = gy[o * CB_LD + px];
            r[0] = fmaf(w4[0], g, r[0]); r[1] = fmaf(w4[1], g, r[1]); r[2] = fmaf(w4[2], g, r[2]); r[3] = fmaf(w4[3], g, r[3]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gv[(i0 + u) * CB_LD + px] = r[u];
            if (valid) a.gx[n * a.g_bs + (long)(i0 + u) * a.HW + p] = r[u] * a.scale[i0 + u];
        }
    }
    for (int i = grp; i < C && !blk4; i += 4) {
        float r = 0.f;
        if (a.matrix) {
            if (a.w_lds) { for (int o = 0; o < C; ++o) r = fmaf(wl[o * C + i], gy[o * CB_LD + px], r); }
            else { for (int o = 0; o < C; ++o) r = fmaf(a.matrix[o * C + i], gy[o * CB_LD + px], r); }
        } else {
            r = gy[(a.gather_inv ? a.gather_inv[i] : i) * CB_LD + px];
        }
        gv[i * CB_LD + px] = r;
        if (valid) a.gx[n * a.g_bs + (long)i * a.HW + p] = r * a.scale[i];
    }
    __syncthreads();
    // reductions over the 64 pixels of this workgroup
    const int tid = threadIdx.x;
    const long aco = (long)(block % a.acc_copies) * a.acc_stride;      // this workgroup's copy of the accumulators
    const bool own = nblocks <= a.acc_copies;      // ... and nobody else's (zeroed per step): plain stores
    if (a.matrix) {
        // many sums per thread (C >= 32, C even): 2 x 2 blocks of (o, i) -- four LDS reads per four FMAs instead of two per FMA
        // (C = 48: 26.9 -> 24.4 us; at C = 12 the 36 blocks would leave most of the workgroup idle: 10.6 -> 12.9 us)
        const bool blk2 = (C & 1) == 0 && C >= 32;
        const int hc = C >> 1;
        for (int pair = tid; !blk2 && pair < C * C; pair += 256) {
            const int o = pair / C, i = pair - o * C;
            float s = 0.f;
#pragma unroll 8
            for (int q = 0; q < CB_PX; ++q) s = fmaf(gy[o * CB_LD + q], v[i * CB_LD + q], s);
            if (own) a.acc_w[aco + pair] = (double)s; else atomic_add_f64(a.acc_w + aco + pair, (double)s);
        }
        for (int blk = tid; blk2 && blk < hc * hc; blk += 256) {
            const int o = 2 * (blk / hc), i = 2 * (blk % hc);
            float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll 8
            for (int q = 0; q < CB_PX; ++q) {
                const float g0 = gy[o * CB_LD + q], g1 = gy[(o + 1) * CB_LD + q], v0 = v[i * CB_LD + q], v1 = v[(i + 1) * CB_LD + q];
                s00 = fmaf(g0, v0, s00); s01 = fmaf(g0, v1, s01); s10 = fmaf(g1, v0, s10); s11 = fmaf(g1, v1, s11);
            }
            const float sv[4] = {s00, s01, s10, s11};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pair = (o + (u >> 1)) * C + i + (u & 1);
                if (own) a.acc_w[aco + pair] = (double)sv[u]; else atomic_add_f64(a.acc_w + aco + pair, (double)sv[u]);
            }
        }
    }
    for (int c = tid; c < C; c += 256) {
        float sb = 0.f, sl = 0.f;
        for (int q = 0; q < CB_PX; ++q) {
            sb = fmaf(gv[c * CB_LD + q], a.scale[c], sb);
            sl = fmaf(gv[c * CB_LD + q], v[c * CB_LD + q], sl);
        }
        if (own) { a.acc_b[aco + c] = (double)sb; a.acc_l[aco + c] = (double)sl * 3.0; }
        else { atomic_add_f64(a.acc_b + aco + c, (double)sb); atomic_add_f64(a.acc_l + aco + c, (double)sl * 3.0); }
    }
}

__global__ void __launch_bounds__(256) k_chanmix_bwd(ChanMixBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    chanmix_bwd_body(a, blockIdx.x, gridDim.x, sm);
}

// The mixer backward of a FlowStep and the split-K reductions of its (up to three) weight-gradient GEMMs in ONE launch: blocks
// [0, mix_blocks) are k_chanmix_bwd's, then rblocks blocks per reduction job.  The two have nothing to do with each other except
// their place in the sweep -- both short (11 - 19 us and 8 - 15 us), one latency-bound with little data, the other pure
// bandwidth: side by side they take about as long as the longer one.
__global__ void __launch_bounds__(256) k_chanmix_bwd_reduce(ChanMixBwdArgs a, WgradReduceJobs j, int mix_blocks, int rblocks) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x;
    if (b < mix_blocks) { chanmix_bwd_body(a, b, mix_blocks, sm); return; }
    const int rb = b - mix_blocks, job = rb / rblocks;
    const WgradReduceJob& r = j.job[job];
    wgrad_reduce_body(r.partial, r.dw, r.splits, r.Mpad, r.Npad, r.Mreal, r.Nreal, r.mode, rb - job * rblocks);
}

int launch_chanmix_bwd(const ChanMixBwdArgs& a, hipStream_t s, const WgradReduceJobs* reduce) {
    GH_REQUIRE(a.C > 0 && a.C <= 192, "chanmix backward: C=%d unsupported (1..192)", a.C);
    const long total = (long)a.N * a.HW;
    if (total == 0) return reduce ? launch_wgrad_reduce_batched(*reduce, s) : GLOWHIP_OK;
    size_t lds = (size_t)3 * a.C * CB_LD * sizeof(float);
    ChanMixBwdArgs b = a;
    // (C % 4 == 0: the kernel stages W with 16-byte loads / stores at an offset of 3 C 65 floats -- any other C would run up to three
    // floats past W and the LDS block, and C % 4 == 2 would issue misaligned 16-byte LDS accesses: ADVICE r4)
    b.w_lds = a.matrix && (a.C & 3) == 0 && lds + (size_t)a.C * a.C * sizeof(float) <= 64 * 1024 && (reinterpret_cast<uintptr_t>(a.matrix) & 15) == 0;
    if (b.w_lds) lds += (size_t)a.C * a.C * sizeof(float);
    const int mix_blocks = cdiv(total, CB_PX);
    // (the reductions ride along where the mixer's LDS block leaves room for several workgroups per CU: they want occupancy)
    if (reduce && reduce->n > 0 && lds <= 48 * 1024) {
        long rblocks = 1;
        for (int i = 0; i < reduce->n; ++i) rblocks = std::max(rblocks, (long)cdiv((long)reduce->job[i].Mreal * ((reduce->job[i].Nreal + 3) / 4), 256));
        if (lds > 32 * 1024)
            (void)hipFuncSetAttribute((const void*)k_chanmix_bwd_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_chanmix_bwd_reduce, dim3((unsigned)(mix_blocks + rblocks * reduce->n)), dim3(256), lds, s, b, *reduce, mix_blocks, (int)rblocks);
        GH_LAUNCH_CHECK("k_chanmix_bwd_reduce");
        return GLOWHIP_OK;
    }
    if (reduce) GH_TRY(launch_wgrad_reduce_batched(*reduce, s));
    if (lds > 32 * 1024)
        (void)hipFuncSetAttribute((const void*)k_chanmix_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_chanmix_bwd, dim3(mix_blocks), dim3(256), lds, s, b);
    GH_LAUNCH_CHECK("k_chanmix_bwd");
    return GLOWHIP_OK;
}

// top prior: logp = sum -0.5*(log 2pi + 2 logs + (z-mean)^2 e^{-2 logs});  g_z (+)= gld * -(z-mean) e^{-2 logs}
__global__ void __launch_bounds__(256) k_prior_bwd(const float* __restrict__ z, const float* __restrict__ mean,
                                                   const float* __restrict__ logs, long ml_bs,
                                                   const float* __restrict__ gld, const float* __restrict__ gz_in,
                                                   float* __restrict__ gz, long per) {
    const long n = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    const float m = mean ? mean[n * ml_bs + e] : 0.f;
    const float l = logs ? logs[n * ml_bs + e] : 0.f;
    const float g = -gld[n] * (z[n * per + e] - m) * expf(-2.0f * l);
    gz[n * per + e] = g + (gz_in ? gz_in[n * per + e] : 0.f);
}

int launch_prior_bwd(const float* z, const float* mean, const float* logs, long ml_bs, const float* gld,
                     const float* gz_in, float* gz, int N, long per, hipStream_t s) {
    if (N == 0 || per == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_prior_bwd, dim3(cdiv(per, 256), N), dim3(256), 0, s, z, mean, logs, ml_bs, gld, gz_in, gz, per);
    GH_LAUNCH_CHECK("k_prior_bwd");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Generic convolution gradients.
// ------------------------------------------------------------------------------------------------
// wT[i][o][t'] = w[o][i][KK-1-t']  (k x k, 'SAME', stride 1: the input gradient is conv(g_y, wT))
__global__ void __launch_bounds__(256) k_weight_flipT(const float* __restrict__ w, float* __restrict__ wT, int Cout,
                                                      int Cin, int KK) {
    const long total = (long)Cout * Cin * KK;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int t = (int)(e % KK);
    const long r = e / KK;
    const int o = (int)(r % Cout), i = (int)(r / Cout);
    wT[e] = w[((long)o * Cin + i) * KK + (KK - 1 - t)];
}

int launch_weight_flipT(const float* w, float* wT, int Cout, int Cin, int ksize, hipStream_t s) {
    const long total = (long)Cout * Cin * ksize * ksize;
    hipLaunchKernelGGL(k_weight_flipT, dim3(cdiv(total, 256)), dim3(256), 0, s, w, wT, Cout, Cin, ksize * ksize);
    GH_LAUNCH_CHECK("k_weight_flipT");
    return GLOWHIP_OK;
}

// dW[o][i][ky][kx] = sum_{n,y,x} gy[n,o,y,x] * x[n,i,y+ky-1,x+kx-1].  One workgroup per (o, i).
template <int KS>
__global__ void __launch_bounds__(256) k_wgrad_direct(const float* __restrict__ gy, const float* __restrict__ x,
                                                      long x_bs, float* __restrict__ dw, int N, int Cin, int H, int W,
                                                      int Cout) {
    __shared__ double red[4];
    const int i = blockIdx.x, o = blockIdx.y;
    const int HW = H * W;
    constexpr int KK = KS * KS;
    double acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = 0.0;
    const long count = (long)N * HW;
    for (long e = threadIdx.x; e < count; e += 256) {
        const long n = e / HW;
        const int p = (int)(e - n * HW);
        const int py = p / W, px = p - py * W;
        const float g = gy[(n * Cout + o) * HW + p];
        const float* xc = x + n * x_bs + (long)i * HW;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int yy = py + ky - KS / 2, xx = px + kx - KS / 2;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc[ky * KS + kx] += (double)(g * xc[yy * W + xx]);
            }
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        const double tot = block_sum<256>(acc[t], red);
        if (threadIdx.x == 0) dw[((long)o * Cin + i) * KK + t] = (float)tot;
    }
}

int launch_wgrad_direct(const float* gy, const float* x, long x_bs, float* dw, int N, int Cin, int H, int W, int Cout,
                        int ksize, hipStream_t s) {
    GH_REQUIRE(ksize == 1 || ksize == 3, "wgrad: kernel size %d unsupported", ksize);
    if (ksize == 3) hipLaunchKernelGGL(k_wgrad_direct<3>, dim3(Cin, Cout), dim3(256), 0, s, gy, x, x_bs, dw, N, Cin, H, W, Cout);
    else hipLaunchKernelGGL(k_wgrad_direct<1>, dim3(Cin, Cout), dim3(256), 0, s, gy, x, x_bs, dw, N, Cin, H, W, Cout);
    GH_LAUNCH_CHECK("k_wgrad_direct");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// fp64 accumulators -> fp32 gradient tensors.
//   out[i] = acc[i] * mul + add_const               (plain)
//   invconv: dW[o][i] = acc[o*C+i] + G*HW*Winv[i*C+o]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_grad_finalize(const double* __restrict__ acc, float* __restrict__ out, int n,
                                                       const double* __restrict__ gsum, double add_mul) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (float)(acc[i] + (gsum ? gsum[0] * add_mul : 0.0));
}

__global__ void __launch_bounds__(256) k_grad_finalize_w(const double* __restrict__ acc, float* __restrict__ out, int C,
                                                         const double* __restrict__ gsum, double hw,
                                                         const float* __restrict__ winv) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * C) return;
    const int o = e / C, i = e - o * C;
    out[e] = (float)(acc[e] + gsum[0] * hw * (double)winv[i * C + o]);
}

// gsum[0] = sum_n gld[n]  (one small workgroup, fixed order)
__global__ void __launch_bounds__(64) k_sum_gld(const float* __restrict__ gld, int N, double* __restrict__ gsum) {
    double a = 0.0;
    for (int i = threadIdx.x; i < N; i += 64) a += (double)gld[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) gsum[0] = a;
}

__global__ void __launch_bounds__(256) k_grad_finalize_batched(const GradJob* __restrict__ jobs,
                                                               const double* __restrict__ gsum) {
    const GradJob j = jobs[blockIdx.y];
    if (j.out == nullptr) return;
    const double g = gsum[0] * j.add_mul;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < j.n; e += gridDim.x * 256) {
        double v = j.acc[e];
        for (int k = 1; k < j.copies; ++k) v += j.acc[k * j.stride + e];
        if (j.winv) {
            const int o = e / j.C, i = e - o * j.C;
            v += g * (double)j.winv[i * j.C + o];
        } else {
            v += g;
        }
        j.out[e] = (float)v;
    }
}

int launch_grad_finalize_batched(const GradJob* jobs_dev, int n_jobs, const double* gsum, hipStream_t s) {
    if (n_jobs == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_grad_finalize_batched, dim3(8, n_jobs), dim3(256), 0, s, jobs_dev, gsum);
    GH_LAUNCH_CHECK("k_grad_finalize_batched");
    return GLOWHIP_OK;
}

int launch_grad_finalize(const double* acc, float* out, int n, const double* gsum, double add_mul, hipStream_t s) {
    if (n == 0 || out == nullptr) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_grad_finalize, dim3(cdiv(n, 256)), dim3(256), 0, s, acc, out, n, gsum, add_mul);
    GH_LAUNCH_CHECK("k_grad_finalize");
    return GLOWHIP_OK;
}

int launch_grad_finalize_w(const double* acc, float* out, int C, const double* gsum, double hw, const float* winv,
                           hipStream_t s) {
    if (out == nullptr) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_grad_finalize_w, dim3(cdiv(C * C, 256)), dim3(256), 0, s, acc, out, C, gsum, hw, winv);
    GH_LAUNCH_CHECK("k_grad_finalize_w");
    return GLOWHIP_OK;
}

int launch_sum_gld(const float* gld, int N, double* gsum, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_gld, dim3(1), dim3(64), 0, s, gld, N, gsum);
    GH_LAUNCH_CHECK("k_sum_gld");
    return GLOWHIP_OK;
}

// gld[n] = -nll_grad[n] / (ln2 * CHW)   (nll = -objective / (ln2*CHW), network/model.py:448-450)
__global__ void __launch_bounds__(256) k_gld_from_nll(const float* __restrict__ nll_grad, float* __restrict__ gld, int N,
                                                      double inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) gld[i] = (float)(-(double)nll_grad[i] * inv);
}

int launch_gld_from_nll(const float* nll_grad, float* gld, int N, double inv, hipStream_t s) {
    if (N == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_gld_from_nll, dim3(cdiv(N, 256)), dim3(256), 0, s, nll_grad, gld, N, inv);
    GH_LAUNCH_CHECK("k_gld_from_nll");
    return GLOWHIP_OK;
}

}  // namespace glowhip
