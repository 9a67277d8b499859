// pointwise.hip -- HBM-bound kernels of the flow path: squeeze, ActNorm (+ data-dependent init),
// invertible 1x1 conv / channel permutation (per-pixel channel mixing from LDS-resident pixels),
// coupling / split tails of the generic path, Gaussian log-density reductions, log-det finalisation.
//
// Every kernel reads NCHW with the pixel index on the lane axis, so a wave64 touches 256 contiguous
// bytes per channel plane (coalesced), and reduces per-sample sums with a fixed shuffle tree plus one
// fixed-point atomic per workgroup (order-independent => bitwise reproducible).
#include "kernels.h"

namespace glowhip {

// ------------------------------------------------------------------------------------------------
// Squeeze2d / unsqueeze (network/module.py:551-592), optional dequantisation-noise add on the input
// (network/model.py:421).  One thread per OUTPUT element (gather form) => coalesced stores.
// ------------------------------------------------------------------------------------------------
// Dequantisation noise drawn in the kernel (network/model.py:421: x + U(0, 1/2^n_bits)): counter-based Philox4x32-10, key =
// the caller's seed, counter = (element index / 4, call number); element i takes word i % 4.  No noise tensor in HBM, no
// separate RNG launch; the same stream is available as a tensor through glowhip_dequant_noise (parity tests).
__device__ __forceinline__ float dequant_noise(unsigned long long seed, unsigned long long call, unsigned long long i, float scale) {
    unsigned int c0 = (unsigned int)(i >> 2), c1 = (unsigned int)(i >> 34), c2 = (unsigned int)call, c3 = (unsigned int)(call >> 32);
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned int)p1; c3 = (unsigned int)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned int w = (i & 3) == 0 ? c0 : ((i & 3) == 1 ? c1 : ((i & 3) == 2 ? c2 : c3));
    return (float)(w >> 8) * (1.0f / 16777216.0f) * scale;          // 24 uniform bits in [0, 1), exact in fp32
}

__global__ void __launch_bounds__(256) k_dequant_noise(float* __restrict__ out, long n, unsigned long long seed,
                                                       unsigned long long call, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = dequant_noise(seed, call, (unsigned long long)i, scale);
}

int launch_dequant_noise(float* out, long n, unsigned long long seed, unsigned long long call, float scale, hipStream_t s) {
    if (n == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_dequant_noise, dim3(cdiv(n, 256)), dim3(256), 0, s, out, n, seed, call, scale);
    GH_LAUNCH_CHECK("k_dequant_noise");
    return GLOWHIP_OK;
}

__global__ void __launch_bounds__(256) k_squeeze(const float* __restrict__ x, const float* __restrict__ noise,
                                                 float* __restrict__ y, long total, int C, int H, int W, int f,
                                                 int reverse, RngSpec rng) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    long src;
    if (!reverse) {
        const int Ho = H / f, Wo = W / f, Co = C * f * f;
        int w = (int)(idx % Wo);
        long t = idx / Wo;
        int h = (int)(t % Ho);
        t /= Ho;
        int co = (int)(t % Co);
        long n = t / Co;
        int c = co / (f * f), r = co % (f * f), i = r / f, j = r % f;
        src = ((n * C + c) * H + (h * f + i)) * (long)W + (w * f + j);
    } else {
        const int Ho = H * f, Wo = W * f, Co = C / (f * f);
        int ww = (int)(idx % Wo);
        long t = idx / Wo;
        int hh = (int)(t % Ho);
        t /= Ho;
        int c = (int)(t % Co);
        long n = t / Co;
        int cs = c * f * f + (hh % f) * f + (ww % f);
        src = ((n * C + cs) * H + hh / f) * (long)W + ww / f;
    }
    float v = x[src];
    if (noise) v += noise[src];
    else if (rng.on) v += dequant_noise(rng.seed, rng.call, (unsigned long long)src, rng.scale);
    y[idx] = v;
}

// The same gather from 8-bit pixels: y = u8 / divisor (+ noise) -- `ToTensor` (dataset/celeba.py:74-86, train.py:40-45),
// the dequantisation noise (network/model.py:421) and the first squeeze in one pass over a quarter of the bytes.
__global__ void __launch_bounds__(256) k_squeeze_u8(const uint8_t* __restrict__ x, const float* __restrict__ noise,
                                                    float* __restrict__ y, long total, int C, int H, int W, int f,
                                                    float divisor, RngSpec rng) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int Ho = H / f, Wo = W / f, Co = C * f * f;
    int w = (int)(idx % Wo);
    long t = idx / Wo;
    int h = (int)(t % Ho);
    t /= Ho;
    int co = (int)(t % Co);
    long n = t / Co;
    int c = co / (f * f), r = co % (f * f), i = r / f, j = r % f;
    const long src = ((n * C + c) * H + (h * f + i)) * (long)W + (w * f + j);
    float v = (float)x[src] / divisor;
    if (noise) v += noise[src];
    else if (rng.on) v += dequant_noise(rng.seed, rng.call, (unsigned long long)src, rng.scale);
    y[idx] = v;
}

// Factor-2 squeeze / unsqueeze as a register transpose (the shapes every flow plan uses; W % 4 == 0): a thread owns FOUR consecutive
// pixels of an input row pair -- two 16-byte loads of fully coalesced rows -- and emits the four output planes' two-pixel runs as
// 8-byte stores (a wave writes 512 contiguous bytes per plane).  One Philox call serves the four pixels of a row segment (the
// generic kernel evaluates all ten rounds per element and keeps one word of four).  32-bit index arithmetic, no LDS.  Same values,
// bit for bit, as k_squeeze / k_squeeze_u8 (tests/test_gpu_parity.py, test_gpu_infer.py).
__device__ __forceinline__ void dequant_noise4(unsigned long long seed, unsigned long long call, unsigned long long i4, float scale, float (&out)[4]) {
    unsigned int c0 = (unsigned int)(i4 >> 2), c1 = (unsigned int)(i4 >> 34), c2 = (unsigned int)call, c3 = (unsigned int)(call >> 32);
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned int)p1; c3 = (unsigned int)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned int w[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = (float)(w[q] >> 8) * (1.0f / 16777216.0f) * scale;
}

// SRC: 0 = float input, 1 = 8-bit input (forward only).  grid.x covers (plane, row pair, group of 4 input pixels) of ONE image.
template <int SRC>
__global__ void __launch_bounds__(256) k_squeeze2_fwd(const void* __restrict__ xin, const float* __restrict__ noise, float* __restrict__ y,
                                                      int C, int H, int W, float divisor, RngSpec rng) {
    const int W4 = W >> 2, Ho = H >> 1, Wo = W >> 1;
    const int per_img = C * Ho * W4;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= per_img) return;
    const int n = blockIdx.y;
    const int k = t % W4, r = t / W4, h = r % Ho, c = r / Ho;
    const long img_in = (long)n * C * H * W;
    float v[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long src = img_in + ((long)c * H + 2 * h + i) * W + 4 * k;
        if (SRC == 1) {
            const uchar4 u = *reinterpret_cast<const uchar4*>((const uint8_t*)xin + src);
            v[i][0] = (float)u.x / divisor; v[i][1] = (float)u.y / divisor; v[i][2] = (float)u.z / divisor; v[i][3] = (float)u.w / divisor;
        } else {
            const float4 f = *reinterpret_cast<const float4*>((const float*)xin + src);
            v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
        }
        if (noise) {
            const float4 f = *reinterpret_cast<const float4*>(noise + src);
            v[i][0] += f.x; v[i][1] += f.y; v[i][2] += f.z; v[i][3] += f.w;
        } else if (rng.on) {
            float z[4];
            dequant_noise4(rng.seed, rng.call, (unsigned long long)src, rng.scale, z);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[i][q] += z[q];
        }
    }
    // out[n][4c + 2i + j][h][2k + m] = in[n][c][2h + i][4k + 2m + j]
    float* o = y + ((long)n * 4 * C + 4 * c) * Ho * Wo + (long)h * Wo + 2 * k;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<float2*>(o + (long)(2 * i + j) * Ho * Wo) = make_float2(v[i][j], v[i][2 + j]);
}

// unsqueeze: x (N, C, H, W) -> y (N, C/4, 2H, 2W); a thread reads two pixels of each of the four planes, writes two 16-byte row segments
__global__ void __launch_bounds__(256) k_squeeze2_rev(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W) {
    const int W2 = W >> 1, Co = C >> 2;
    const int per_img = Co * H * W2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= per_img) return;
    const int n = blockIdx.y;
    const int k = t % W2, r = t / W2, h = r % H, c = r / H;
    const float* in = x + ((long)n * C + 4 * c) * H * W + (long)h * W + 2 * k;
    float2 p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = *reinterpret_cast<const float2*>(in + (long)q * H * W);
    float* o = y + (((long)n * Co + c) * 2 * H + 2 * h) * (2 * W) + 4 * k;
    *reinterpret_cast<float4*>(o) = make_float4(p[0].x, p[1].x, p[0].y, p[1].y);
    *reinterpret_cast<float4*>(o + 2 * W) = make_float4(p[2].x, p[3].x, p[2].y, p[3].y);
}

int launch_squeeze_u8(const uint8_t* x, const float* noise, float* y, int N, int C, int H, int W, int f, float divisor,
                      hipStream_t s, const RngSpec* rng) {
    GH_REQUIRE(f >= 1 && H % f == 0 && W % f == 0, "squeeze2d(u8): H,W must be divisible by the factor");
    long total = (long)N * C * H * W;
    if (total == 0) return GLOWHIP_OK;
    if (f == 2 && W % 4 == 0 && N <= 65535 && (long)C * H * W < (1l << 31) && ((size_t)x & 3) == 0 && (!noise || ((size_t)noise & 15) == 0)) {
        hipLaunchKernelGGL(k_squeeze2_fwd<1>, dim3(cdiv((long)C * (H / 2) * (W / 4), 256), N), dim3(256), 0, s, (const void*)x, noise, y, C, H, W,
                           divisor, rng ? *rng : RngSpec{});
        GH_LAUNCH_CHECK("k_squeeze2_fwd(u8)");
        return GLOWHIP_OK;
    }
    hipLaunchKernelGGL(k_squeeze_u8, dim3(cdiv(total, 256)), dim3(256), 0, s, x, noise, y, total, C, H, W, f, divisor, rng ? *rng : RngSpec{});
    GH_LAUNCH_CHECK("k_squeeze_u8");
    return GLOWHIP_OK;
}

int launch_squeeze(const float* x, const float* noise, float* y, int N, int C, int H, int W, int f, int reverse,
                   hipStream_t s, const RngSpec* rng) {
    GH_REQUIRE(f >= 1, "squeeze2d: factor must be >= 1");
    if (!reverse) GH_REQUIRE(H % f == 0 && W % f == 0, "squeeze2d: H,W must be divisible by factor");
    else GH_REQUIRE(C >= f * f && C % (f * f) == 0, "unsqueeze2d: C must be a multiple of factor^2");
    long total = (long)N * C * H * W;
    if (total == 0) return GLOWHIP_OK;
    const bool small = N <= 65535 && (long)C * H * W < (1l << 31);
    if (f == 2 && !reverse && W % 4 == 0 && small && ((size_t)x & 15) == 0 && ((size_t)y & 7) == 0 && (!noise || ((size_t)noise & 15) == 0)) {
        hipLaunchKernelGGL(k_squeeze2_fwd<0>, dim3(cdiv((long)C * (H / 2) * (W / 4), 256), N), dim3(256), 0, s, (const void*)x, noise, y, C, H, W,
                           1.0f, rng ? *rng : RngSpec{});
        GH_LAUNCH_CHECK("k_squeeze2_fwd");
        return GLOWHIP_OK;
    }
    if (f == 2 && reverse && W % 2 == 0 && small && !noise && !(rng && rng->on) && ((size_t)x & 7) == 0 && ((size_t)y & 15) == 0) {
        hipLaunchKernelGGL(k_squeeze2_rev, dim3(cdiv((long)(C / 4) * H * (W / 2), 256), N), dim3(256), 0, s, x, y, C, H, W);
        GH_LAUNCH_CHECK("k_squeeze2_rev");
        return GLOWHIP_OK;
    }
    hipLaunchKernelGGL(k_squeeze, dim3(cdiv(total, 256)), dim3(256), 0, s, x, noise, y, total, C, H, W, f, reverse, rng ? *rng : RngSpec{});
    GH_LAUNCH_CHECK("k_squeeze");
    return GLOWHIP_OK;
}

__global__ void __launch_bounds__(256) k_copy_strided(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                      long ybs, long per_sample) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_sample) return;
    long n = blockIdx.y;
    y[n * ybs + i] = x[n * xbs + i];
}

int launch_copy_strided(const float* x, long xbs, float* y, long ybs, int N, long per_sample, hipStream_t s) {
    if (N == 0 || per_sample == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_copy_strided, dim3(cdiv(per_sample, 256), N), dim3(256), 0, s, x, xbs, y, ybs, per_sample);
    GH_LAUNCH_CHECK("k_copy_strided");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// ActNorm data-dependent init (network/module.py:86-120): one workgroup per channel, two passes
// (mean, then mean of the centred square), fp64 accumulation, fixed reduction tree.
// ------------------------------------------------------------------------------------------------
// `pooled` (batch_variance=True, :109-110): logs[c] receives the channel's centred second moment instead; k_actnorm_pool_logs
// turns the C moments into the one pooled value.
__global__ void __launch_bounds__(256) k_actnorm_init(const float* __restrict__ x, long xbs, int N, int HW,
                                                      float scale, float* __restrict__ bias,
                                                      float* __restrict__ logs, int pooled) {
    __shared__ double red[4];
    __shared__ float s_bias;
    const int c = blockIdx.x;
    const long count = (long)N * HW;
    const float* xc = x + (long)c * HW;
    double acc = 0.0;
    for (long i = threadIdx.x; i < count; i += 256) {
        long n = i / HW;
        int p = (int)(i - n * HW);
        acc += (double)xc[n * xbs + p];
    }
    double tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) s_bias = (float)(-(tot / (double)count));
    __syncthreads();
    const float b = s_bias;
    acc = 0.0;
    for (long i = threadIdx.x; i < count; i += 256) {
        long n = i / HW;
        int p = (int)(i - n * HW);
        float d = xc[n * xbs + p] + b;
        acc += (double)(d * d);
    }
    tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) {
        float var = (float)(tot / (double)count);
        bias[c] = b;
        logs[c] = pooled ? var : logf(scale / (sqrtf(var) + 1e-6f)) / LOGSCALE;
    }
}

// batch_variance=True: the reference takes mean((x + bias)^2) over ALL dimensions (`ops.reduce_mean(x ** 2, keepdim=True)`,
// network/module.py:109-110 -- every channel holds N*HW elements, so that is the mean of the per-channel moments) and copies the one
// resulting log-scale into every channel.  One workgroup: fp64 sum of the C moments in a fixed tree, then the broadcast.
__global__ void __launch_bounds__(256) k_actnorm_pool_logs(float* __restrict__ logs, int C, float scale) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) acc += (double)logs[c];
    __shared__ float s_logs;
    const double tot = block_sum<256>(acc, red);      // (valid on thread 0; ends with a barrier: every moment is read before the first is overwritten)
    if (threadIdx.x == 0) {
        const float var = (float)(tot / (double)C);
        s_logs = logf(scale / (sqrtf(var) + 1e-6f)) / LOGSCALE;
    }
    __syncthreads();
    const float v = s_logs;
    for (int c = threadIdx.x; c < C; c += 256) logs[c] = v;
}

int launch_actnorm_init(const float* x, long xbs, int N, int C, int HW, float scale, float* bias, float* logs,
                        hipStream_t s, int batch_variance) {
    GH_REQUIRE(N > 0 && C > 0 && HW > 0, "actnorm_init: empty input");
    hipLaunchKernelGGL(k_actnorm_init, dim3(C), dim3(256), 0, s, x, xbs, N, HW, scale, bias, logs, batch_variance);
    GH_LAUNCH_CHECK("k_actnorm_init");
    if (batch_variance) {
        hipLaunchKernelGGL(k_actnorm_pool_logs, dim3(1), dim3(256), 0, s, logs, C, scale);
        GH_LAUNCH_CHECK("k_actnorm_pool_logs");
    }
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Channel mixer: ActNorm (+bias, *scale) fused with the per-pixel C x C mat-vec of the invertible
// 1x1 convolution (network/module.py:359-363) or the Permutation2d gather (:392-397).
// HBM traffic = read C + write C floats per pixel, the algorithmic minimum.
// ------------------------------------------------------------------------------------------------
// 64 pixels x 4 channel groups per workgroup: wave g stages channels g, g+4, ... of the 64 pixels into LDS
// (column [c][px]: lane = bank, conflict-free) and then produces output channels [g*C/4, (g+1)*C/4) -- the
// matrix rows a wave needs are wave-uniform (scalar cache), OB outputs share each LDS read.  Splitting the
// outputs over 4 waves keeps the grid >= 4x larger than one-thread-per-pixel, which is what the deep levels
// (C=48 on 4096 pixels) need to occupy the chip.
// PX pixels per workgroup, 256 / PX output groups per pixel: 64 x 4 for the large levels (the output group is the wave id, so
// the matrix row is a scalar operand), 16 x 16 for the deep ones (4096 pixels x 48 channels would otherwise be 64
// workgroups of 576 serial FMAs per thread: 16 us for 9 MFLOP).
template <int PX>
__global__ void __launch_bounds__(256) k_chanmix(ChanMixArgs a, int stage_matrix) {
    constexpr int OG = 256 / PX;
    extern __shared__ __attribute__((aligned(16))) float v[];  // [C][PX]
    const int px = threadIdx.x & (PX - 1);
    const int og = PX == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x / PX);
    const long gp = (long)blockIdx.x * PX + px;
    const long total = (long)a.N * a.HW;
    const bool valid = gp < total;
    const long n = valid ? gp / a.HW : 0;
    const int p = valid ? (int)(gp - n * a.HW) : 0;
    const int C = a.C;
    const float* pa = a.in_a + n * a.in_a_bs + p;
    const float* pb = a.in_b + n * a.in_b_bs + p;
    float* po = a.out + n * a.out_bs + p;
    const bool an = a.bias != nullptr;
    if (a.sq_src) {       // squeezed view of the un-squeezed input (+ dequantisation noise, 8-bit scaling): the squeeze pass is gone
        const int Wo = a.sq_W >> 1, h = p / Wo, w = p - h * Wo, C0 = C >> 2;
        const long img = (long)n * C0 * (4l * a.HW);
        for (int c = og; c < C; c += OG) {
            float xv = 0.f;
            if (valid) {
                const long src = img + ((long)(c >> 2) * (2 * (a.HW / Wo)) + 2 * h + ((c >> 1) & 1)) * a.sq_W + 2 * w + (c & 1);
                xv = a.sq_u8 ? (float)reinterpret_cast<const uint8_t*>(a.sq_src)[src] / a.sq_div : reinterpret_cast<const float*>(a.sq_src)[src];
                if (a.sq_noise) xv += a.sq_noise[src];
                else if (a.sq_rng.on) xv += dequant_noise(a.sq_rng.seed, a.sq_rng.call, (unsigned long long)src, a.sq_rng.scale);
            }
            if (!a.reverse && an) xv = (xv + a.bias[c]) * a.scale[c];
            v[c * PX + px] = xv;
        }
    } else
    // four channels per trip, every load of a trip issued (from clamped, always valid addresses) before the first value is used
    for (int c0 = og; c0 < C; c0 += 4 * OG) {
        float xv[4], bb[4], sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min(c0 + u * OG, C - 1);
            xv[u] = (c < a.Ca) ? pa[(long)c * a.HW] : pb[(long)(c - a.Ca) * a.HW];
            bb[u] = (!a.reverse && an) ? a.bias[c] : 0.f;
            sc[u] = (!a.reverse && an) ? a.scale[c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * OG;
            if (c >= C) continue;
            float x = valid ? xv[u] : 0.f;
            if (!a.reverse && an) x = (x + bb[u]) * sc[u];
            v[c * PX + px] = x;
        }
    }
    // the matrix goes through LDS as well (one coalesced pass): read from global inside the FMA loop, its C*C dependent
    // loads were the whole run time of this kernel at the deep levels.  Eight requests in flight per trip.
    float* mlds = v + C * PX;
    if (a.matrix && stage_matrix)
        for (int e0 = threadIdx.x; e0 < C * C; e0 += 256 * 8) {
            float mv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mv[k] = a.matrix[min(e0 + 256 * k, C * C - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (e0 + 256 * k < C * C) mlds[e0 + 256 * k] = mv[k];
        }
    __syncthreads();
    const int per = (C + OG - 1) / OG;
    const int o_begin = og * per, o_end = min(C, o_begin + per);
    constexpr int OB = 4;
    for (int o = o_begin; o < o_end; o += OB) {
        float r[OB];
        if (a.matrix) {
#pragma unroll
            for (int j = 0; j < OB; ++j) r[j] = 0.f;
            const float* m = stage_matrix ? mlds + o * C : a.matrix + (long)o * C;
            if (o + OB <= o_end) {
                for (int i = 0; i < C; ++i) {
                    const float vi = v[i * PX + px];
#pragma unroll
                    for (int j = 0; j < OB; ++j) r[j] = fmaf(m[j * C + i], vi, r[j]);
                }
            } else {
                for (int i = 0; i < C; ++i) {
                    const float vi = v[i * PX + px];
#pragma unroll
                    for (int j = 0; j < OB; ++j)
                        if (o + j < o_end) r[j] = fmaf(m[j * C + i], vi, r[j]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < OB; ++j) {
                const int oc = o + j;
                r[j] = oc < o_end ? v[(a.gather ? a.gather[oc] : oc) * PX + px] : 0.f;
            }
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < OB; ++j) {
                const int oc = o + j;
                if (oc < o_end) {
                    float out = r[j];
                    if (a.reverse && an) out = out * a.scale[oc] - a.bias[oc];
                    po[(long)oc * a.HW] = out;
                }
            }
        }
    }
}

// Wide levels (C > 110: the C x C matrix no longer fits beside the pixel tile; 8x8 / 4x4 pixels per image, so a launch has
// few pixels): workgroup = 16 pixels x 32 OUTPUT channels, grid = (pixel tiles, C / 32).  The 16 x C input tile and the 32
// matrix rows are staged in LDS; same operation order per output as k_chanmix (r = fma(m[o][i], v[i], r), i ascending).
__global__ void __launch_bounds__(256) k_chanmix_wide(ChanMixArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // v [C][16], then m [32][C]
    const int C = a.C, px = threadIdx.x & 15, og = threadIdx.x >> 4;      // og: 0..15, two outputs each
    float* v = wsm;
    float* m = wsm + C * 16;
    const long gp = (long)blockIdx.x * 16 + px;
    const long total = (long)a.N * a.HW;
    const bool valid = gp < total;
    const long n = valid ? gp / a.HW : 0;
    const int p = valid ? (int)(gp - n * a.HW) : 0;
    const float* pa = a.in_a + n * a.in_a_bs + p;
    const float* pb = a.in_b + n * a.in_b_bs + p;
    const bool an = a.bias != nullptr;
    // eight channels per round, every load of a round issued (from clamped, always valid addresses) before the first value is used:
    // as a plain loop the 24 rounds of C = 384 were 24 dependent trips to memory (31 us per launch for 75 MFLOP)
    for (int c0 = og; c0 < C; c0 += 16 * 8) {
        float xv[8], bb[8], sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = min(c0 + 16 * u, C - 1);
            xv[u] = (c < a.Ca) ? pa[(long)c * a.HW] : pb[(long)(c - a.Ca) * a.HW];
            bb[u] = (!a.reverse && an) ? a.bias[c] : 0.f;
            sc[u] = (!a.reverse && an) ? a.scale[c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + 16 * u;
            if (c >= C) continue;
            float x = valid ? xv[u] : 0.f;
            if (!a.reverse && an) x = (x + bb[u]) * sc[u];
            v[c * 16 + px] = x;
        }
    }
    float* po = a.out + n * a.out_bs + p;
    // gridDim.y == 1 (output aliases an input: in place): this workgroup walks all output slices itself, its pixels already in LDS
    const int nsl = gridDim.y == 1 ? (C + 31) / 32 : 1;
    for (int sl = 0; sl < nsl; ++sl) {
        const int o0 = (gridDim.y == 1 ? sl : (int)blockIdx.y) * 32;
        __syncthreads();
        if (a.matrix) {      // 32 consecutive rows of the matrix = one contiguous run of 32 * C floats (clamped at the matrix' end)
            const float* src = a.matrix + (long)o0 * C;
            const int have = min(32, C - o0) * C;
            for (int e0 = threadIdx.x; e0 < 32 * C; e0 += 256 * 8) {
                float mv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) mv[u] = src[min(e0 + 256 * u, have - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + 256 * u;
                    if (e < 32 * C) m[e] = e < have ? mv[u] : 0.f;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ol = og * 2 + j, o = o0 + ol;
            if (o >= C) continue;
            float r;
            if (a.matrix) {
                r = 0.f;
                const float* mr = m + ol * C;
                int i = 0;
                if ((C & 3) == 0)                 // (16-byte aligned matrix rows)
                for (; i + 4 <= C; i += 4) {      // (four LDS reads of each operand in flight; same operation order)
                    const float4 m4 = *reinterpret_cast<const float4*>(mr + i);
                    const float v0 = v[i * 16 + px], v1 = v[(i + 1) * 16 + px], v2 = v[(i + 2) * 16 + px], v3 = v[(i + 3) * 16 + px];
                    r = fmaf(m4.x, v0, r); r = fmaf(m4.y, v1, r); r = fmaf(m4.z, v2, r); r = fmaf(m4.w, v3, r);
                }
                for (; i < C; ++i) r = fmaf(mr[i], v[i * 16 + px], r);
            } else {
                r = v[(a.gather ? a.gather[o] : o) * 16 + px];
            }
            if (a.reverse && an) r = r * a.scale[o] - a.bias[o];
            if (valid) po[(long)o * a.HW] = r;
        }
    }
}

bool chanmix_squeeze_foldable(int C) { return C % 4 == 0 && (size_t)C * C * sizeof(float) <= 48 * 1024; }

int launch_chanmix(const ChanMixArgs& a, hipStream_t s) {
    GH_REQUIRE(a.C > 0 && a.C <= 512, "channel mixer: C=%d unsupported (1..512)", a.C);
    const long total = (long)a.N * a.HW;
    if (total == 0) return GLOWHIP_OK;
    GH_REQUIRE(!a.sq_src || (chanmix_squeeze_foldable(a.C) && !a.reverse), "channel mixer: the squeezed-view input needs the LDS-matrix kernel, forward");
    if ((size_t)a.C * a.C * sizeof(float) > 48 * 1024) {
        // output slices over gridDim.y only when the output does not alias an input (other workgroups read the same pixels)
        const float* o_lo = a.out; const float* o_hi = a.out + (long)a.N * a.out_bs;
        const bool alias = (a.in_a >= o_lo && a.in_a < o_hi) || (a.in_b >= o_lo && a.in_b < o_hi);
        const size_t lds = ((size_t)a.C * 16 + (size_t)32 * a.C) * sizeof(float);
        if (lds > 32 * 1024) (void)hipFuncSetAttribute((const void*)k_chanmix_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_chanmix_wide, dim3(cdiv(total, 16), alias ? 1 : cdiv(a.C, 32)), dim3(256), lds, s, a);
        GH_LAUNCH_CHECK("k_chanmix_wide");
        return GLOWHIP_OK;
    }
    const int stage = (size_t)a.C * a.C * sizeof(float) <= 48 * 1024;   // C <= 110: the matrix fits next to the pixel tile
    const size_t mbytes = stage ? (size_t)a.C * a.C * sizeof(float) : 0;
    if (total < 16384 && a.C >= 16 && stage) {   // deep levels: 16 pixels x 16 output groups per workgroup
        hipLaunchKernelGGL(k_chanmix<16>, dim3(cdiv(total, 16)), dim3(256), (size_t)a.C * 16 * sizeof(float) + mbytes, s, a, stage);
        GH_LAUNCH_CHECK("k_chanmix");
        return GLOWHIP_OK;
    }
    const size_t lds = (size_t)a.C * 64 * sizeof(float) + mbytes;
    if (lds > 32 * 1024)
        (void)hipFuncSetAttribute((const void*)k_chanmix<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_chanmix<64>, dim3(cdiv(total, 64)), dim3(256), lds, s, a, stage);
    GH_LAUNCH_CHECK("k_chanmix");
    return GLOWHIP_OK;
}

// logdet_out[n] = (in ? in[n] : 0) + sign * mul * sum_k term[k]   (data-independent layer terms)
__global__ void __launch_bounds__(64) k_add_const_logdet(const float* __restrict__ in, float* __restrict__ out, int N,
                                                         const float* __restrict__ term, float mul, int count,
                                                         float sign) {
    double acc = 0.0;
    for (int k = threadIdx.x; k < count; k += 64) acc += (double)term[k];
    acc = wave_sum(acc);
    acc = __shfl(acc, 0, 64);
    const float d = (float)(acc * (double)mul) * sign;
    for (int n = threadIdx.x; n < N; n += 64) out[n] = (in ? in[n] : 0.f) + d;
}

int launch_add_const_logdet(const float* in, float* out, int N, const float* term, float mul, int count, float sign,
                            hipStream_t s) {
    if (N == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_add_const_logdet, dim3(1), dim3(64), 0, s, in, out, N, term, mul, count, sign);
    GH_LAUNCH_CHECK("k_add_const_logdet");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Coupling tail of the generic path (network/model.py:105-113 forward, :131-139 reverse).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_coupling_tail(CouplingTailArgs a) {
    __shared__ double red[4];
    const long n = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)a.Ch * a.HW;
    double ld = 0.0;
    if (e < per) {
        const int c = (int)(e / a.HW);
        const int p = (int)(e - (long)c * a.HW);
        const float z2 = a.z2_in[n * a.z2_in_bs + e];
        float out;
        if (a.affine) {
            const float* hn = a.h + n * 2 * per;
            const float shift = hn[(long)(2 * c) * a.HW + p];
            const float sc = sigmoidf_(hn[(long)(2 * c + 1) * a.HW + p] + 2.0f);
            out = a.reverse ? (z2 / sc - shift) : ((z2 + shift) * sc);
            ld = (double)logf(sc);
        } else {
            const float hv = a.h[n * per + e];
            out = a.reverse ? (z2 - hv) : (z2 + hv);
        }
        a.z2_out[n * a.z2_out_bs + e] = out;
    }
    if (a.affine && a.acc) {
        double tot = block_sum<256>(ld, red);
        if (threadIdx.x == 0) fix_atomic_add(a.acc, n, a.N, a.reverse ? -tot : tot);
    }
}

int launch_coupling_tail(const CouplingTailArgs& a, hipStream_t s) {
    const long per = (long)a.Ch * a.HW;
    if (a.N == 0 || per == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_coupling_tail, dim3(cdiv(per, 256), a.N), dim3(256), 0, s, a);
    GH_LAUNCH_CHECK("k_coupling_tail");
    return GLOWHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Split2d tail (network/module.py:526-536) + GaussianDiag (:437-483).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gauss_logp1(float mean, float logs, float x) {
    const float d = x - mean;
    return -0.5f * (LOG_2PI_F + 2.0f * logs + (d * d) / expf(2.0f * logs));
}

__global__ void __launch_bounds__(256) k_split_tail(SplitTailArgs a) {
    __shared__ double red[4];
    const long n = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)a.Ch * a.HW;
    double lp = 0.0;
    if (e < per) {
        const int c = (int)(e / a.HW);
        const int p = (int)(e - (long)c * a.HW);
        const float* hn = a.h + n * 2 * per;
        const float mean = hn[(long)(2 * c) * a.HW + p];
        const float logs = hn[(long)(2 * c + 1) * a.HW + p];
        if (!a.reverse) {
            lp = (double)gauss_logp1(mean, logs, a.z2[n * a.z2_bs + e]);
        } else {
            a.z2_out[n * a.z2_out_bs + e] = mean + expf(logs) * a.eps[n * per + e];
        }
    }
    if (!a.reverse && a.acc) {
        double tot = block_sum<256>(lp, red);
        if (threadIdx.x == 0) fix_atomic_add(a.acc, n, a.N, tot);
    }
}

int launch_split_tail(const SplitTailArgs& a, hipStream_t s) {
    const long per = (long)a.Ch * a.HW;
    if (a.N == 0 || per == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_split_tail, dim3(cdiv(per, 256), a.N), dim3(256), 0, s, a);
    GH_LAUNCH_CHECK("k_split_tail");
    return GLOWHIP_OK;
}

__global__ void __launch_bounds__(256) k_gaussian_logp(const float* __restrict__ x, long xbs,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ logs, long mlbs, long per,
                                                       unsigned long long* __restrict__ acc) {
    __shared__ double red[4];
    const long n = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    double lp = 0.0;
    if (e < per) {
        const float m = mean ? mean[n * mlbs + e] : 0.f;
        const float l = logs ? logs[n * mlbs + e] : 0.f;
        lp = (double)gauss_logp1(m, l, x[n * xbs + e]);
    }
    double tot = block_sum<256>(lp, red);
    if (threadIdx.x == 0) fix_atomic_add(acc, n, gridDim.y, tot);
}

int launch_gaussian_logp(const float* x, long xbs, const float* mean, const float* logs, long mlbs, int N, int C,
                         int HW, unsigned long long* acc, hipStream_t s) {
    const long per = (long)C * HW;
    if (N == 0 || per == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_gaussian_logp, dim3(cdiv(per, 256), N), dim3(256), 0, s, x, xbs, mean, logs, mlbs, per, acc);
    GH_LAUNCH_CHECK("k_gaussian_logp");
    return GLOWHIP_OK;
}

__global__ void __launch_bounds__(256) k_zero_acc(unsigned long long* acc, int words, unsigned* cnt, int cnt_words) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < words) acc[i] = 0ull;      // accumulators, non-finite flags (and the extra accumulator rows)
    if (i < cnt_words) cnt[i] = 0u;    // arrival counters of the fused finishing (they return to zero by themselves; a call that was cut short must not poison the next)
}

int launch_zero_acc(unsigned long long* acc, int N, hipStream_t s, int extra_rows, unsigned* cnt, size_t cnt_words) {
    if (N == 0) return GLOWHIP_OK;
    const int words = (2 + extra_rows) * N;
    if (!cnt) cnt_words = 0;
    hipLaunchKernelGGL(k_zero_acc, dim3(cdiv(std::max<long>(words, (long)cnt_words), 256)), dim3(256), 0, s, acc, words, cnt, (int)cnt_words);
    GH_LAUNCH_CHECK("k_zero_acc");
    return GLOWHIP_OK;
}

__global__ void __launch_bounds__(256) k_finalize(const float* __restrict__ in,
                                                  const unsigned long long* __restrict__ acc,
                                                  const double* __restrict__ konst, double sign, double offset,
                                                  double scale, float* __restrict__ out,
                                                  float* __restrict__ out_unscaled, int N, int extra_rows) {
    int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    unsigned long long a = acc ? acc[n] : 0ull;
    for (int k = 0; k < extra_rows; ++k) a += acc[(long)(2 + k) * N + n];      // (two's-complement sums: any grouping gives the same bits)
    double v = (in ? (double)in[n] : 0.0) + offset + (konst ? sign * konst[0] : 0.0) + (acc ? fix_to_double(a) : 0.0);
    if (acc && acc[N + n] != 0ull) {      // partial sums of this sample that were NaN / +inf / -inf (common.h)
        const unsigned long long fl = acc[N + n];
        v = ((fl & 1ull) || (fl & 6ull) == 6ull) ? __builtin_nan("") : ((fl & 2ull) ? __builtin_inf() : -__builtin_inf());
    }
    if (out_unscaled) out_unscaled[n] = (float)v;
    if (out) out[n] = (float)(v * scale);
}

int launch_finalize(const float* in, const unsigned long long* acc, const double* konst, double sign, double offset,
                    double scale, float* out, float* out_unscaled, int N, hipStream_t s, int extra_rows) {
    if (N == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_finalize, dim3(cdiv(N, 256)), dim3(256), 0, s, in, acc, konst, sign, offset, scale, out,
                       out_unscaled, N, acc ? extra_rows : 0);
    GH_LAUNCH_CHECK("k_finalize");
    return GLOWHIP_OK;
}

// Conv2d's ActNorm + ReLU as a separate in-place pass (network/module.py:258-259 + :310): only the data-dependent init pass
// needs the un-fused form (the statistics are taken between the convolution and its ActNorm)
__global__ void __launch_bounds__(256) k_bias_scale_relu(float* __restrict__ h, long total4, int HW4, int C, const float* __restrict__ bias,
                                                         const float* __restrict__ scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)((i / HW4) % C);
    const float b = bias[c], sc = scale[c];
    float4 v = reinterpret_cast<float4*>(h)[i];
    v.x = relu_((v.x + b) * sc); v.y = relu_((v.y + b) * sc); v.z = relu_((v.z + b) * sc); v.w = relu_((v.w + b) * sc);
    reinterpret_cast<float4*>(h)[i] = v;
}
__global__ void __launch_bounds__(256) k_bias_scale_relu1(float* __restrict__ h, long total, int HW, int C, const float* __restrict__ bias,
                                                          const float* __restrict__ scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / HW) % C);
    h[i] = relu_((h[i] + bias[c]) * scale[c]);
}

int launch_bias_scale_relu(float* h, int N, int C, int HW, const float* bias, const float* scale, hipStream_t s) {
    const long total = (long)N * C * HW;
    if (total == 0) return GLOWHIP_OK;
    if (HW % 4 == 0) {
        hipLaunchKernelGGL(k_bias_scale_relu, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, h, total / 4, HW / 4, C, bias, scale);
    } else {
        hipLaunchKernelGGL(k_bias_scale_relu1, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h, total, HW, C, bias, scale);
    }
    GH_LAUNCH_CHECK("k_bias_scale_relu");
    return GLOWHIP_OK;
}

// Range / non-finite status of a finished call (glowhip_plan_status): one workgroup per sample
__global__ void __launch_bounds__(256) k_status(const unsigned long long* __restrict__ acc, int N, const float* __restrict__ result,
                                                long elems, int32_t* __restrict__ status) {
    __shared__ int any;
    const int n = blockIdx.x;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    int bad = 0;
    if (result) {
        const float* r = result + (long)n * elems;
        for (long e = threadIdx.x; e < elems; e += 256) bad |= !isfinite(r[e]);
    }
    if (bad) any = 1;      // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0) status[n] = (int32_t)(acc[N + n] & 7ull) | (any ? 8 : 0);
}

int launch_status(const unsigned long long* acc, int N, const float* result, long elems, int32_t* status, hipStream_t s) {
    if (N == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_status, dim3(N), dim3(256), 0, s, acc, N, result, elems, status);
    GH_LAUNCH_CHECK("k_status");
    return GLOWHIP_OK;
}

}  // namespace glowhip
