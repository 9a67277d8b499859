// conv_direct.hip -- shape-generic direct convolution (any Cin/Cout, k in {1,3}, 'SAME', stride 1):
// the fallback for shapes the MFMA kernels do not cover (odd channel counts, tiny images) and the
// independent cross-check of the MFMA kernels in the tests.  One thread per output pixel, OB output
// channels per thread; the weights of an output-channel block are wave-uniform (scalar cache), the
// input plane is read with the pixel index on the lane axis (coalesced), 9 taps reuse each load OB x.
// Epilogue (network/module.py:258-259 / :296): (+bias) (+post_bias) (*exp(3 logs)) (relu).
#include "kernels.h"

namespace glowhip {

template <int KS, int OB>
__global__ void __launch_bounds__(256) k_conv_direct(ConvArgs a) {
    const int HW = a.H * a.W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int ob = blockIdx.y * OB;
    const long n = blockIdx.z;
    if (p >= HW) return;
    const int py = p / a.W, px = p - py * a.W;
    float acc[OB];
#pragma unroll
    for (int j = 0; j < OB; ++j) acc[j] = 0.f;
    const float* xn = a.x + n * a.x_bs;
    constexpr int KK = KS * KS;
    const long wstride = (long)a.Cin * KK;
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* xc = xn + (long)ci * HW;
        float xv[KK];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int yy = py + ky - KS / 2, xx = px + kx - KS / 2;
                const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                xv[ky * KS + kx] = ok ? xc[yy * a.W + xx] : 0.f;
            }
        const float* wc = a.w + (long)ob * wstride + (long)ci * KK;
#pragma unroll
        for (int j = 0; j < OB; ++j) {
            if (ob + j < a.Cout) {
#pragma unroll
                for (int t = 0; t < KK; ++t) acc[j] = fmaf(wc[j * wstride + t], xv[t], acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < OB; ++j) {
        const int o = ob + j;
        if (o < a.Cout) {
            float v = acc[j];
            if (a.bias) v += a.bias[o];
            if (a.post_bias) v += a.post_bias[o];
            if (a.post_scale) v *= a.post_scale[o];
            else if (a.post_logs) v *= expf(a.post_logs[o] * LOGSCALE);
            if (a.relu) v = relu_(v);
            a.y[(n * a.Cout + o) * HW + p] = v;
        }
    }
}

int launch_conv_direct(const ConvArgs& a, hipStream_t s) {
    GH_REQUIRE(a.ksize == 1 || a.ksize == 3, "conv2d: kernel size %d unsupported (1 or 3)", a.ksize);
    GH_REQUIRE(a.Cin > 0 && a.Cout > 0 && a.H > 0 && a.W > 0, "conv2d: empty shape");
    if (a.N == 0) return GLOWHIP_OK;
    constexpr int OB = 8;
    dim3 grid(cdiv((long)a.H * a.W, 256), cdiv(a.Cout, OB), a.N);
    if (a.ksize == 3) hipLaunchKernelGGL((k_conv_direct<3, OB>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_conv_direct<1, OB>), grid, dim3(256), 0, s, a);
    GH_LAUNCH_CHECK("k_conv_direct");
    return GLOWHIP_OK;
}

}  // namespace glowhip
