// api.hip -- extern "C" entry points of the single-layer primitives + error plumbing.
#include <stdarg.h>
#include <stdio.h>

#include "kernels.h"
#include "conv_mfma.h"

namespace glowhip {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace glowhip

using namespace glowhip;

extern "C" {

int glowhip_version(void) { return GLOWHIP_VERSION; }
void glowhip_debug_force_tail_tile(int pixels) { conv_mfma_tail_force_tile(pixels); }
const char* glowhip_last_error(void) { return g_err; }

int glowhip_squeeze2d(const float* x, float* y, int N, int C, int H, int W, int factor, int reverse,
                      glowhip_stream_t stream) {
    GH_REQUIRE(x && y, "squeeze2d: null tensor");
    GH_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0, "squeeze2d: bad shape");
    return launch_squeeze(x, nullptr, y, N, C, H, W, factor, reverse, (hipStream_t)stream);
}

int glowhip_actnorm_init(const float* x, long batch_stride, int N, int C, int HW, float scale, float* bias, float* logs,
                         glowhip_stream_t stream) {
    GH_REQUIRE(x && bias && logs, "actnorm_init: null tensor");
    return launch_actnorm_init(x, batch_stride, N, C, HW, scale, bias, logs, (hipStream_t)stream);
}

int glowhip_actnorm_init_batch_variance(const float* x, long batch_stride, int N, int C, int HW, float scale, float* bias,
                                        float* logs, glowhip_stream_t stream) {
    GH_REQUIRE(x && bias && logs, "actnorm_init: null tensor");
    return launch_actnorm_init(x, batch_stride, N, C, HW, scale, bias, logs, (hipStream_t)stream, 1);
}

// ActNorm.forward as a stand-alone elementwise pass (network/module.py:122-149)
__global__ void __launch_bounds__(256) k_actnorm_elementwise(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ logs, long total, int C, int HW,
                                                             int reverse) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / HW) % C);
    const float l3 = logs[c] * LOGSCALE;
    y[i] = reverse ? (x[i] * expf(-l3) - bias[c]) : ((x[i] + bias[c]) * expf(l3));
}

int glowhip_actnorm(const float* x, float* y, const float* bias, const float* logs, int N, int C, int HW, int reverse,
                    const float* logdet_in, float* logdet_out, glowhip_stream_t stream) {
    GH_REQUIRE(x && y && bias && logs, "actnorm: null tensor");
    GH_REQUIRE(N >= 0 && C > 0 && C <= 4096 && HW > 0, "actnorm: bad shape N=%d C=%d HW=%d", N, C, HW);
    hipStream_t s = (hipStream_t)stream;
    if (logdet_out)
        GH_TRY(launch_add_const_logdet(logdet_in, logdet_out, N, logs, LOGSCALE * (float)HW, C, reverse ? -1.f : 1.f, s));
    if (N == 0) return GLOWHIP_OK;
    long total = (long)N * C * HW;
    hipLaunchKernelGGL(k_actnorm_elementwise, dim3(cdiv(total, 256)), dim3(256), 0, s, x, y, bias, logs, total, C, HW,
                       reverse);
    GH_LAUNCH_CHECK("k_actnorm_elementwise");
    return GLOWHIP_OK;
}

size_t glowhip_invconv_scratch_bytes(int C) { return C > 0 ? invconv_scratch_bytes(C) : 0; }

int glowhip_invconv_prepare(const float* w, int C, float* winv, float* logabsdet, void* scratch,
                            glowhip_stream_t stream) {
    GH_REQUIRE(w, "invconv_prepare: null weight");
    return launch_invconv_prepare(w, C, winv, logabsdet, scratch, (hipStream_t)stream);
}

int glowhip_invconv(const float* x, float* y, const float* m, const float* logabsdet, int N, int C, int HW, int reverse,
                    const float* logdet_in, float* logdet_out, glowhip_stream_t stream) {
    GH_REQUIRE(x && y && m, "invconv: null tensor");
    GH_REQUIRE(x != y, "invconv: in-place call not supported");
    hipStream_t s = (hipStream_t)stream;
    if (logdet_out) {
        GH_REQUIRE(logabsdet, "invconv: logdet requested without logabsdet");
        GH_TRY(launch_add_const_logdet(logdet_in, logdet_out, N, logabsdet, (float)HW, 1, reverse ? -1.f : 1.f, s));
    }
    ChanMixArgs a{};
    const long chw = (long)C * HW;
    a.in_a = x; a.in_a_bs = chw; a.in_b = x; a.in_b_bs = chw; a.Ca = C;
    a.out = y; a.out_bs = chw; a.matrix = m; a.reverse = 0; a.N = N; a.C = C; a.HW = HW;
    return launch_chanmix(a, s);
}

int glowhip_permute_channels(const float* x, float* y, const int32_t* idx, int N, int C, int HW,
                             glowhip_stream_t stream) {
    GH_REQUIRE(x && y && idx, "permute_channels: null tensor");
    GH_REQUIRE(x != y, "permute_channels: in-place call not supported");
    ChanMixArgs a{};
    const long chw = (long)C * HW;
    a.in_a = x; a.in_a_bs = chw; a.in_b = x; a.in_b_bs = chw; a.Ca = C;
    a.out = y; a.out_bs = chw; a.gather = idx; a.reverse = 0; a.N = N; a.C = C; a.HW = HW;
    return launch_chanmix(a, (hipStream_t)stream);
}

int glowhip_conv2d(const float* x, long x_batch_stride, const float* w, const float* bias, float* y, int N, int Cin,
                   int H, int W, int Cout, int ksize, const float* post_bias, const float* post_logs, int relu,
                   glowhip_stream_t stream) {
    GH_REQUIRE(x && w && y, "conv2d: null tensor");
    ConvArgs a{x, x_batch_stride, w, bias, post_bias, post_logs, nullptr, relu, y, N, Cin, H, W, Cout, ksize};
    return launch_conv_direct(a, (hipStream_t)stream);
}

int glowhip_gaussian_logp(const float* x, long x_stride, const float* mean, const float* logs, long ml_stride, int N,
                          int C, int HW, const float* in, float* out, void* scratch16N, glowhip_stream_t stream) {
    GH_REQUIRE(x && out && scratch16N, "gaussian_logp: null argument");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* acc = (unsigned long long*)scratch16N;
    GH_TRY(launch_zero_acc(acc, N, s));
    GH_TRY(launch_gaussian_logp(x, x_stride, mean, logs, ml_stride, N, C, HW, acc, s));
    return launch_finalize(in, acc, nullptr, 1.0, 0.0, 1.0, out, nullptr, N, s);
}

}  // extern "C"
