// backward.h -- launcher interface of the gradient kernels (backward.hip) used by plan_train.hip.
#pragma once
#include "common.h"

namespace glowhip {

struct CouplingBwdArgs {
    const float* hout;       // (N,Cout,HW) saved post-scale output of f.4
    const float* z2out; long z_bs;   // second half of the step output (z2')
    const float* g2; long g_bs;      // incoming gradient of z2'   (second half of g_out)
    float* gy2;                      // outgoing gradient of y2 (same strides as g2; may alias g2)
    float* gpre;             // (N,Cout,HW) gradient of (conv + bias) of f.4
    const float* e4;         // (Cout) exp(3 logs)
    const float* gld;        // (N) dL/dlogdet_n
    double* acc_b; double* acc_l;    // (Cout) f.4 bias / logs gradient accumulators
    int N, Ch, Cout, HW, affine;
    int acc_copies = 1; long acc_stride = 0;   // as in ChanMixBwdArgs: workgroup b adds into copy b % acc_copies (k_coupling_bwd4)
};
int launch_coupling_bwd(const CouplingBwdArgs& a, hipStream_t s);

struct SplitBwdArgs {
    const float* hout;       // (N,2Ch,HW) saved prior conv output
    const float* z2; long z_bs;      // the split-off half
    float* gz2; long g_bs;           // its gradient (written)
    float* gpre;             // (N,2Ch,HW)
    const float* e4; const float* gld;
    double* acc_b; double* acc_l;
    int N, Ch, HW;
};
int launch_split_bwd(const SplitBwdArgs& a, hipStream_t s);

int launch_act_bwd(float* g, const float* h, const float* e, int N, int Cm, int HW, double* acc_b, double* acc_l,
                   hipStream_t s);

struct ChanMixBwdArgs {
    const float* x; long x_bs;       // step input
    const float* gy; float* gx; long g_bs;   // gradient of y in, gradient of x out (may alias)
    const float* bias; const float* scale;   // actnorm bias, exp(3 logs)
    const float* matrix;             // W (C,C) or null
    const int32_t* gather_inv;       // inverse permutation table or null
    double* acc_w; double* acc_b; double* acc_l;
    int N, C, HW;
    // the accumulators exist in acc_copies copies acc_stride doubles apart (workgroup b adds into copy b % acc_copies; the
    // finalize job sums them): 1024 workgroups x (C^2 + 2C) fp64 atomics on ten cache lines ran at the rate of those lines'
    // L2 channels (35 us per launch, whatever the level)
    int acc_copies = 1; long acc_stride = 0;
    // add_part != null: the gradient of y's FIRST add_C channels is gy + add_scale * (the partial sums a backward k_cnet launch left
    // in add_part: MS row-split copies + halo rows, geometry add_*) -- gathered here instead of by a k_cbwd_finish launch
    const float* add_part = nullptr; float add_scale = 0.f;
    int add_C = 0, add_MS = 0, add_tiles = 0, add_R = 0, add_NI = 0, add_lpxt = 0, add_H = 0, add_W = 0;
    int w_lds = 0;     // (set by launch_chanmix_bwd: the matrix is staged in LDS)
};
struct WgradReduceJobs;
// reduce != null: the split-K reductions of the FlowStep's weight-gradient GEMMs run in the same launch (k_chanmix_bwd_reduce)
int launch_chanmix_bwd(const ChanMixBwdArgs& a, hipStream_t s, const WgradReduceJobs* reduce = nullptr);

int launch_prior_bwd(const float* z, const float* mean, const float* logs, long ml_bs, const float* gld,
                     const float* gz_in, float* gz, int N, long per, hipStream_t s);
int launch_weight_flipT(const float* w, float* wT, int Cout, int Cin, int ksize, hipStream_t s);
int launch_wgrad_direct(const float* gy, const float* x, long x_bs, float* dw, int N, int Cin, int H, int W, int Cout,
                        int ksize, hipStream_t s);
// ---- wgrad_mfma.hip
int launch_shift_expand(const float* src, long src_bs, float* out, int N, int C, int H, int W, int rows_pad, int sign,
                        hipStream_t s);
bool wgrad_mfma_supported(int HW, int Mpad, int Npad);
size_t wgrad_mfma_partial_floats(int Mpad, int Npad, int N, int HW);
// operand (0: A, 1: B) of launch_wgrad_mfma given as the (N, C, H, W) tensor whose shift-expanded rows (c * 9 + tap) it stands for
// (what launch_shift_expand would write): the GEMM's loader gathers them itself.  The operand pointer / batch stride are the tensor's.
struct WgradTaps { int operand, C, H, W, sign; };
struct WgradReduceJob;
int launch_wgrad_mfma(const float* A, long a_bs, const float* B, long b_bs, float* partial, float* dw, int N, int HW,
                      int Mpad, int Npad, int Mreal, int Nreal, int mode, hipStream_t s, float sh_scale = 0.f,   // sh_scale > 0: f16-pipe kernel, gradient operand A pre-scaled by it
                      double* rowsum = nullptr,    // (f16-pipe kernel) rowsum[m] += sum over all pixels of A's row m -- the bias gradient when A = g_u
                      const WgradTaps* taps = nullptr,
                      struct WgradReduceJob* defer = nullptr,    // non-null: skip the split-K reduction, describe it in *defer instead
                      int b_valid = 0,    // > 0 (f16-pipe kernel, plain B): only that many rows of B exist, the rest of the column tile is zero
                      int b_half = 0,     // 1 (f16-pipe kernel, plain B): B points at an fp16 tensor (the taped h1 / h2), b_bs in elements
                      int tiled = 0);     // bit 0 / bit 1 (f16-pipe kernel, plain operand): A / B is pixel-tile-major [pixel / 32][rows][pixel % 32]
                                          // over the batch's pixels (what the taping / backward k_cnet write, sh.h), rows = Mpad / Npad; its batch stride is unused
                                          // bit 2: A holds g * sh_scale * 2^11 (the backward k_cnet's g_u2 / g_u0); row sums and dW come out as for g
// f.4's + f.0's GEMMs of one FlowStep as one launch (wgrad_mfma.hip); GLOWHIP_EINVAL (nothing launched) for shapes it does not take
bool wgrad_pair_ok(int HW, int m4, int hid, int n0);
int launch_wgrad_pair(const float* gpre, long gpre_bs, const void* h2_half, float* partial4, float* dw4, int m4, int m4_real,
                      const float* gu0, const float* y1, long y1_bs, float* partial0, float* dw0, int n0, int n0_real,
                      int N, int HW, int hid, float sh_scale, double* rowsum0, const WgradTaps& t4, const WgradTaps& t0, int tiled4,
                      int tiled0, struct WgradReduceJob* rj4, struct WgradReduceJob* rj0, hipStream_t s);
// ... and f.2's (A = g_u2, B = h1 fp16, both pixel-tile-major, A pre-scaled) as the third
int launch_wgrad_trio(const float* gu2, const void* h1_half, float* partial2, float* dw2, double* rowsum2,
                      const float* gpre, long gpre_bs, const void* h2_half, float* partial4, float* dw4, int m4, int m4_real,
                      const float* gu0, const float* y1, long y1_bs, float* partial0, float* dw0, int n0, int n0_real,
                      int N, int HW, int hid, float sh_scale, double* rowsum0, const WgradTaps& t4, const WgradTaps& t0,
                      struct WgradReduceJob* rj2, struct WgradReduceJob* rj4, struct WgradReduceJob* rj0, hipStream_t s);
// fp16 pixel-tile-major [pixel / 32][R][pixel % 32] -> fp32 (N, R, HW): a taped hidden tensor for the per-layer backward kernels
int launch_half_to_float(const void* src_half, float* dst, int N, int R, int HW, hipStream_t s);
struct WgradReduceJob { const float* partial; float* dw; int splits, Mpad, Npad, Mreal, Nreal, mode; };
struct WgradReduceJobs { WgradReduceJob job[3]; int n; };
int launch_wgrad_reduce_batched(const WgradReduceJobs& j, hipStream_t s);

// one launch for all reduction-type parameter gradients of a backward sweep
struct GradJob {
    const double* acc; float* out; int n;
    double add_mul;          // out[i] = acc[i] + gsum * add_mul            (winv == null)
    const float* winv; int C; // out[o*C+i] = acc[o*C+i] + gsum * add_mul * winv[i*C+o]   (invconv weight)
    int copies = 1; long stride = 0;   // acc[i] = sum over copies of acc[copy * stride + i]
};
int launch_grad_finalize_batched(const GradJob* jobs_dev, int n_jobs, const double* gsum, hipStream_t s);
// ActNorm log-scale gradient of a convolution layer y = (conv(x, W) + b) * exp(3 logs) from its weight and bias gradients:
//   d logs[r] = 3 sum_p g_y y = 3 sum_p g_u (u + b) = 3 (<W[r], dW[r]> + b[r] db[r]),   g_u = g_y exp(3 logs), u = conv(x, W)
// (sum_p g_u[r][p] u[r][p] = sum_k W[r][k] sum_p g_u[r][p] x_k[p] = <W[r], dW[r]>): no pass over the activations at all.
struct LogsJob { const float* w; const float* dw; const float* b; const double* db; float* out; int rows, K; };
int launch_logs_from_dw_batched(const LogsJob* jobs_dev, int n_jobs, hipStream_t s);
int launch_grad_finalize(const double* acc, float* out, int n, const double* gsum, double add_mul, hipStream_t s);
int launch_grad_finalize_w(const double* acc, float* out, int C, const double* gsum, double hw, const float* winv,
                           hipStream_t s);
int launch_sum_gld(const float* gld, int N, double* gsum, hipStream_t s);
int launch_gld_from_nll(const float* nll_grad, float* gld, int N, double inv, hipStream_t s);

}  // namespace glowhip
