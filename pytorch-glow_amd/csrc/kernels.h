// kernels.h -- internal launcher interface between the plan executor / C-ABI and the kernel files.
#pragma once
#include "common.h"

namespace glowhip {

// ---------------------------------------------------------------- pointwise.hip
// in-kernel dequantisation noise (pointwise.hip): U(0, scale) from Philox4x32-10(seed; element index, call number)
struct RngSpec { int on; unsigned long long seed, call; float scale; };
int launch_squeeze(const float* x, const float* noise, float* y, int N, int C, int H, int W, int f, int reverse,
                   hipStream_t s, const RngSpec* rng = nullptr);
int launch_squeeze_u8(const uint8_t* x, const float* noise, float* y, int N, int C, int H, int W, int f, float divisor,
                      hipStream_t s, const RngSpec* rng = nullptr);
int launch_dequant_noise(float* out, long n, unsigned long long seed, unsigned long long call, float scale, hipStream_t s);
int launch_copy_strided(const float* x, long xbs, float* y, long ybs, int N, long per_sample, hipStream_t s);
int launch_actnorm_init(const float* x, long xbs, int N, int C, int HW, float scale, float* bias, float* logs,
                        hipStream_t s, int batch_variance = 0);

// Channel mixer: the ActNorm + (Invertible1x1Conv | Permutation2d) pair of a FlowStep as ONE pass.
struct ChanMixArgs {
    const float* in_a; long in_a_bs;  // channels [0, Ca)
    const float* in_b; long in_b_bs;  // channels [Ca, C)  (element (n,c,p) at in_b[n*bs + (c-Ca)*HW + p])
    int Ca;
    float* out; long out_bs;
    const float* bias;     // (C) or null: no ActNorm
    const float* scale;    // (C) exp(+3 logs) for forward, exp(-3 logs) for reverse
    const float* matrix;   // (C,C) row-major matrix to apply, or null
    const int32_t* gather; // (C) channel gather table, or null
    int reverse;           // 0: actnorm then mix; 1: mix then inverse actnorm
    int N, C, HW;
    // Optional (forward, k_chanmix only): the input is the SQUEEZED view of an un-squeezed tensor that is never materialised --
    // channel c of pixel (h, w) = sq_src[n][c / 4][2 h + (c / 2) % 2][2 w + c % 2] (Squeeze2d, network/module.py:573-592), with the
    // dequantisation noise of network/model.py:421 (tensor or in-kernel Philox draw) and the 8-bit scaling of the data loader
    // (dataset/celeba.py:74-86) applied on the way in.  sq_src: float (sq_u8 = 0) or uint8 (sq_u8 = 1) of shape (N, C/4, 2 Ho, sq_W).
    const void* sq_src; int sq_u8; float sq_div; const float* sq_noise; RngSpec sq_rng; int sq_W;
};
int launch_chanmix(const ChanMixArgs& a, hipStream_t s);
bool chanmix_squeeze_foldable(int C);      // the squeezed-view input (ChanMixArgs::sq_src) is served for this channel count

int launch_add_const_logdet(const float* in, float* out, int N, const float* term_a, float mul_a, int count_a,
                            float sign, hipStream_t s);

// tails of the generic path
struct CouplingTailArgs {
    const float* h;        // (N, Cout, HW) contiguous: coupling-net output
    const float* z2_in; long z2_in_bs;
    float* z2_out; long z2_out_bs;
    int N, Ch, HW;         // Ch = channels of z2
    int affine, reverse;
    unsigned long long* acc;  // (N) fixed-point logdet accumulators (affine only)
};
int launch_coupling_tail(const CouplingTailArgs& a, hipStream_t s);

struct SplitTailArgs {
    const float* h;        // (N, 2*Ch, HW): prior conv output, mean = even, logs = odd channels
    const float* z2; long z2_bs;   // forward: z2 to score
    const float* eps;      // reverse: (N,Ch,HW) injected draw
    float* z2_out; long z2_out_bs; // reverse: sampled z2
    int N, Ch, HW, reverse;
    unsigned long long* acc;
};
int launch_split_tail(const SplitTailArgs& a, hipStream_t s);

int launch_gaussian_logp(const float* x, long xbs, const float* mean, const float* logs, long mlbs, int N, int C,
                         int HW, unsigned long long* acc, hipStream_t s);
int launch_zero_acc(unsigned long long* acc, int N, hipStream_t s, int extra_rows = 0, unsigned* cnt = nullptr, size_t cnt_words = 0);   // extra_rows: ACC_EXTRA for a plan's workspace (common.h)
// out[n] = scale * ((in ? in[n] : 0) + offset + sign*(konst ? *konst : 0) + fix(acc[n]))
int launch_finalize(const float* in, const unsigned long long* acc, const double* konst, double sign, double offset,
                    double scale, float* out, float* out_unscaled, int N, hipStream_t s, int extra_rows = 0);

// in place: h[n][c][p] = relu((h + bias[c]) * scale[c])  (Conv2d's ActNorm + ReLU after the data-dependent init set them)
int launch_bias_scale_relu(float* h, int N, int C, int HW, const float* bias, const float* scale, hipStream_t s);
// status[n] = sticky non-finite flags of sample n (acc[N + n], common.h) | 8 if any of result[n*elems .. +elems) is not finite
int launch_status(const unsigned long long* acc, int N, const float* result, long elems, int32_t* status, hipStream_t s);

// ---------------------------------------------------------------- lu.hip
size_t invconv_scratch_bytes(int C);
int launch_invconv_prepare(const float* w, int C, float* winv, float* logabsdet, void* scratch, hipStream_t s);

// batched parameter preparation (glowhip_plan_pack): device-resident job tables, offsets into `packed`
struct StepPrepJob {
    const float* w;         // invconv weight (C,C) or null (permutation step)
    const float* an_logs;   // actnorm.logs (C)
    int C, HW;
    size_t winv_off, logabsdet_off, konst_off, scratch_off;
};
int launch_step_prepare_batched(const StepPrepJob* jobs_dev, int n, int max_lds_c, void* packed, hipStream_t s, int want_inverse = 1,
                                int max_c = 0, int all_small = 0);   // want_inverse = 0: log|det W| only (W^-1 is left stale)
bool step_prepare_small_takes(const StepPrepJob& j);   // all_small = every job of the table passes this (lu.hip k_step_prepare_small)

struct ScaleJob { const float* logs; size_t scale_off, inv_off; int n; int has_inv; };
enum { REPACK_WIDE = 0, REPACK_TAIL = 1, REPACK_FIRST = 2,                 // exact-fp32 MFMA kernels' images
       REPACK_SH2_GEMM = 3, REPACK_SH2_FIRST = 4, REPACK_SH2_TAIL = 5 };   // true-scale split-half images with per-row scales (sh.h SH2)
struct RepackJob {
    const float* w; size_t out_off; int kind;
    int Cin, Cout, K, Kpad;      // wide: K = Cin*k*k
    int paired, MT; long total;  // tail
    const float* fold_bias; const float* fold_logs;  // first: ActNorm folded into the image (NULL: plain weights)
    int use;                     // who reads this image: bit 0 = inference kernels (encode/decode/glow_forward), bit 1 = training
    int transposed;              // source is the FORWARD weight (Cin,Cout,3,3) of which this is the input-gradient conv:
                                 // element (o, ci, tap) = w[ci][o][8 - tap]
    size_t w_off;                // w == nullptr: the source is packed + w_off (a transposed copy made by launch_flipT_batched, or W^-1)
    int after_lu;                // the source is W^-1 (packed + w_off): built after the LU factorisations, on their stream
    int kperm;                   // SH2_GEMM / SH2_TAIL: k-permuted image (sh.h sh2_kperm: the coupling-network kernels of cnet_sh.hip /
                                 // cnet1w_sh.hip, whose B operand of f.2 / f.4 is the previous layer's accumulator block)
};
// dst[i][o][ks-1-tap] = src[o][i][tap] (ks = 9: 3x3 weights, 1: a plain transpose): the weight of the input-gradient convolution,
// in the reference layout, for the SH2 image kernels of the backward k_cnet launch (plan_train.hip)
struct FlipJob { const float* src; size_t dst_off; int O, I, ks; };
int launch_flipT_batched(const FlipJob* jobs_dev, int n_jobs, const int* max_tiles3 /* per member of a (f.4, f.2, f.0) triple */, void* packed, hipStream_t s);
// rj_dev: the repack jobs SORTED by kind group (legacy kinds | SH2_GEMM | SH2_FIRST | SH2_TAIL), n_kind[4] their counts;
// tail_blocks: workgroups per SH2_TAIL job (8 output channels each)
// s_legacy: the stream of the legacy-kind image kernel (the same as s, or a side stream forked from it)
// first_blocks: workgroups per SH2_FIRST job (32 rows each for the jobs with >= 64 input channels; 2 x 256 rows otherwise)
int launch_pack_batched(const ScaleJob* sj_dev, int n_scale, const RepackJob* rj_dev, const int* n_kind, int tail_blocks, void* packed,
                        hipStream_t s, hipStream_t s_legacy, int first_blocks = 2);
// n SH2_GEMM jobs on their own (the images of W^-1, after the LU factorisations)
int launch_repack_sh2_gemm(const RepackJob* rj_dev, int n, void* packed, hipStream_t s);

// ---------------------------------------------------------------- conv_direct.hip
struct ConvArgs {
    const float* x; long x_bs;
    const float* w;          // (Cout,Cin,k,k) reference layout
    const float* bias;       // (Cout) or null, added before post_bias
    const float* post_bias;  // (Cout) or null
    const float* post_logs;  // (Cout) or null -> * exp(3*logs)
    const float* post_scale; // (Cout) or null -> * scale (precomputed exp(3*logs)); wins over post_logs
    int relu;
    float* y;                // (N,Cout,H,W) contiguous
    int N, Cin, H, W, Cout, ksize;
};
int launch_conv_direct(const ConvArgs& a, hipStream_t s);

}  // namespace glowhip
