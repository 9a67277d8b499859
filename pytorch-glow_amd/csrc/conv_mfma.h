// conv_mfma.h -- MFMA (fp32-input, exact fp32) convolution kernels of the coupling network.
#pragma once
#include "common.h"

namespace glowhip {

// "Wide" convolutions: many output channels (hidden), ActNorm + ReLU epilogue (f.0 3x3 and f.2 1x1 of
// network/module.py:300-319).  Implicit GEMM  Y[o][pixel] = sum_k Wt[k][o] * im2col(X)[k][pixel].
bool conv_mfma_wide_supported(int Cin, int H, int W, int Cout, int ksize);
size_t conv_mfma_wide_packed_bytes(int Cin, int Cout, int ksize);
int conv_mfma_wide_pack(const float* w, int Cin, int Cout, int ksize, float* wt, hipStream_t s);
// post_bias / post_scale may be NULL (0 / 1) and relu = 0 for a plain GEMM (input-gradient use) -- 1x1 LDS-DMA path only
// splitk_scratch (optional, splitk_floats floats, must not overlap x or y): lets a launch with few output tiles and a long
// reduction split the reduction over workgroups (partial sums there, summed in a fixed order by a second small kernel)
int launch_conv_mfma_wide(const float* x, long x_bs, const float* wt, const float* post_bias, const float* post_scale,
                          float* y, int N, int Cin, int H, int W, int Cout, int ksize, hipStream_t s, int relu = 1,
                          float* splitk_scratch = nullptr, size_t splitk_floats = 0);

void conv_mfma_wide_disable_glds(int off);  // testing hook: 1 = use the register-staged k_conv_wide for 1x1 too

// f.0 with a stationary LDS pixel window (conv_mfma_first.hip): Cin a multiple of 6 (C/2 of every Glow level).
bool conv_mfma_first_supported(int Cin, int H, int W, int Cout);
size_t conv_mfma_first_packed_bytes(int Cin, int Cout);
// wf: packed image (weights * exp(3 logs), then bias * exp(3 logs)) produced by the REPACK_FIRST job
int launch_conv_mfma_first(const float* x, long x_bs, const float* wf, const float* bias_scaled, float* y, int N,
                           int Cin, int H, int W, int Cout, hipStream_t s, int relu = 1);

// "Tail" convolution: 3x3, few output channels (f.4 / Split2d prior), with the coupling / prior
// arithmetic and the per-sample log-det reduction fused into the epilogue.
enum TailMode {
    TAIL_PLAIN = 0,      // y = (conv + bias) * scale                      -> z2_out (N,Cout,HW)
    TAIL_AFFINE_FWD,     // z2 = (z2 + shift) * sigmoid(s + 2), acc += sum log sigmoid
    TAIL_AFFINE_REV,     // z2 = z2 / sigmoid(s + 2) - shift,  acc -= sum log sigmoid
    TAIL_ADD_FWD,        // z2 = z2 + h
    TAIL_ADD_REV,        // z2 = z2 - h
    TAIL_SPLIT_FWD,      // acc += logp(z2 | mean, logs)
    TAIL_SPLIT_REV,      // z2_out = mean + exp(logs) * eps   (eps passed as z2_in)
};
struct TailConvArgs {
    const float* x; long x_bs;   // (N,Cin,H,W)
    const float* wp;             // packed weights (conv_mfma_tail_pack)
    const float* bias;           // (Cout)
    const float* scale;          // (Cout) exp(3 logs)
    int N, Cin, H, W, Cout;
    int mode;
    const float* z2_in; long z2_in_bs;
    float* z2_out; long z2_out_bs;
    unsigned long long* acc;
    const float* zeros;          // >= 16 B of zeros in global memory (16-byte aligned), or null: no LDS-DMA kernels
    float* hout;                 // optional (training tape): (N,Cout,HW) <- (conv + bias) * scale
};
constexpr int TAIL_CK = 32;  // input channels per LDS chunk of the tail kernel
// number of 16-row MFMA tiles on the out-channel axis
__host__ __device__ inline int tail_mt(int Cout, int paired) {
    const int rows = paired ? 4 * ((Cout / 2 + 1) / 2) : Cout;
    return (rows + 15) / 16;
}
// out-row m -> original output channel (or -1).  paired: a lane's 4 consecutive rows are
// {even_c0, even_c1, odd_c0, odd_c1} of two coupling channels c0 = 2g, c1 = 2g+1.
__host__ __device__ inline int tail_row_channel(int m, int Cout, int paired) {
    if (!paired) return m < Cout ? m : -1;
    const int g = m >> 2, r = m & 3;
    const int c = 2 * g + (r & 1);
    return c < Cout / 2 ? 2 * c + (r >> 1) : -1;
}
inline int tail_chunks(int Cin) { return (Cin + TAIL_CK - 1) / TAIL_CK; }
inline int wide_kpad(int Cin, int ksize) { return (Cin * ksize * ksize + 31) / 32 * 32; }
bool conv_mfma_tail_supported(int Cin, int H, int W, int Cout);
size_t conv_mfma_tail_packed_bytes(int Cin, int Cout);
// paired=1: output channels come in (even, odd) = (shift|mean, scale|logs) pairs (affine coupling, Split2d)
int conv_mfma_tail_pack(const float* w, int Cin, int Cout, int paired, float* wp, hipStream_t s);
int launch_conv_mfma_tail(const TailConvArgs& a, hipStream_t s);
void conv_mfma_tail_force_tile(int tp);
void plan_disable_sh(int off);   // testing hook (plan.hip)
void plan_pack_one_stream(int on);   // testing hook (plan.hip): glowhip_plan_pack without its side-stream fork
void plan_train_disable_sh(int off);   // testing hook (plan_train.hip)
void plan_train_disable_cnet(int off); // testing hook (plan_train.hip): the training forward without the taping k_cnet
void wgrad_force_narrow(int on);          // testing hook (wgrad_mfma.hip): f.2's weight-gradient GEMM on 128-column tiles everywhere (A/B)
void plan_train_disable_cnet_bwd(int off); // testing hook (plan_train.hip): the input-gradient chain without the backward k_cnet
int launch_tail_dma_narrow(const TailConvArgs& a, int paired, hipStream_t s, int TP, int Y);  // W in {8,16}
int launch_tail_dma_wide(const TailConvArgs& a, int paired, hipStream_t s, int TP, int Y);    // W in {32,64,128}  // testing hook: 0 = automatic, else 16/32/64/128 pixels per block

}  // namespace glowhip
