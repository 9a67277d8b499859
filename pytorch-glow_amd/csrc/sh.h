// sh.h -- "split-half" (SH) operands: fp32 values carried as TWO fp16 numbers so the coupling network's contractions run
// on the f16 matrix pipe (v_mfma_f32_32x32x16_f16, 16x the rate of the fp32-input MFMA) at fp32 accuracy.
//
//   hi = fp16(v)                       |v - hi|            <= 2^-11 |v|   (round to nearest, 11-bit significand: half an ulp)
//   lo = fp16((v - hi) * 2^11)         |v - hi - lo/2^11|  <= 2^-22 |v|   worst case (lo keeps 11 bits of a residual that is itself
//                                                                          <= 2^-11 |v|); fp32's own half ulp is 2^-24 |v|.  The residual
//                                                                          is usually well below its bound: the measured distance to fp64
//                                                                          below equals the fp32 reference's
//
// (v - hi is exact in fp32; the 2^11 pre-scale keeps lo a NORMAL fp16 number.)  A product is evaluated as
//   a*b ~= a.hi*b.hi + (a.hi*b.lo + a.lo*b.hi) / 2^11            (the dropped lo*lo term is <= 2^-24 |a b|)
// with every fp16 x fp16 product exact in the fp32 accumulator (11+11 bits), two accumulators (main, cross) and
// fp32 accumulation as in the fp32 MFMA.  (This two-accumulator form serves the weight-gradient GEMMs of the training step, whose
// gradient operand needs its 30 binades; the forward / inverse / input-gradient kernels -- cnet_sh.hip, dnet_sh.hip -- use the
// single-accumulator SH2 form at the end of this file.)  Three f16 MFMAs per k-step instead of one fp32 MFMA of 1/16 the rate.
// Range: |v| < 65504 (fp16 max; larger values become inf and surface as a non-finite nll, they are never clipped
// silently); below 6.1e-5 the representation is absolute, 2.9e-11.  tests/diag_split_precision.py: on the celeba64
// model the deviation from an fp64 evaluation is 5.8e-6 (z), the fp32 reference's own is 5.4e-6.
//
// Weights use the form half [plane][K/8][M][8] (8 consecutive k of a row = one 16-byte group = a lane's MFMA A fragment).
#pragma once
#include "common.h"

namespace glowhip {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr float SH_LO_SCALE = 2048.0f;
constexpr float SH_LO_INV = 1.0f / 2048.0f;

__device__ __forceinline__ void sh_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * SH_LO_SCALE);
}

// =====================================================================================================================
// SH2: split-half operands at their TRUE scale -> ONE accumulator per output (cnet_sh.hip)
// =====================================================================================================================
// Every operand is first multiplied by an exact power of two that places it high in the fp16 range, then split
//   hi = fp16(V),  lo = fp16(V - hi)            (V = v * 2^e; no 2^11 pre-scale of lo)
// and a product is accumulated in a SINGLE fp32 accumulator:  acc += a.hi*b.hi + a.hi*b.lo + a.lo*b.hi  (three MFMAs, each
// fp16 x fp16 product exact).  Half the accumulator registers of the two-accumulator form above, which is what lets a
// workgroup own a 128-pixel x 512-row output tile and halves the weight bytes fetched per MFMA.
//   * gfx950's v_mfma_f32_32x32x16_f16 keeps fp16 SUBNORMAL inputs (scripts/ubench/mfma_denorm.hip, measured on MI355X), so a
//     small lo loses nothing but what fp16's fixed 2^-24 spacing cannot hold: |V - hi - lo| <= max(2^-22 |V|, 2^-25).
//   * weights: per output row o the exponent e[o] puts max_k |w'[o][k]| into [2^12, 2^13): the absolute floor 2^-25 is
//     2^-37 of the row's largest weight.  The row factor 2^-e[o] is undone in the epilogue (exact).
//   * activations (z1 window, h1, h2): fixed factor SH2_ACT_SCALE = 2^4: range |v| < 4094 (larger -> inf -> non-finite nll,
//     picked up by the sticky flag / exact-fp32 fall-back), floor 2^-29 = 1.9e-9 below |v| = 2^-7.
constexpr float SH2_ACT_SCALE = 16.0f;
constexpr float SH2_ACT_INV = 1.0f / 16.0f;
__device__ __forceinline__ void sh2_split(float V, _Float16& hi, _Float16& lo) {
    hi = (_Float16)V;
    lo = (_Float16)(V - (float)hi);
}
// Four values at once with the packed conversion (v_cvt_pk_f16_f32: two values per instruction, round to nearest even like the
// scalar conversion -- same bits as four sh2_split calls)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// MIX: the residual V - hi as ONE v_fma_mix_f32 per value (hi read as the f16 half it is; exact like conversion + subtraction):
// 29.3 -> 23.8 cycles per value-wave in scripts/ubench/epi_split.hip.  As inline asm it pins registers: the level-1 instance of
// k_cnet (at the 256-register limit) lost its allocation to it (h2 hand-over 5 k -> 32 k cycles), so only instances with registers
// to spare take it.
template <bool MIX = false>
__device__ __forceinline__ void sh2_split4(const f32x4_t& V, h4& hi, h4& lo) {
#pragma unroll
    for (int t = 0; t < 4; t += 2) {
        const f32x2_t vv = {V[t], V[t + 1]};
        const h2 x = __builtin_convertvector(vv, h2);
        if (MIX) {
            const unsigned xb = __builtin_bit_cast(unsigned, x);
            float m0, m1;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(m0) : "v"(xb), "v"(V[t]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(m1) : "v"(xb), "v"(V[t + 1]));
            const f32x2_t mm = {m0, m1};
            const h2 ym = __builtin_convertvector(mm, h2);
            hi[t] = x[0]; hi[t + 1] = x[1]; lo[t] = ym[0]; lo[t + 1] = ym[1];
            continue;
        }
        const float r0 = V[t] - (float)x[0], r1 = V[t + 1] - (float)x[1];
        const f32x2_t rr = {r0, r1};
        const h2 y = __builtin_convertvector(rr, h2);
        hi[t] = x[0]; hi[t + 1] = x[1]; lo[t] = y[0]; lo[t + 1] = y[1];
    }
}
// ReLU on the BIT PATTERN of the NEGATED value (one v_min_i32 instead of compare + select), NaN-propagating like torch.relu
// (common.h relu_).  k_cnet carries its activations negated: u = -t comes out of the epilogue's fma for free (negated row scale
// and bias tables), and as signed integers
//     u = -t with t > 0, t = +inf, or any NaN with its sign bit SET   ->  negative integer  ->  kept        (= -relu(t))
//     u = -t with t < 0 (incl. t = -inf)                              ->  positive integer  ->  +0
// so nrelu_bits(u) = -relu(t).  The negation then rides along for free: -h1 on the B side gives -(W2 h1) in the next
// accumulator, whose epilogue again wants the negated value; the last stage multiplies by a negated row scale.  Every step is
// an exact sign flip of what the positive chain computes (round-to-nearest is symmetric), so the results are the same bits.
// NaNs: gfx950 GENERATES 0xffc00000 -- sign bit set -- on both the matrix pipe and the VALU (inf - inf, inf * 0, and NaN
// operands in; scripts/ubench/epi_split.hip, measured), and NaNs that come in from memory are canonicalised to that pattern on
// the way in (canon_nan), so every NaN inside the kernel has its sign bit set and survives.
__device__ __forceinline__ float nrelu_bits(float u) { return __int_as_float(min(__float_as_int(u), 0)); }
__device__ __forceinline__ float canon_nan(float v) { return v != v ? __uint_as_float(0xffc00000u) : v; }
// Weight image of one convolution as the A operand: half [plane][Kp/8][M][8] (as above), then M floats rowscale, then M floats
// bias:  out_scaled[o] = acc * rowscale[o] + bias[o]  is the layer output times SH2_ACT_SCALE (f.0, f.2: rowscale = 2^-e,
// bias = b' * 16) or the plain value (f.4 rows: rowscale = 2^-e / 16, no bias).
__host__ __device__ static inline size_t sh2_image_bytes(int Kp, int M) { return (size_t)2 * Kp * M * sizeof(_Float16) + (size_t)2 * M * sizeof(float); }
__host__ __device__ static inline size_t sh2_rowscale_off(int Kp, int M) { return (size_t)2 * Kp * M * sizeof(_Float16); }

// k-PERMUTED images (RepackJob::kperm; f.2 and f.4 of the layers that run on k_cnet / k_cnet1w).  The accumulator block of a
// v_mfma_f32_32x32x16_f16 holds, per lane, rows 8 g + 4 kl + t (g = 0..3, t = 0..3, kl = lane / 32) of one pixel column; the next
// layer's B fragment wants 8 consecutive k of that column per lane and k-step.  A contraction may visit its k in any order as long
// as both operands agree, so inside every 32-k block the image's position (k-step s, 8-wide group kl, element j) holds
//     k = 16 s + 8 (j / 4) + 4 kl + j % 4            <=>  accumulator register 8 s + j of lane group kl
// and registers 8 s .. 8 s + 7 of an accumulator block ARE the B fragment of k-step s (k_cnet1w: h1 / h2 never leave the register
// file; k_cnet stores them to LDS in the same order).  sh2_kperm_src(P): the original k of image position P (any multiple of 4
// positions maps to 4 consecutive k).
__host__ __device__ static inline int sh2_kperm_src(int P) {
    const int b = P >> 5, s = (P >> 4) & 1, kl = (P >> 3) & 1, j = P & 7;
    return 32 * b + 16 * s + 8 * (j >> 2) + 4 * kl + (j & 3);
}

// ---- the whole coupling network f() = f.0 -> f.2 -> f.4 as ONE kernel + a light finishing kernel (cnet_sh.hip) --------
bool cnet_supported(int Cin, int H, int W, int hidden, int Cout);
int cnet_g0(int Cin);                 // 8-wide k groups of the f.0 image (even count)
int cnet_mpad4(int Cout);             // rows of the taps-as-rows f.4 image (multiple of 32) of ONE group of Cout output channels
int cnet_groups(int Cout);            // f.4 runs in that many groups of Cout / groups output channels (1 up to 56 channels; 0: unsupported)
size_t cnet_w4_bytes(int hidden, int Cout);   // all groups' images, one after the other
size_t cnet_scratch_floats(int N, int H, int W, int Cout);   // partial-sum scratch (floats) for batch N; <= N * the per-sample bound
size_t cnet_scratch_floats_per_sample(int H, int W, int Cout);
// What a k_cnet launch leaves behind for whoever finishes the step -- the finishing kernel, or the NEXT step's k_cnet while it
// builds its window: the partial sums of h = f(z1) in `scratch`, how they are tiled, and the coupling they feed.
struct CnetPending {
    const float* scratch; int MS, tiles, R, NI, lpxt;
    const float* bias; const float* scale;       // f.4 bias (Cout), exp(3 logs) (Cout) of that step
    int mode, Cout;                              // TailMode of that step's coupling
    const float* z; long z_bs;                   // the state that step read: z1 = channels [0, C/2), z2 = [C/2, C)
    int one_wave;                                // the launch ran on k_cnet1w (cnet1w_sh.hip); run-time evidence for the tests
    int finished;                                // the launch finished the step itself (CnetArgs::fin_cnt): no finishing kernel to launch
};
// channel mixer applied to a finished state (C = 0: none)
//   forward: the NEXT step's  y = M ((z + bias) * scale);   reverse: THIS step's  x = (M z) * scale - bias
struct CnetMixer { int C, reverse; const float* bias; const float* scale; const float* matrix; const int32_t* gather; };
struct CnetArgs {
    const float* x; long x_bs;                   // z1: channels [0, Cin) of (N, *, H, W), batch stride x_bs (unused with `pre`)
    const void* w0; const void* w2; const void* w4;   // SH2 images (REPACK_SH2_FIRST / _GEMM / _TAIL)
    int N, Cin, H, W, hidden, Cout;
    float* scratch;                              // cnet_scratch_floats(N, ...) floats
    // ---- finishing kernel of THIS step: coupling + log-det + channel mixer
    const float* bias; const float* scale;       // f.4 bias (Cout), exp(3 logs) (Cout)
    int mode;                                    // TailMode: the four coupling modes
    const float* z_in; long z_in_bs;             // (N, C, H, W): z1 = channels [0, C/2), z2 = [C/2, C)
    float* z_out; long z_out_bs;                 // result (may be z_in: a workgroup reads all of its own pixels before it writes)
    unsigned long long* acc;
    CnetMixer mix;                               // applied by this step's finishing kernel (mix.C = 0: only z2 is updated)
    // ---- optional: finish the PREVIOUS step while the window is built (one launch per FlowStep instead of two).  The
    // workgroup applies `pre`'s coupling and `pre_mix` to every window pixel (its own and the halo), takes z1 of the result as its
    // f.0 input and writes the result of its OWN pixels to pre_z_new (a buffer other than pre.z: neighbours still read that).
    int pre_on; CnetPending pre; CnetMixer pre_mix; float* pre_z_new; long pre_z_new_bs;
    // ---- optional: the training tape (plan_train.hip).  k_cnet stores h1 = relu(f.0 ...) and h2 = relu(f.2 ...) as FP16
    // (N, hidden, H, W) from its epilogues (the pointers are typed float* for the backward launch, which stores fp32 through them); the finishing kernel stores hout = (f.4 + bias) * exp(3 logs) as (N, Cout, H, W) and,
    // when it mixes out of place (z_out != z_in), ALSO writes the coupled z2 back into z_in -- which then holds this step's
    // output (y1, z2'), the tensor the backward sweep reads.
    float* tape_h1; float* tape_h2; float* tape_hout;
    unsigned short* mask1; unsigned short* mask2;     // sign bits of h1 / h2: [hidden / 32][N H W][2] 16-bit words (k_cnet MODE 1 writes, 2 reads)
    float in_scale, out_scale;                        // taping / backward launches: window values * in_scale, stored tensors * out_scale
    // ---- backward launch (bwd = 1; plan_train.hip): x = d L / d(f.4 output) (N, Cin = f.4's Cout, H, W); w0 / w2 / w4 = the SH2
    // images of f.4's, f.2's, f.0's TRANSPOSED weights (exp(3 logs) of f.2 / f.0 folded into the first two); tape_h1 <- g_u2,
    // tape_h2 <- g_u0 (fp32 (N, hidden, H, W)); the partial sums in `scratch` are d L / d y1's contribution (Cout = C/2 channels),
    // added into the gradient by k_chanmix_bwd (backward.h ChanMixBwdArgs::add_part)
    int bwd;
    // ---- optional: FUSED FINISHING.  fin_cnt != null: z_out / acc / mix above are valid at launch time and the launch may finish
    // the step itself -- every workgroup publishes its partial sums, then bumps the arrival counters fin_cnt[tile] of its tile and
    // of the neighbouring tiles of the image (they need its halo rows); whoever brings a counter to its full count finishes THAT tile
    // (coupling + log-det + mixer, the finishing kernel's own code) and resets the counter.  Nobody waits for anybody.  fin_cnt: one
    // zeroed word per tile.  CnetPending::finished tells the caller whether the launch took the offer.
    unsigned* fin_cnt;
};
int launch_cnet_main(const CnetArgs& a, hipStream_t s, CnetPending* out);            // k_cnet only; *out describes its partial sums
int launch_cnet_finish(const CnetArgs& a, const CnetPending& p, hipStream_t s);     // the finishing kernel for those sums
int launch_cnet(const CnetArgs& a, hipStream_t s);                                   // both
bool cnet_tape_supported(int Cin, int H, int W, int hidden, int Cout, int N);        // a taping instance of k_cnet exists for the shape
bool cnet_chain_enabled();   // testing hook (off by default: measured slower)
bool cnet_pre_supported(int Cin, int H, int W, int hidden, int Cout, int C);         // window-time finishing fits the LDS
void cnet_force(int ms, int flags);   // testing hook: ms in {0 (automatic), 1, 2, 4}

// ---- FlowSteps of the deep levels (C >= 192, a few hundred pixels per launch): one launch per LAYER, rows split over workgroups,
// activations between the launches as ready-made SH2 B operands in L2 (dnet_sh.hip) -------------------------------------------
constexpr int DNET_KS_MAX = 8;                 // largest K split of the f.4 launch (partial-sum copies in the scratch)
struct DnetLevel {
    int N, C, H, W, hidden, Cout;
    float* state; long state_bs;               // (N, C, H, W) fp32: the level's running state, updated in place
    void* scratch;                             // N * dnet_scratch_bytes_per_sample(...) bytes
};
bool dnet_supported(int C, int H, int W, int hidden, int Cout);
size_t dnet_scratch_bytes_per_sample(int C, int H, int W, int hidden, int Cout);
int dnet_level_begin(const DnetLevel& L, hipStream_t s);      // zero the padded operands' borders (once per level and call)
// first launch of a level: forward u = SH2((src + an_bias) * an_scale); reverse: src's first C/2 channels as the padded F0 operand;
// both copy src into L.state when it lives elsewhere
int dnet_prep(const DnetLevel& L, const float* src, long src_bs, const float* an_bias, const float* an_scale, int reverse, hipStream_t s);
// MIX: state <- W u (forward; w_image = SH2_GEMM image of W) or (W^-1 u) * post_scale - post_bias (reverse); want_pad: also the
// first C/2 channels as the padded operand of the next F0
int dnet_mix(const DnetLevel& L, const void* w_image, const float* post_scale, const float* post_bias, int want_pad, hipStream_t s);
// F0 -> F2 -> F4 on the padded z1 operand; *ks_out = number of K-split partial copies F4 left in the scratch
int dnet_coupling_net(const DnetLevel& L, const void* w0_sh2_first, const void* w2_sh2_gemm, const void* w4_sh2_first, int* ks_out, hipStream_t s);
// FIN: coupling (mode = TailMode) on L.state in place + log-det; want_u: the next MIX's operand, with the next step's ActNorm
// (u_bias / u_scale, forward) or plain (null, reverse)
int dnet_finish(const DnetLevel& L, int ks, const float* f4_bias, const float* f4_scale, int mode, unsigned long long* acc,
                const float* u_bias, const float* u_scale, int want_u, hipStream_t s);

}  // namespace glowhip
