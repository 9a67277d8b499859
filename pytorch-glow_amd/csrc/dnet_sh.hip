// dnet_sh.hip -- FlowSteps of the DEEP levels (C >= 192: 8x8 / 4x4 / 2x2 pixels per image; network/model.py:82-154 at the shapes
// network/model.py:242-261 produces for L >= 5) on split-half SH2 operands (sh.h).
//
// k_cnet (cnet_sh.hip) keeps a pixel tile resident and streams ALL weights of the coupling network through every workgroup:
// right when a launch has tens of thousands of pixels, hopeless when it has 256 (config E's last level: 16 images of 4x4
// pixels against 11.7 MB of weight images per FlowStep).  Here the roles are swapped: a launch is ONE layer, its rows (output
// channels) are split over workgroups, every weight is fetched by exactly one row tile, and the small activation tensors travel
// between the launches through L2 as ready-made B operands:
//
//   plain  SH2 tensor:  half [plane][K/8][P][8]            P = N*H*W pixels               (B of a 1x1 layer)
//   padded SH2 tensor:  half [plane][K/8][N*(H+2)*(W+2)][8]  zero border around every image (B of a 3x3 layer: the fragment
//                                                                                         of tap (dy, dx) is the SAME read shifted
//                                                                                         by dy*(W+2)+dx slots; im2col never exists)
// Per FlowStep, forward (reverse runs the same five kernels in the mirrored order):
//   MIX   y = W u                   u = SH2((z + b) * exp(3 logs)) from the previous FIN / PREP; writes the fp32 state and y1 padded
//   F0    h1 = relu(W0' * y1 + b0') 3x3 as implicit GEMM over (chunk, tap) k groups (the f.0 image of k_cnet) -> h1 plain
//   F2    h2 = relu(W2' h1 + b2')   -> h2 padded
//   F4    K-split partial sums of the 3x3 convolution 512 -> Cout as a DIRECT implicit GEMM (the same image kind as f.0: at 4x4
//         pixels the taps-as-rows form of k_cnet would have to exchange T rows between workgroups)
//   FIN   sum of the partials, (h + bias) * exp(3 logs), coupling, per-sample log-det (Q31.32), next step's ActNorm -> u
// The invertible 1x1 convolution itself (C x C over P pixels, network/module.py:359-369) is a genuine GEMM at these widths
// (AI = C/4 F/B >= 48: SURVEY 8a R5) and runs on the same kernel.  Every product is the 3-MFMA SH2 form, accumulation fp32.
#include "sh.h"
#include "conv_mfma.h"

namespace glowhip {

struct DnGemmArgs {
    const _Float16* A; int M, Kg;             // SH2 image: half [plane][Kg][M][8], then M row scales, M biases
    const _Float16* B; long b_plane; long b_slots;   // activation planes (halfs per plane), slots per k group
    int taps;                                  // 0: plain B (slot = pixel); 1: padded B, k group g = (chunk g / 9, tap g % 9)
    int nchunk_b;                              // taps: real 8-channel chunks of B (groups beyond 9 * nchunk_b carry zero weights)
    int N, H, W, P;                            // P = N * H * W
    int ksplit;                                // K split over blockIdx.z (EPI_PARTIAL only)
    int epi;                                   // DN_EPI_*
    _Float16* out_sh; long out_plane; long out_slots; int out_padded; int out_rows_sh;   // SH2 output (rows < out_rows_sh)
    float* out_f32; long out_bs;               // EPI_PARTIAL: [ks][N][M][HW]; EPI_MIX: state (N, *, H, W) with batch stride out_bs
    const float* post_scale; const float* post_bias;     // EPI_MIX reverse: y * post_scale[row] - post_bias[row]
};
enum { DN_EPI_ACT = 0, DN_EPI_PARTIAL = 1, DN_EPI_MIX = 2 };

// pixel p -> slot of the padded layout for tap (0, 0) (= the window's top-left corner; the pixel itself is at + (W+2) + 1)
__device__ __forceinline__ long dn_pad_slot(int p, int HW, int W) {
    const int n = p / HW, q = p - n * HW, y = q / W, x = q - y * W;
    return (long)n * ((HW / W + 2) * (W + 2)) + y * (W + 2) + x;
}

// NW waves; tile = RT x PT MFMA tiles of 32 rows x 32 pixels; the waves split the k-steps (two 8-wide k groups each) and their
// partial tiles are added through LDS in a fixed order.  Operands come straight from L2: the launch is a few microseconds of
// latency-bound work whatever is done, what matters is that each weight is read by one workgroup row only.
template <int NW, int RT, int PT, int NPF>
__global__ void __launch_bounds__(NW * 64) k_dn_gemm(DnGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][RT * PT][16][64]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, kl = lane >> 5, ml = lane & 31;
    const int HW = a.H * a.W, WP = a.W + 2;
    const int p0 = blockIdx.x * (32 * PT), r0 = blockIdx.y * (32 * RT);
    const int S = a.Kg >> 1;                               // k-steps
    const int s_lo = (int)((long)S * blockIdx.z / a.ksplit), s_hi = (int)((long)S * (blockIdx.z + 1) / a.ksplit);
    const long a_plane = (long)a.Kg * a.M * 8;
    const _Float16* ap = a.A + (long)(r0 + ml) * 8;        // + (g * M + i * 32) * 8
    long bslot[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int p = min(p0 + j * 32 + ml, a.P - 1);      // (clamped: lanes past the last pixel read a valid one and store nothing)
        bslot[j] = a.taps ? dn_pad_slot(p, HW, a.W) : (long)p;
    }
    f32x16_t acc[RT][PT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Operand sets NPF k-steps ahead, in NAMED register sets (the loop is unrolled by NPF).  A wave has only a handful of k-steps and
    // the weights come from HBM (every weight of a forward is read once: 580 MB at config E, nothing of it stays in a cache): with
    // one set in flight the launch was a chain of memory round trips (13 per wave in F0 at C = 384).  Every load is issued
    // unconditionally from a clamped (valid) step, so the waits are counted; only the MFMAs of a step past the wave's last are skipped.
    h8 Af[NPF][2 * RT], Bf[NPF][2 * PT];
    const int n_mine = max(0, (s_hi - s_lo - wid + NW - 1) / NW);       // k-steps of this wave: s_lo + wid + t * NW
    auto step_of = [&](int t) { return min(s_lo + wid + min(t, max(n_mine - 1, 0)) * NW, S - 1); };
    auto load = [&](int s, h8 (&A)[2 * RT], h8 (&B)[2 * PT]) {
        const int g = 2 * s + kl;
        const _Float16* pa = ap + (long)g * a.M * 8;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            A[i] = *reinterpret_cast<const h8*>(pa + i * 256);
            A[RT + i] = *reinterpret_cast<const h8*>(pa + i * 256 + a_plane);
        }
        long goff;
        if (a.taps) {
            const int ch = g / 9, tap = g - ch * 9, dy = tap / 3, dx = tap - dy * 3;
            goff = ch < a.nchunk_b ? (long)ch * a.b_slots + dy * WP + dx : -1;
        } else {
            goff = (long)g * a.b_slots;
        }
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            // (a padding group reads slot 0 of chunk 0: the zero corner of the first image -- finite, and multiplied by zero weights)
            const _Float16* pb = a.B + (goff < 0 ? 0 : goff + bslot[j]) * 8;
            B[j] = *reinterpret_cast<const h8*>(pb);
            B[PT + j] = *reinterpret_cast<const h8*>(pb + a.b_plane);
        }
    };
#pragma unroll
    for (int d = 0; d < NPF; ++d) load(step_of(d), Af[d], Bf[d]);
    // row scale / bias of the wave's first epilogue item: requested here, so that their round trip is not paid after the barrier
    const float* rowscale = reinterpret_cast<const float*>(a.A + 2 * a_plane);
    const float* rbias = rowscale + a.M;
    f32x4_t rs_first, bb_first;
    {
        const int t = wid >> 2, gq = wid & 3, i = t / PT;
        const int row = min(r0 + i * 32 + 8 * gq + 4 * kl, a.M - 4);
        rs_first = *reinterpret_cast<const f32x4_t*>(rowscale + row);
        bb_first = *reinterpret_cast<const f32x4_t*>(rbias + row);
    }
    for (int t0 = 0; t0 < n_mine; t0 += NPF) {
#pragma unroll
        for (int d = 0; d < NPF; ++d) {
            if (t0 + d < n_mine) {         // (wave-uniform)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[d][i], Bf[d][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[d][i], Bf[d][PT + j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[d][RT + i], Bf[d][j], acc[i][j], 0, 0, 0);
            }
            load(step_of(t0 + d + NPF), Af[d], Bf[d]);
        }
    }
    // ---- the waves' partial tiles -> LDS [wave][tile][register][lane] (lane-contiguous: conflict-free both ways)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wid * (RT * PT) + i * PT + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    // ---- items = (tile, group of four accumulator registers = four consecutive rows of one pixel per lane), wave w takes w, w + NW, ...
    for (int it = wid; it < RT * PT * 4; it += NW) {
        const int t = it >> 2, gq = it & 3, i = t / PT, j = t - i * PT;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; ++w)                       // fixed order: the same bits on every run
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += red[((w * (RT * PT) + t) * 16 + 4 * gq + q) * 64 + lane];
        const int row = r0 + i * 32 + 8 * gq + 4 * kl;     // first of the lane's four rows
        const int p = p0 + j * 32 + ml;
        if (p >= a.P || row >= a.M) continue;
        const f32x4_t rs = it == wid ? rs_first : *reinterpret_cast<const f32x4_t*>(rowscale + row);
        const int n = p / HW, q_ = p - n * HW;
        if (a.epi == DN_EPI_PARTIAL) {                     // true value of the partial sum (the activation scale undone)
            float* o = a.out_f32 + (((long)blockIdx.z * a.N + n) * a.M + row) * HW + q_;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[(long)q * HW] = v[q] * (rs[q] * SH2_ACT_INV);
            continue;
        }
        f32x4_t val;                                       // the layer output times SH2_ACT_SCALE
        if (a.epi == DN_EPI_ACT) {
            const f32x4_t bb = it == wid ? bb_first : *reinterpret_cast<const f32x4_t*>(rbias + row);
#pragma unroll
            for (int q = 0; q < 4; ++q) val[q] = relu_(fmaf(v[q], rs[q], bb[q]));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y = v[q] * (rs[q] * SH2_ACT_INV);
                if (a.post_scale) y = y * a.post_scale[row + q] - a.post_bias[row + q];
                a.out_f32[(long)n * a.out_bs + (long)(row + q) * HW + q_] = y;
                val[q] = y * SH2_ACT_SCALE;
            }
        }
        if (a.out_sh && row < a.out_rows_sh) {
            h4 hi, lo;
            sh2_split4<false>(val, hi, lo);
            const long slot = a.out_padded ? dn_pad_slot(p, HW, a.W) + WP + 1 : (long)p;
            _Float16* dst = a.out_sh + (((long)(row >> 3) * a.out_slots + slot) * 8 + 4 * kl);
            *reinterpret_cast<h4*>(dst) = hi;
            *reinterpret_cast<h4*>(dst + a.out_plane) = lo;
        }
    }
}

// ---- PREP: fp32 state -> SH2 operand of the first launch of a level (forward: u = (z + bias) * scale for MIX; reverse: the first
// C/2 channels as the padded input of F0), and the state copied into the level's own buffer when it lives elsewhere.
struct DnPrepArgs {
    const float* src; long src_bs; float* copy_to; long copy_bs;    // copy_to may be null (or == src: no copy)
    int C, Csh, N, H, W, P;                                         // Csh: leading channels that go to out_sh (multiple of 8)
    const float* bias; const float* scale;                          // ActNorm applied to the SH2 copy (null: none)
    _Float16* out_sh; long out_plane; long out_slots; int out_padded;
};
__global__ void __launch_bounds__(256) k_dn_prep(DnPrepArgs a) {
    const int HW = a.H * a.W, WP = a.W + 2;
    const int p = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c4 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;        // first of the thread's four channels
    if (p >= a.P || c4 >= a.C) return;
    const int n = p / HW, q_ = p - n * HW;
    f32x4_t v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = a.src[(long)n * a.src_bs + (long)(c4 + q) * HW + q_];
    if (a.copy_to && a.copy_to != a.src) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a.copy_to[(long)n * a.copy_bs + (long)(c4 + q) * HW + q_] = v[q];
    }
    if (c4 >= a.Csh) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (a.bias) v[q] = (v[q] + a.bias[c4 + q]) * a.scale[c4 + q];
        v[q] *= SH2_ACT_SCALE;
    }
    h4 hi, lo;
    sh2_split4<false>(v, hi, lo);
    const long slot = a.out_padded ? dn_pad_slot(p, HW, a.W) + WP + 1 : (long)p;
    _Float16* dst = a.out_sh + (((long)(c4 >> 3) * a.out_slots + slot) * 8 + (c4 & 4));
    *reinterpret_cast<h4*>(dst) = hi;
    *reinterpret_cast<h4*>(dst + a.out_plane) = lo;
}

// ---- FIN: the K-split partial sums of f.4 -> (h + bias) * exp(3 logs) -> coupling on the state in place -> per-sample log-det;
// then the SH2 operand of the next MIX: forward u = ActNorm_next(state), reverse u = state (the inverse ActNorm follows W^-1).
struct DnFinArgs {
    const float* part; int ks;                 // [ks][N][Cout][HW]
    const float* bias; const float* scale;     // f.4 bias, exp(3 logs)  (Cout)
    int mode;                                  // TailMode: the four coupling modes
    float* z; long z_bs;                       // state (N, C, H, W): z1 = channels [0, C/2), z2 = [C/2, C)
    int C, Cout, N, H, W, P;
    unsigned long long* acc;
    const float* u_bias; const float* u_scale; // ActNorm of the NEXT step for u (null: u = state)
    _Float16* u; long u_plane; long u_slots;   // null: no operand wanted (last step of the level, forward)
};
__global__ void __launch_bounds__(256) k_dn_fin(DnFinArgs a) {
    const int HW = a.H * a.W, Ch = a.C / 2;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 64 + lane;
    const int c4 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;        // first of the thread's four coupling channels
    const bool ok = p < a.P && c4 < Ch;
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    const int pc = ok ? p : 0, cc = ok ? c4 : 0;
    const int n = pc / HW, q_ = pc - n * HW;
    float* zp = a.z + (long)n * a.z_bs + q_;
    f32x4_t z1, z2;
    long long ldq = 0;
    float bad = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cc + q, ce = paired ? 2 * c : c, co = ce + (paired ? 1 : 0);
        float pe[DNET_KS_MAX], po[DNET_KS_MAX];             // every partial requested before the first is used (clamped, selected)
#pragma unroll
        for (int k = 0; k < DNET_KS_MAX; ++k) {
            const float* pp = a.part + (((long)min(k, a.ks - 1) * a.N + n) * a.Cout + ce) * HW + q_;
            pe[k] = pp[0];
            po[k] = pp[paired ? HW : 0];
        }
        float se = 0.f, so = 0.f;
#pragma unroll
        for (int k = 0; k < DNET_KS_MAX; ++k) {            // fixed order
            se += k < a.ks ? pe[k] : 0.f;
            so += k < a.ks ? po[k] : 0.f;
        }
        z1[q] = zp[(long)c * HW];
        const float zin = zp[(long)(Ch + c) * HW];
        const float A_ = (se + a.bias[ce]) * a.scale[ce];
        float zr;
        if (!paired) {
            zr = a.mode == TAIL_ADD_FWD ? zin + A_ : zin - A_;
        } else {                                           // network/model.py:108-113 / 134-139 (same arithmetic as cnet_sh.hip fin_apply_k)
            const float B_ = (so + a.bias[co]) * a.scale[co];
            const float sc = sigmoidf_(B_ + 2.0f);
            const float lg = logf(sc);
            zr = a.mode == TAIL_AFFINE_FWD ? (zin + A_) * sc : zin / sc - A_;
            const float term = a.mode == TAIL_AFFINE_FWD ? lg : -lg;
            if (!ok) { }
            else if (isfinite(term)) ldq += __double2ll_rn((double)term * FIX_SCALE);
            else bad = term;
        }
        z2[q] = zr;
        if (ok) zp[(long)(Ch + c) * HW] = zr;
    }
    if (paired) {
        if (ok && bad != 0.f) fix_flag_nonfinite(a.acc, n, a.N, bad);
        // per-sample sums inside the wave: a wave holds 64 consecutive pixels = whole images (HW <= 64) or a part of one
        const int seg = HW < 64 ? HW : 64;
        if ((seg & (seg - 1)) == 0 && (HW < 64 || HW % 64 == 0)) {
            for (int o = seg >> 1; o > 0; o >>= 1) ldq += __shfl_xor(ldq, o, 64);
            if (ok && (lane & (seg - 1)) == 0 && ldq != 0) {
                const int row = (blockIdx.y * 4 + (threadIdx.x >> 6)) % (1 + ACC_EXTRA);
                atomicAdd(a.acc + (row == 0 ? n : (long)(1 + row) * a.N + n), (unsigned long long)ldq);
            }
        } else if (ok && ldq != 0) {
            atomicAdd(a.acc + n, (unsigned long long)ldq);
        }
    }
    if (!ok || !a.u) return;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int c0 = half ? Ch + cc : cc;
        f32x4_t v = half ? z2 : z1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (a.u_bias) v[q] = (v[q] + a.u_bias[c0 + q]) * a.u_scale[c0 + q];
            v[q] *= SH2_ACT_SCALE;
        }
        h4 hi, lo;
        sh2_split4<false>(v, hi, lo);
        _Float16* dst = a.u + (((long)(c0 >> 3) * a.u_slots + p) * 8 + (c0 & 4));
        *reinterpret_cast<h4*>(dst) = hi;
        *reinterpret_cast<h4*>(dst + a.u_plane) = lo;
    }
}

// ================================================================================================ host side
bool dnet_supported(int C, int H, int W, int hidden, int Cout) {
    // rows in whole 32-row tiles, k groups in whole k-steps, channel quads inside one 8-channel chunk
    return C % 32 == 0 && (C / 2) % 8 == 0 && Cout % 32 == 0 && hidden % 32 == 0 && H >= 1 && W >= 1 && H * W <= 4096;
}

static int dn_ksplit(int M, int P, int S) {      // K split of F4: all workgroups resident at once (64-row x 64-pixel tiles take 128 KB of
                                                 // LDS: one per CU; the smaller ones two), >= 4 k-steps per wave
    const bool big = P >= 512 && M % 64 == 0;
    const int tiles = ((M + (big ? 63 : 31)) / (big ? 64 : 32)) * ((P + 63) / 64);
    const int cap = big ? 256 : 512;
    int ks = 1;
    while (ks < DNET_KS_MAX && tiles * ks * 2 <= cap && S / (ks * 2) >= 32) ks *= 2;
    return ks;
}

size_t dnet_scratch_bytes_per_sample(int C, int H, int W, int hidden, int Cout) {
    const size_t HW = (size_t)H * W, SL = (size_t)(H + 2) * (W + 2);
    return 4 * ((size_t)C * HW + (size_t)(C / 2) * SL + (size_t)hidden * HW + (size_t)hidden * SL + (size_t)DNET_KS_MAX * Cout * HW) + 8 * 256;
}

static int dn_launch_gemm(const DnGemmArgs& a, hipStream_t s) {
    // 32-row tiles unless the launch has pixels to spare (every row tile re-reads the B operand from L2, every weight is read once
    // either way): more workgroups, and -- with fewer registers per operand set -- more k-steps of weights in flight per wave
    const bool pt2 = a.P > 32, rt2 = a.M % 64 == 0 && a.P >= 512;
    const dim3 grid((a.P + (pt2 ? 63 : 31)) / (pt2 ? 64 : 32), a.M / (rt2 ? 64 : 32), a.ksplit);
#define DN_GO(RT, PT, NPF)                                                                                                          \
    {                                                                                                                               \
        const size_t lds = (size_t)8 * RT * PT * 16 * 64 * sizeof(float);                                                           \
        static bool attr_set = false;      /* (per instance: the attribute sticks to the function) */                              \
        if (lds > 32 * 1024 && !attr_set) {                                                                                         \
            (void)hipFuncSetAttribute((const void*)k_dn_gemm<8, RT, PT, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                                        \
        }                                                                                                                           \
        hipLaunchKernelGGL((k_dn_gemm<8, RT, PT, NPF>), grid, dim3(512), lds, s, a);                                                \
    }
    if (rt2 && pt2) DN_GO(2, 2, 4) else if (pt2) DN_GO(1, 2, 6) else DN_GO(1, 1, 8)
#undef DN_GO
    GH_LAUNCH_CHECK("k_dn_gemm");
    return GLOWHIP_OK;
}

// carve the level's scratch (all offsets 256-byte aligned)
struct DnScratch { _Float16* u; _Float16* ypad; _Float16* h1; _Float16* h2pad; float* part; long u_plane, y_plane, h1_plane, h2_plane, SLN; };
static DnScratch dn_carve(const DnetLevel& L, void* base) {
    const long HW = (long)L.H * L.W, P = (long)L.N * HW, SLN = (long)L.N * (L.H + 2) * (L.W + 2);
    DnScratch d;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) / 256 * 256; return r; };
    d.u_plane = (long)L.C * P; d.y_plane = (long)(L.C / 2) * SLN; d.h1_plane = (long)L.hidden * P; d.h2_plane = (long)L.hidden * SLN;
    d.SLN = SLN;
    d.u = (_Float16*)take((size_t)4 * L.C * P);
    d.ypad = (_Float16*)take((size_t)4 * (L.C / 2) * SLN);
    d.h1 = (_Float16*)take((size_t)4 * L.hidden * P);
    d.h2pad = (_Float16*)take((size_t)4 * L.hidden * SLN);
    d.part = (float*)take((size_t)4 * DNET_KS_MAX * L.Cout * P);
    return d;
}

int dnet_level_begin(const DnetLevel& L, hipStream_t s) {
    // the padded tensors' borders are zero for the whole level: nothing ever writes them
    const DnScratch d = dn_carve(L, L.scratch);
    if (hipMemsetAsync(d.ypad, 0, (size_t)4 * (L.C / 2) * d.SLN, s) != hipSuccess ||
        hipMemsetAsync(d.h2pad, 0, (size_t)4 * L.hidden * d.SLN, s) != hipSuccess) {
        set_error("dnet: hipMemsetAsync of the padded operands failed");
        return GLOWHIP_ELAUNCH;
    }
    return GLOWHIP_OK;
}

int dnet_prep(const DnetLevel& L, const float* src, long src_bs, const float* an_bias, const float* an_scale, int reverse, hipStream_t s) {
    const DnScratch d = dn_carve(L, L.scratch);
    const int P = L.N * L.H * L.W;
    DnPrepArgs a{};
    a.src = src; a.src_bs = src_bs; a.copy_to = L.state; a.copy_bs = L.state_bs;
    a.C = L.C; a.N = L.N; a.H = L.H; a.W = L.W; a.P = P;
    if (!reverse) { a.Csh = L.C; a.bias = an_bias; a.scale = an_scale; a.out_sh = d.u; a.out_plane = d.u_plane; a.out_slots = P; a.out_padded = 0; }
    else { a.Csh = L.C / 2; a.out_sh = d.ypad; a.out_plane = d.y_plane; a.out_slots = d.SLN; a.out_padded = 1; }
    hipLaunchKernelGGL(k_dn_prep, dim3((P + 63) / 64, (L.C / 4 + 3) / 4), dim3(256), 0, s, a);
    GH_LAUNCH_CHECK("k_dn_prep");
    return GLOWHIP_OK;
}

// MIX of one step: forward y = W u (+ y1 padded for F0); reverse x = (W^-1 u) * exp(-3 logs) - bias (+ x1 padded for the next
// executed step's F0 when `want_pad`)
int dnet_mix(const DnetLevel& L, const void* w_image, const float* post_scale, const float* post_bias, int want_pad, hipStream_t s) {
    const DnScratch d = dn_carve(L, L.scratch);
    const int P = L.N * L.H * L.W;
    DnGemmArgs g{};
    g.A = (const _Float16*)w_image; g.M = L.C; g.Kg = L.C / 8;
    g.B = d.u; g.b_plane = d.u_plane; g.b_slots = P; g.taps = 0;
    g.N = L.N; g.H = L.H; g.W = L.W; g.P = P; g.ksplit = 1; g.epi = DN_EPI_MIX;
    g.out_sh = want_pad ? d.ypad : nullptr; g.out_plane = d.y_plane; g.out_slots = d.SLN; g.out_padded = 1; g.out_rows_sh = L.C / 2;
    g.out_f32 = L.state; g.out_bs = L.state_bs; g.post_scale = post_scale; g.post_bias = post_bias;
    return dn_launch_gemm(g, s);
}

// F0 -> F2 -> F4 of one step: partial sums of h = f(z1) from the padded z1 operand
int dnet_coupling_net(const DnetLevel& L, const void* w0, const void* w2, const void* w4, int* ks_out, hipStream_t s) {
    const DnScratch d = dn_carve(L, L.scratch);
    const int P = L.N * L.H * L.W, Ch = L.C / 2;
    DnGemmArgs g{};
    g.N = L.N; g.H = L.H; g.W = L.W; g.P = P; g.ksplit = 1;
    // F0: 3x3, C/2 -> hidden, ActNorm + ReLU folded / fused
    g.A = (const _Float16*)w0; g.M = L.hidden; g.Kg = cnet_g0(Ch);
    g.B = d.ypad; g.b_plane = d.y_plane; g.b_slots = d.SLN; g.taps = 1; g.nchunk_b = Ch / 8;
    g.epi = DN_EPI_ACT; g.out_sh = d.h1; g.out_plane = d.h1_plane; g.out_slots = P; g.out_padded = 0; g.out_rows_sh = L.hidden;
    GH_TRY(dn_launch_gemm(g, s));
    // F2: 1x1, hidden -> hidden
    g.A = (const _Float16*)w2; g.Kg = L.hidden / 8;
    g.B = d.h1; g.b_plane = d.h1_plane; g.b_slots = P; g.taps = 0;
    g.out_sh = d.h2pad; g.out_plane = d.h2_plane; g.out_slots = d.SLN; g.out_padded = 1;
    GH_TRY(dn_launch_gemm(g, s));
    // F4: 3x3, hidden -> Cout, K split over workgroups
    g.A = (const _Float16*)w4; g.M = L.Cout; g.Kg = cnet_g0(L.hidden);
    g.B = d.h2pad; g.b_plane = d.h2_plane; g.b_slots = d.SLN; g.taps = 1; g.nchunk_b = L.hidden / 8;
    g.epi = DN_EPI_PARTIAL; g.out_sh = nullptr; g.out_f32 = d.part;
    g.ksplit = dn_ksplit(L.Cout, P, g.Kg / 2);
    *ks_out = g.ksplit;
    return dn_launch_gemm(g, s);
}

int dnet_finish(const DnetLevel& L, int ks, const float* f4_bias, const float* f4_scale, int mode, unsigned long long* acc,
                const float* u_bias, const float* u_scale, int want_u, hipStream_t s) {
    const DnScratch d = dn_carve(L, L.scratch);
    const int P = L.N * L.H * L.W;
    DnFinArgs a{};
    a.part = d.part; a.ks = ks; a.bias = f4_bias; a.scale = f4_scale; a.mode = mode;
    a.z = L.state; a.z_bs = L.state_bs; a.C = L.C; a.Cout = L.Cout; a.N = L.N; a.H = L.H; a.W = L.W; a.P = P; a.acc = acc;
    a.u_bias = u_bias; a.u_scale = u_scale; a.u = want_u ? d.u : nullptr; a.u_plane = d.u_plane; a.u_slots = P;
    hipLaunchKernelGGL(k_dn_fin, dim3((P + 63) / 64, (L.C / 8 + 3) / 4), dim3(256), 0, s, a);
    GH_LAUNCH_CHECK("k_dn_fin");
    return GLOWHIP_OK;
}

}  // namespace glowhip
