// wgrad_mfma.hip -- weight gradients of the coupling network's convolutions on the matrix cores.
//
// All three are one "NT" GEMM with the reduction over pixels:  C[m][n] = sum_{img,p} A[img][m][p] * B[img][n][p]
//   f.2 (1x1):  A = g_u2 (512 rows),            B = h1 (512 rows)                     -> dW2[o][i]
//   f.4 (3x3):  A = shift-expanded g_pre rows (o*9+tap) : g[o][p - d(tap)],  B = h2   -> dW4[o][i][tap]
//   f.0 (3x3):  A = g_u0 (512 rows),  B = shift-expanded y1 rows (i*9+tap) : y1[i][p + d(tap)] -> dW0[o][i][tap]
// The 3x3 cases expand the SMALL operand (12..48 resp. 6..24 channels -> x9 rows, a few tens of MB) with a cheap
// gather kernel instead of teaching the GEMM about taps.  The GEMM is k_conv_wide's K-major LDS / 32x32x2 MFMA
// loop with a transposing operand loader (rows are pixel-contiguous in HBM, the MFMA wants rows across lanes);
// the pixel axis is split over workgroups (split-K), partial tiles go to a scratch buffer and a second kernel adds
// them in a fixed order (deterministic) while scattering into the reference weight layout.
#include "backward.h"
#include "sh.h"
#include "wgrad_reduce.h"
#include <type_traits>

GH_STAMPS_DEFINE(wgrad)

namespace glowhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[img][c*9 + tap][p] = src[img][c][p + sign*d(tap)] (0 outside the image); rows >= C*9 up to rows_pad are zero.
// sign = +1: operand of a forward-style access (x[p + d]); sign = -1: gradient operand (g[p - d]).
__global__ void __launch_bounds__(256) k_shift_expand(const float* __restrict__ src, long src_bs, float* __restrict__ out,
                                                      int C, int H, int W, int rows_pad, int sign) {
    const int HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int row = blockIdx.y;
    const long n = blockIdx.z;
    if (p >= HW) return;
    float v = 0.f;
    if (row < C * 9) {
        const int c = row / 9, tap = row - c * 9;
        const int dy = (tap / 3 - 1) * sign, dx = (tap % 3 - 1) * sign;
        const int y = p / W + dy, x = p % W + dx;
        if (y >= 0 && y < H && x >= 0 && x < W) v = src[n * src_bs + (long)c * HW + y * W + x];
    }
    out[(n * rows_pad + row) * HW + p] = v;
}

int launch_shift_expand(const float* src, long src_bs, float* out, int N, int C, int H, int W, int rows_pad, int sign,
                        hipStream_t s) {
    if (N == 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_shift_expand, dim3(cdiv(H * W, 256), rows_pad, N), dim3(256), 0, s, src, src_bs, out, C, H, W,
                       rows_pad, sign);
    GH_LAUNCH_CHECK("k_shift_expand");
    return GLOWHIP_OK;
}

// partial[split][m][n] for one BM x BN tile and one slice of the pixel axis
template <int BN>
__global__ void __launch_bounds__(256)
k_wgrad_gemm(const float* __restrict__ A, long a_bs, const float* __restrict__ B, long b_bs, float* __restrict__ partial,
             int HW, int Mpad, int Npad, int ktiles_total, int ktiles_per_split) {
    constexpr int BM = 128, BK = 32;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_F4 = BM * BK / 4 / 256, B_F4 = BN * BK / 4 / 256;   // float4 per thread per K-tile (4, 4|2)
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, kl = lane >> 5, ml = lane & 31;
    // XCD-aware order (8 XCDs with private L2s, block b lands on XCD b % 8): the tiles of ONE pixel slice share their operand
    // panels -- every A panel is read by all tile_n, every B panel by all tile_m -- so they go to the same XCD back to back and
    // the panels travel from HBM once per slice instead of once per tile (the tile-major grid spread a slice over all 8 L2s:
    // 4x the operand bytes from memory at 512 x 512).
    const int tiles_n = Npad / BN;
    const int ntiles = (Mpad / BM) * tiles_n;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntiles, tile = logical - split * ntiles;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int kt0 = split * ktiles_per_split;
    const int kt1 = min(ktiles_total, kt0 + ktiles_per_split);
    const int tiles_per_img = HW / BK;

    // loader coordinates: each thread owns one row of the tile and 16 (A) / 16|8 (B) consecutive pixels of it
    const int a_row = tid & (BM - 1), a_q = tid / BM;                 // a_q in {0,1}: pixel quads [a_q*4, a_q*4+4)
    const int b_row = tid & (BN - 1), b_q = tid / BN;                 // BN=128: {0,1} x 4 quads; BN=64: {0..3} x 2 quads
    f32x4 ra[A_F4], rb[B_F4];
    auto load_tile = [&](int kt) {
        const int img = kt / tiles_per_img, p0 = (kt - img * tiles_per_img) * BK;
        const float* ap = A + (long)img * a_bs + (long)(tile_m * BM + a_row) * HW + p0 + a_q * (A_F4 * 4);
        const float* bp = B + (long)img * b_bs + (long)(tile_n * BN + b_row) * HW + p0 + b_q * (B_F4 * 4);
#pragma unroll
        for (int j = 0; j < A_F4; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ap + j * 4);
#pragma unroll
        for (int j = 0; j < B_F4; ++j) rb[j] = *reinterpret_cast<const f32x4*>(bp + j * 4);
    };
    auto store_tile = [&](int buf) {   // transpose: K-major LDS image, lanes = consecutive rows => conflict-free
#pragma unroll
        for (int j = 0; j < A_F4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) As[buf][a_q * (A_F4 * 4) + j * 4 + e][a_row] = ra[j][e];
#pragma unroll
        for (int j = 0; j < B_F4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[buf][b_q * (B_F4 * 4) + j * 4 + e][b_row] = rb[j][e];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1) load_tile(kt + 1);
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = As[buf][kl][wr * WM + i * 32 + ml];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Bs[buf][kl][wc * WN + j * 32 + ml];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < BK / 2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = As[buf][(kk + 1) * 2 + kl][wr * WM + i * 32 + ml];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = Bs[buf][(kk + 1) * 2 + kl][wc * WN + j * 32 + ml];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                if (kk + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
            if (kt + 1 < kt1) store_tile(buf ^ 1);
            __syncthreads();
        }
    }
    // partial tile: C[row = m][col = n]
    float* out = partial + ((long)split * Mpad + tile_m * BM) * Npad + tile_n * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                out[(long)row * Npad + wc * WN + j * 32 + ml] = acc[i][j][r];
            }
}


// The same GEMM on the f16 matrix pipe (sh.h, two-accumulator split-half form): both operands are converted to (hi, lo * 2^11)
// fp16 pairs on their way into LDS -- a thread's 8 consecutive pixels of a row are exactly one 16-byte MFMA fragment group, so
// the "transposing" store of the fp32 kernel (16 scalar LDS stores per thread and tile) becomes 4 vector stores -- and every
// product is three v_mfma_f32_32x32x16_f16.  24 MFMAs of 32 cycles per 32-pixel k-tile instead of 64 of 64 cycles: the kernel
// goes from MFMA-bound to operand-delivery-bound.  The gradient operand (A) is multiplied by a_scale = 2^k on the way in (it is
// ~1e-6 .. 1e-10 early in training, below fp16's normal range) and the partial sums by 2^-k on the way out (exact).
// VA / VB: operand A / B is VIRTUAL -- the shift-expanded rows (c * 9 + tap) : src[c][p + sign * d(tap)] of a (N, vC, vH, vW)
// tensor (rows >= 9 vC are zero), gathered by the loader itself (four 4-byte loads per chunk, clamped addresses, selected
// values) instead of being written out by k_shift_expand and read back (33 + 17 MB per level-1 FlowStep, 17 + 12 us of launches)
// BV: only b_valid rows of the plain operand B exist (an operand narrower than a column tile); the others are zero
// BH: operand B is an fp16 tensor (the hidden activations h1 / h2 as the taping k_cnet stores them: 2 bytes per value on the tape
// and in this loader); it goes into LDS as it is -- no lo plane, and the a.hi x b.lo product falls away (two MFMAs per k-step
// instead of three).  The gradient operand A keeps both planes: it is what needs the range.
// One GEMM's arguments (the kernel body below is shared by the one-GEMM launch and by the launch that runs f.4's and f.0's GEMMs
// side by side: k_wgrad_gemm_pair)
static int g_wgrad_narrow = 0;
void wgrad_force_narrow(int on) { g_wgrad_narrow = on; }
struct WgArgs {
    const float* A; long a_bs; const float* B; long b_bs; float* partial;
    int HW, Mpad, Npad, ktiles_total, ktiles_per_split; float a_scale, a_pre; double* rowsum;
    int vC, vH, vW, vsign, b_valid, tiled;
    int nblocks;       // workgroups of this GEMM (tiles x splits)
};
template <int BN, bool BH> constexpr int wgrad_sh_lds_bytes() {
    return (2 * 2 * 4 * (128 + 2) * 8 + 2 * (BH ? 1 : 2) * 4 * (BN + 2) * 8) * (int)sizeof(_Float16);
}
template <int BN, bool VA = false, bool VB = false, bool BV = false, bool BH = false>
__device__ __forceinline__ void wgrad_sh_body(const WgArgs& wa, const int block /* logical: tiles of a pixel slice adjacent */, char* lds) {
    const float* __restrict__ A = wa.A; const long a_bs = wa.a_bs; const float* __restrict__ B = wa.B; const long b_bs = wa.b_bs;
    float* __restrict__ partial = wa.partial;
    const int HW = wa.HW, Mpad = wa.Mpad, Npad = wa.Npad, ktiles_total = wa.ktiles_total, ktiles_per_split = wa.ktiles_per_split;
    const float a_scale = wa.a_scale, a_pre = wa.a_pre; double* __restrict__ rowsum = wa.rowsum;
    const int vC = wa.vC, vH = wa.vH, vW = wa.vW, vsign = wa.vsign, b_valid = wa.b_valid, tiled = wa.tiled;
    constexpr int BM = 128, BK = 32;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_F4 = BM * BK / 4 / 256, B_F4 = BN * BK / 4 / 256;   // float4 per thread per K-tile (4, 4|2)
    // [buffer][plane][k group][row][8], every k group 32 bytes longer than its rows: the loader's 8-byte stores of a quarter wave
    // are 2 rows x 4 k groups, and with a group stride of a whole number of 128-byte bank rounds the four groups of a row met in
    // the same 4 banks (SQ_LDS_BANK_CONFLICT: 0.6 of the LDS-active cycles); now each group has its own 8 banks.  Rows stay
    // contiguous: the fragment reads (32 rows x 16 bytes) are as before.
    constexpr int GPAD = 2;                      // (rows of padding per k group)
    static_assert(wgrad_sh_lds_bytes<BN, BH>() == (int)((2 * 2 * (BK / 8) * (BM + GPAD) * 8 + 2 * (BH ? 1 : 2) * (BK / 8) * (BN + GPAD) * 8) * sizeof(_Float16)), "LDS size");
    typedef _Float16 (*AsT)[2][BK / 8][BM + GPAD][8];
    typedef _Float16 (*BsT)[BH ? 1 : 2][BK / 8][BN + GPAD][8];
    const AsT As = reinterpret_cast<AsT>(lds);
    const BsT Bs = reinterpret_cast<BsT>(lds + 2 * 2 * (BK / 8) * (BM + GPAD) * 8 * sizeof(_Float16));
    static_assert(!BH || (!VB && !BV), "an fp16 operand B is a plain one");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, kl = lane >> 5, ml = lane & 31;
    // XCD-aware order (8 XCDs with private L2s, block b lands on XCD b % 8): the tiles of ONE pixel slice share their operand
    // panels -- every A panel is read by all tile_n, every B panel by all tile_m -- so they go to the same XCD back to back and
    // the panels travel from HBM once per slice instead of once per tile (the tile-major grid spread a slice over all 8 L2s:
    // 4x the operand bytes from memory at 512 x 512).
    const int tiles_n = Npad / BN;
    const int ntiles = (Mpad / BM) * tiles_n;
    const int logical = block;      // (the caller's xcd_remap of its block index)
    const int split = logical / ntiles, tile = logical - split * ntiles;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int kt0 = split * ktiles_per_split;
    const int kt1 = min(ktiles_total, kt0 + ktiles_per_split);
    const int tiles_per_img = HW / BK;
    // Operand loader: chunk q = tid + 256 j of a tile is (row q / 8, pixels 4 (q % 8) .. + 4) -- eight consecutive lanes read the
    // 128 contiguous bytes one row contributes to a k-tile, a wave instruction 8 whole cache lines.  (One row per lane, 64 bytes
    // each, asked the texture-address path for 64 partial lines per instruction: f.2's gradient at level 1 took 166 us for the
    // 67 us of HBM time its operands need.)
    const int l_row = tid >> 3, l_c = tid & 7;
    // A plain operand comes in one of two layouts: (N, rows, HW), or -- `tiled` bit 0 (A) / 1 (B), what the taping / backward
    // k_cnet write -- pixel-tile-major [pixel / 32][rows][pixel % 32] over the batch's pixels: a k-tile's 128-row panel is ONE
    // contiguous block (16 KB fp32 / 8 KB fp16; a wave request 1 KB / 512 B of consecutive addresses) instead of 128 runs of
    // 128 / 64 bytes HW * 4 bytes apart -- 32 K sequential streams per launch in DRAM's view, which the loads paid for with 40 of
    // the kernel's 93 us at level 1 (same kernel with no requests: 54 us).
    const bool a_t = tiled & 1, b_t = tiled & 2;
    const int a_rs = a_t ? BK : HW, b_rs = b_t ? BK : HW;            // row stride in elements
    const unsigned lane_off_a = (unsigned)(l_row * a_rs + l_c * 4), lane_off_b = (unsigned)(l_row * b_rs + l_c * 4);
    // One k-tile's operand chunks in registers.  THREE of them by name (s0 / s1 / s2, the k loop is unrolled by three): tile
    // t + 3 is requested while tile t is multiplied and tile t + 1 goes into LDS -- a request has two k-tiles' worth of MFMAs (of
    // two workgroups per CU) to come back in.  One stage, requested one tile ahead, left every k-tile waiting out most of its own
    // trip to memory (~3.5 k cycles per k-tile against 1 k of MFMA issue per SIMD at level 1).
    // (two stages where both operands are fp32 at 128 columns: 32 registers per stage next to the 128 accumulators)
    struct Stage { f32x4 ra[A_F4], rb[BH ? 1 : B_F4]; h4 rbh[BH ? B_F4 : 1]; };
    constexpr int NST = ((BN == 128 && !BH) || VA) ? 2 : 3;      // (a gathered A: 16 requests + their in-image masks per stage)
    // virtual operand: per chunk row (fixed over the k loop) the source offset c * HW + dy * vW + dx, dy, dx; off < 0: a zero row
    constexpr int VN = VA ? A_F4 : (VB ? B_F4 : 1);
    int v_off[VN], v_dy[VN], v_dx[VN];
    if (VA || VB) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            const int r = (VA ? tile_m * BM : tile_n * BN) + l_row + 32 * j;
            const int c = r / 9, tap = r - c * 9;
            v_dy[j] = (tap / 3 - 1) * vsign; v_dx[j] = (tap % 3 - 1) * vsign;
            v_off[j] = r < 9 * vC ? c * HW + v_dy[j] * vW + v_dx[j] : -(1 << 30);
        }
    }
    const int vlw = (VA || VB) ? __builtin_ctz(vW) : 0;
    auto load_virtual = [&](const float* V, long v_bs, int img, int p0, f32x4* dst, int j) {
        const int p = p0 + l_c * 4, y = p >> vlw, x = p & (vW - 1);
        const float* vb = V + (long)img * v_bs + p;
        {
            const int yy = y + v_dy[j];
            const bool rowok = v_off[j] > -(1 << 29) && yy >= 0 && yy < vH;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xx = x + e + v_dx[j];
                const bool ok = rowok && xx >= 0 && xx < vW;
                const float v = ok ? vb[v_off[j] + e] : V[0];      // (unconditional load from a valid address, selected)
                dst[j][e] = ok ? v : 0.f;
            }
        }
    };
    // part q of a k-tile's requests: the thread's q-th row chunk of either operand (a k-tile = 4 parts; step() issues them
    // between its MFMA groups)
    auto load_part = [&](int kt, Stage& st, int q) {
        f32x4 (&ra)[A_F4] = st.ra; f32x4 (&rb)[BH ? 1 : B_F4] = st.rb; h4 (&rbh)[BH ? B_F4 : 1] = st.rbh;
        (void)ra; (void)rb; (void)rbh;
#ifdef WG_ABL_NOLOAD        // (timing experiments only: nothing is requested from memory)
        if constexpr (!VA && BH) {
            ra[q] = f32x4{(float)kt, 1.f, 2.f, 3.f}; rbh[q] = h4{(_Float16)kt, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
            return;
        }
#endif
        const int img = kt / tiles_per_img, p0 = (kt - img * tiles_per_img) * BK;
        if constexpr (VA) {
            load_virtual(A, a_bs, img, p0, ra, q);
        } else {
            // (uniform base + the thread's 32-bit offset: no 64-bit address arithmetic per request)
            const char* ap = reinterpret_cast<const char*>(A + (a_t ? ((long)kt * Mpad + tile_m * BM) * BK : (long)img * a_bs + (long)tile_m * BM * HW + p0));
            ra[q] = *reinterpret_cast<const f32x4*>(ap + (long)q * 32 * a_rs * 4 + lane_off_a * 4u);
        }
        if (q >= B_F4) return;
        if constexpr (BH) {
            const char* bp = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(B) +
                                                           (b_t ? ((long)kt * Npad + tile_n * BN) * BK : (long)img * b_bs + (long)tile_n * BN * HW + p0));
            rbh[q] = *reinterpret_cast<const h4*>(bp + (long)q * 32 * b_rs * 2 + lane_off_b * 2u);
        } else if constexpr (VB) {
            load_virtual(B, b_bs, img, p0, rb, q);
        } else if constexpr (BV) {      // rows >= b_valid do not exist in the tensor: loaded from row 0, zeroed
            const int row = tile_n * BN + l_row + 32 * q;
            const bool ok = row < b_valid;
            const f32x4 v = *reinterpret_cast<const f32x4*>(B + (long)img * b_bs + (long)(ok ? row : 0) * HW + p0 + l_c * 4);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rb[q] = ok ? v : z;
        } else {
            const char* bp = reinterpret_cast<const char*>(B + (b_t ? ((long)kt * Npad + tile_n * BN) * BK : (long)img * b_bs + (long)tile_n * BN * HW + p0));
            rb[q] = *reinterpret_cast<const f32x4*>(bp + (long)q * 32 * b_rs * 4 + lane_off_b * 4u);
        }
    };
    auto load_tile = [&](int kt, Stage& st) {
#pragma unroll
        for (int q = 0; q < A_F4; ++q) load_part(kt, st, q);
    };
    // (same bits as sh_split on v * pre: the residual t - hi is exact either way; here it is ONE v_fma_mix_f32 per value, which
    // reads hi as the f16 half it is, and the two scalings are packed multiplications: 3 VALU instructions per value instead of 5)
    // (a_pre: a_scale for a plain g; 2^-11 for the backward k_cnet's g_u0, which is stored times a_scale * 2^11 for the kernel
    // below.  The 2^11 on lo is what keeps the range: early in training g * a_scale is ~1e-5 and smaller (f.4 starts at zero),
    // where a true-scale lo would be gone and hi itself subnormal -- test_tiny_gradients_behind_near_zero_tail_weights_purely_relative.)
    auto split4 = [](const f32x4& v, float pre, h4& hi, h4& lo, auto ps) {
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            const f32x2_t vv = ps.value ? f32x2_t{v[t], v[t + 1]} : f32x2_t{v[t], v[t + 1]} * pre;
            const h2 x = __builtin_convertvector(vv, h2);
            const f32x2_t rr = f32x2_t{__builtin_fmaf((float)x[0], -1.0f, vv[0]), __builtin_fmaf((float)x[1], -1.0f, vv[1])} * SH_LO_SCALE;
            const h2 y = __builtin_convertvector(rr, h2);
            hi[t] = x[0]; hi[t + 1] = x[1]; lo[t] = y[0]; lo[t + 1] = y[1];
        }
    };
    float rsum[A_F4];                  // rowsum != null: sum over this slice's pixels of the thread's A chunks (the bias gradient)
    const bool do_rsum = rowsum && tile_n == 0;      // (only the first column tile's workgroups hand their sums over)
#pragma unroll
    for (int j = 0; j < A_F4; ++j) rsum[j] = 0.f;
    auto store_tile = [&](int buf, Stage& st) {   // 4 consecutive pixels of a row = half a fragment group: 8-byte stores, 16 lanes = 128 contiguous bytes
        f32x4 (&ra)[A_F4] = st.ra; f32x4 (&rb)[BH ? 1 : B_F4] = st.rb; h4 (&rbh)[BH ? B_F4 : 1] = st.rbh;
        (void)ra; (void)rb; (void)rbh;
        if (do_rsum) {
#pragma unroll
            for (int j = 0; j < A_F4; ++j) rsum[j] += (ra[j][0] + ra[j][1]) + (ra[j][2] + ra[j][3]);
        }
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            h4 hi, lo;
#ifdef WG_ABL_NOCONV        // (timing experiments only: no split)
            hi = __builtin_bit_cast(h4, f32x2_t{ra[j][0], ra[j][1]}); lo = __builtin_bit_cast(h4, f32x2_t{ra[j][2], ra[j][3]});
#else
            split4(ra[j], a_pre, hi, lo, std::false_type{});
#endif
            *reinterpret_cast<h4*>(&As[buf][0][l_c >> 1][l_row + 32 * j][(l_c & 1) * 4]) = hi;
            *reinterpret_cast<h4*>(&As[buf][1][l_c >> 1][l_row + 32 * j][(l_c & 1) * 4]) = lo;
        }
        if constexpr (BH) {
#pragma unroll
            for (int j = 0; j < B_F4; ++j) *reinterpret_cast<h4*>(&Bs[buf][0][l_c >> 1][l_row + 32 * j][(l_c & 1) * 4]) = rbh[j];
        } else {
#pragma unroll
            for (int j = 0; j < B_F4; ++j) {
                h4 hi, lo;
                split4(rb[j], 1.0f, hi, lo, std::true_type{});
                *reinterpret_cast<h4*>(&Bs[buf][0][l_c >> 1][l_row + 32 * j][(l_c & 1) * 4]) = hi;
                *reinterpret_cast<h4*>(&Bs[buf][BH ? 0 : 1][l_c >> 1][l_row + 32 * j][(l_c & 1) * 4]) = lo;
            }
        }
    };

    f32x16_t accm[TM][TN], accx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    const int nk = kt1 - kt0;
    if (nk > 0) {
        Stage s0, s1, s2;
#ifdef WG_ABL_SAMETILE      // (timing experiments only: every request hits the same k-tile -- L2 hits)
        auto tile_of = [&](int t) { return kt0 + (min(t, nk - 1) & 1); };
#else
        auto tile_of = [&](int t) { return kt0 + min(t, nk - 1); };     // (clamped: the steady loop does not branch around its requests)
#endif
        auto multiply = [&](int buf, auto&& between) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int grp = 2 * ks + kl;
                h8 ah[TM], al[TM], bh[TN], bl[BH ? 1 : TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const h8*>(&As[buf][0][grp][wr * WM + i * 32 + ml][0]);
                    al[i] = *reinterpret_cast<const h8*>(&As[buf][1][grp][wr * WM + i * 32 + ml][0]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const h8*>(&Bs[buf][0][grp][wc * WN + j * 32 + ml][0]);
                    if constexpr (!BH) bl[j] = *reinterpret_cast<const h8*>(&Bs[buf][BH ? 0 : 1][grp][wc * WN + j * 32 + ml][0]);
                }
                between(2 * ks);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accm[i][j], 0, 0, 0);
                between(2 * ks + 1);
                if constexpr (!BH) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
            }
        };
        // step t: tile t is in LDS buffer t & 1; `done` held it and takes tile t + NST, `next` holds tile t + 1 and goes into LDS
        // The requests of tile t + NST go out in four parts BETWEEN the MFMA groups: issued in one burst at the head of the step
        // the 64 wave requests of a CU's eight waves queued up in its memory pipeline (56 B/clk: ~860 cycles for the 48 KB of two
        // k-tiles) and every wave sat behind its own requests before its first MFMA -- the requests cost 30 of 93 us at level 1
        // even when they all hit the L2.
        auto step = [&](int t, Stage& done, Stage& next, auto last) {
            const bool req = !last.value || t + NST < nk;
            const int ktn = tile_of(t + NST);
            constexpr bool SPREAD = BH || BN == 64;      // (the two-plane 128-column instances have no registers for it: 16 spilled)
            if (!SPREAD && req) { load_tile(ktn, done); __builtin_amdgcn_sched_barrier(0); }
#ifndef WG_ABL_NOMFMA
            multiply(t & 1, [&](int q) {
                if constexpr (SPREAD) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (req) load_part(ktn, done, q);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
#else
            if (SPREAD && req) load_tile(ktn, done);
#endif
            if (!last.value || t + 1 < nk) store_tile((t + 1) & 1, next);
            __syncthreads();
        };
        load_tile(tile_of(0), s0); load_tile(tile_of(1), s1);
        if constexpr (NST == 3) load_tile(tile_of(2), s2);
        store_tile(0, s0);
        __syncthreads();
        int t = 0;
        if constexpr (NST == 3) {
            for (; t + 4 <= nk; t += 3) {
                step(t, s0, s1, std::false_type{}); step(t + 1, s1, s2, std::false_type{}); step(t + 2, s2, s0, std::false_type{});
            }
            if (t < nk) step(t, s0, s1, std::true_type{});
            if (t + 1 < nk) step(t + 1, s1, s2, std::true_type{});
            if (t + 2 < nk) step(t + 2, s2, s0, std::true_type{});
        } else {
            for (; t + 3 <= nk; t += 2) { step(t, s0, s1, std::false_type{}); step(t + 1, s1, s0, std::false_type{}); }
            if (t < nk) step(t, s0, s1, std::true_type{});
            if (t + 1 < nk) step(t + 1, s1, s0, std::true_type{});
        }
    }
    if (do_rsum && nk > 0) {      // the eight chunks of a row sit in eight consecutive lanes: one fp64 atomic per row and slice
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            float v = rsum[j];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            if (l_c == 0) atomicAdd(rowsum + tile_m * BM + l_row + 32 * j, (double)v * (double)(a_pre / a_scale));      // (a pre-scaled A: back to g)
        }
    }
    const float inv = 1.0f / a_scale;
    float* out = partial + ((long)split * Mpad + tile_m * BM) * Npad + tile_n * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                out[(long)row * Npad + wc * WN + j * 32 + ml] = (accm[i][j][r] + accx[i][j][r] * SH_LO_INV) * inv;
            }
}

template <int BN, bool VA = false, bool VB = false, bool BV = false, bool BH = false>
__global__ void __launch_bounds__(256, 2)      // two workgroups per CU: at 269 registers (141 + 128 accumulators) the kernel ran ONE
                                               // wave per SIMD and every k-tile waited out its own HBM round trip (7.4 k cycles per
                                               // k-tile against 768 of MFMA work)
k_wgrad_gemm_sh(WgArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[wgrad_sh_lds_bytes<BN, BH>()];
    wgrad_sh_body<BN, VA, VB, BV, BH>(a, xcd_remap(blockIdx.x, gridDim.x), lds);
}

// f.4's GEMM (gathered A, fp16 B) and f.0's (plain A, gathered B with BN0 columns) of one FlowStep in ONE launch: the first
// a4.nblocks workgroups (a multiple of 8: xcd_remap) are f.4's.  Side by side the two fill the chip with half the pixel slices
// each -- and the partial tiles they write and k_wgrad_reduce_batched reads back are what those launches mostly cost
// (4 tiles x 128 slices x 64 KB = 32 MB for f.4 alone, at every level).
template <int BN0>
__global__ void __launch_bounds__(256, 2) k_wgrad_gemm_pair(WgArgs a4, WgArgs a0, int live4) {
    constexpr int L4 = wgrad_sh_lds_bytes<128, true>(), L0 = wgrad_sh_lds_bytes<BN0, false>();
    __shared__ __attribute__((aligned(16))) char lds[L4 > L0 ? L4 : L0];
    if ((int)blockIdx.x < a4.nblocks) {      // (a4.nblocks is live4 rounded up to a multiple of 8; the blocks in between have no tile)
        const int l = xcd_remap(blockIdx.x, a4.nblocks);
        if (l < live4) wgrad_sh_body<128, true, false, false, true>(a4, l, lds);
    } else {
        wgrad_sh_body<BN0, false, true, false, false>(a0, xcd_remap(blockIdx.x - a4.nblocks, a0.nblocks), lds);
    }
}

// The f.2 case of the kernel above -- plain A = the backward k_cnet's g_u2, stored times a_scale * 2^11 (PS), plain B = the fp16
// tape -- with the step rebuilt around the wave's own MFMA stream: see step().  (The template keeps the general kernel's
// parameters; the gathered / two-plane instances measured slower in this form -- a gathered value's in-image select pinned
// into the slot of its request waits out the request -- and stay on the kernel above.)
// NT = 512 (BN = 256 only): EIGHT waves -- two rows x four columns of the same 64 x 64 wave tiles -- share one 128-row A panel: see
// k_wgrad_gemm_ps512 below.
template <int BN, bool VA = false, bool VB = false, bool BV = false, bool BH = false, bool PS = false, int NT = 256>
__device__ __forceinline__ void wgrad_ps_body(const WgArgs& wa, const int block /* logical, as in wgrad_sh_body */, char* lds) {
    const float* __restrict__ A = wa.A; const long a_bs = wa.a_bs; const float* __restrict__ B = wa.B; const long b_bs = wa.b_bs;
    float* __restrict__ partial = wa.partial;
    const int HW = wa.HW, Mpad = wa.Mpad, Npad = wa.Npad, ktiles_total = wa.ktiles_total, ktiles_per_split = wa.ktiles_per_split;
    const float a_scale = wa.a_scale, a_pre = wa.a_pre; double* __restrict__ rowsum = wa.rowsum;
    const int vC = wa.vC, vH = wa.vH, vW = wa.vW, vsign = wa.vsign, b_valid = wa.b_valid, tiled = wa.tiled;
    (void)a_bs; (void)b_bs; (void)vC; (void)vH; (void)vW; (void)vsign; (void)b_valid;
    constexpr int BM = 128, BK = 32;
    constexpr int NWC = NT / 128;                 // columns of waves (two rows of them)
    constexpr int RPC = NT / 8;                   // rows of a loader chunk: eight threads per row
    constexpr int WM = BM / 2, WN = BN / NWC, TM = WM / 32, TN = WN / 32;
    constexpr int A_F4 = BM * BK / 4 / NT, B_F4 = BN * BK / 4 / NT;   // float4 per thread per K-tile (4, 4|2; eight waves: 2, 4)
    constexpr int NPARTS = A_F4 > B_F4 ? A_F4 : B_F4;                  // parts of a k-tile's requests: part q = chunk q of A (q < A_F4) and of B (q < B_F4)
    // [buffer][plane][k group][row][8], every k group 32 bytes longer than its rows: the loader's 8-byte stores of a quarter wave
    // are 2 rows x 4 k groups, and with a group stride of a whole number of 128-byte bank rounds the four groups of a row met in
    // the same 4 banks (SQ_LDS_BANK_CONFLICT: 0.6 of the LDS-active cycles); now each group has its own 8 banks.  Rows stay
    // contiguous: the fragment reads (32 rows x 16 bytes) are as before.
    constexpr int GPAD = 2;                      // (rows of padding per k group)
    typedef _Float16 (*AsT)[2][BK / 8][BM + GPAD][8];
    typedef _Float16 (*BsT)[BH ? 1 : 2][BK / 8][BN + GPAD][8];
    const AsT As = reinterpret_cast<AsT>(lds);
    const BsT Bs = reinterpret_cast<BsT>(lds + 2 * 2 * (BK / 8) * (BM + GPAD) * 8 * sizeof(_Float16));
    static_assert(!BH || (!VB && !BV), "an fp16 operand B is a plain one");
    static_assert(!PS || !VA, "a pre-scaled A is a plain one");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / NWC, wc = wid % NWC, kl = lane >> 5, ml = lane & 31;
    // XCD-aware order (8 XCDs with private L2s, block b lands on XCD b % 8): the tiles of ONE pixel slice share their operand
    // panels -- every A panel is read by all tile_n, every B panel by all tile_m -- so they go to the same XCD back to back and
    // the panels travel from HBM once per slice instead of once per tile (the tile-major grid spread a slice over all 8 L2s:
    // 4x the operand bytes from memory at 512 x 512).
    const int tiles_n = Npad / BN;
    const int ntiles = (Mpad / BM) * tiles_n;
    const int logical = block;
    const int split = logical / ntiles, tile = logical - split * ntiles;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int kt0 = split * ktiles_per_split;
    const int kt1 = min(ktiles_total, kt0 + ktiles_per_split);
    const int tiles_per_img = HW / BK;
    // Operand loader: chunk q = tid + 256 j of a tile is (row q / 8, pixels 4 (q % 8) .. + 4) -- eight consecutive lanes read the
    // 128 contiguous bytes one row contributes to a k-tile, a wave instruction 8 whole cache lines.  (One row per lane, 64 bytes
    // each, asked the texture-address path for 64 partial lines per instruction: f.2's gradient at level 1 took 166 us for the
    // 67 us of HBM time its operands need.)
    const int l_row = tid >> 3, l_c = tid & 7;
    // A plain operand comes in one of two layouts: (N, rows, HW), or -- `tiled` bit 0 (A) / 1 (B), what the taping / backward
    // k_cnet write -- pixel-tile-major [pixel / 32][rows][pixel % 32] over the batch's pixels: a k-tile's 128-row panel is ONE
    // contiguous block (16 KB fp32 / 8 KB fp16; a wave request 1 KB / 512 B of consecutive addresses) instead of 128 runs of
    // 128 / 64 bytes HW * 4 bytes apart -- 32 K sequential streams per launch in DRAM's view, which the loads paid for with 40 of
    // the kernel's 93 us at level 1 (same kernel with no requests: 54 us).
    const bool a_t = tiled & 1, b_t = tiled & 2;
    const int a_rs = a_t ? BK : HW, b_rs = b_t ? BK : HW;            // row stride in elements
    const unsigned lane_off_a = (unsigned)(l_row * a_rs + l_c * 4), lane_off_b = (unsigned)(l_row * b_rs + l_c * 4);
    // panel of k-tile kt = (image img, tile tin of the image): base + img * s_img + tin * s_tin elements, either layout (no
    // branch on the layout inside the k loop)
    const long a_img = a_t ? (long)tiles_per_img * Mpad * BK : a_bs, a_tin = a_t ? (long)Mpad * BK : BK;
    const long b_img = b_t ? (long)tiles_per_img * Npad * BK : b_bs, b_tin = b_t ? (long)Npad * BK : BK;
    const long a_base = (long)tile_m * BM * a_rs, b_base = (long)tile_n * BN * b_rs;
    // One k-tile's operand chunks in registers.  TWO of them by name (sx / sy, the k loop is unrolled by two), refilled chunk by
    // chunk: during step t the stage holding tile t + 1 is split and stored into LDS one chunk at a time, and each chunk's
    // registers are re-requested for tile t + 3 as soon as they have been read -- a request has two steps to come back in (the
    // wait in front of the split: 8 cycles per k-tile by the stamps), with the registers of two tiles.
    struct Stage { f32x4 ra[A_F4], rb[BH ? 1 : B_F4]; h4 rbh[BH ? B_F4 : 1]; };
    // virtual operand: per chunk row (fixed over the k loop) the source offset c * HW + dy * vW + dx, dy, dx; off < 0: a zero row
    constexpr int VN = VA ? A_F4 : (VB ? B_F4 : 1);
    int v_off[VN], v_dy[VN], v_dx[VN];
    if (VA || VB) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            const int r = (VA ? tile_m * BM : tile_n * BN) + l_row + RPC * j;
            const int c = r / 9, tap = r - c * 9;
            v_dy[j] = (tap / 3 - 1) * vsign; v_dx[j] = (tap % 3 - 1) * vsign;
            v_off[j] = r < 9 * vC ? c * HW + v_dy[j] * vW + v_dx[j] : -(1 << 30);
        }
    }
    const int vlw = (VA || VB) ? __builtin_ctz(vW) : 0;
    auto load_virtual = [&](const float* V, long v_bs, int img, int p0, f32x4* dst, int j) {
        const int p = p0 + l_c * 4, y = p >> vlw, x = p & (vW - 1);
        const float* vb = V + (long)img * v_bs + p;
        {
            const int yy = y + v_dy[j];
            const bool rowok = v_off[j] > -(1 << 29) && yy >= 0 && yy < vH;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xx = x + e + v_dx[j];
                const bool ok = rowok && xx >= 0 && xx < vW;
                const float v = ok ? vb[v_off[j] + e] : V[0];      // (unconditional load from a valid address, selected)
                dst[j][e] = ok ? v : 0.f;
            }
        }
    };
    // part q of a k-tile's requests: the thread's q-th row chunk of either operand (a k-tile = 4 parts; step() issues them
    // between its MFMA groups)
    auto load_part = [&](int kt, Stage& st, int q) {
        f32x4 (&ra)[A_F4] = st.ra; f32x4 (&rb)[BH ? 1 : B_F4] = st.rb; h4 (&rbh)[BH ? B_F4 : 1] = st.rbh;
        (void)ra; (void)rb; (void)rbh;
#ifdef WG_ABL_NOLOAD        // (timing experiments only: nothing is requested from memory)
        if constexpr (!VA && BH) {
            ra[q] = f32x4{(float)kt, 1.f, 2.f, 3.f}; rbh[q] = h4{(_Float16)kt, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
            return;
        }
#endif
        // (both operands are pixel-tile-major here -- launch_wgrad_mfma / _trio require it: a k-tile's panel is kt * rows * 32
        // elements in, no division by the tiles per image in front of the requests)
        const int img = VA || VB || BV ? kt / tiles_per_img : 0, tin = VA || VB || BV ? kt - img * tiles_per_img : 0, p0 = tin * BK;
        (void)p0;
        if (q >= A_F4) {
        } else if constexpr (VA) {
            load_virtual(A, a_bs, img, p0, ra, q);
        } else {
            // (uniform base + the thread's 32-bit offset: no 64-bit address arithmetic per request)
            const char* ap = reinterpret_cast<const char*>(A + a_base + (PS ? (long)kt * Mpad * BK : img * a_img + tin * a_tin));
            ra[q] = *reinterpret_cast<const f32x4*>(ap + (long)q * RPC * a_rs * 4 + lane_off_a * 4u);
        }
        if (q >= B_F4) return;
        if constexpr (BH) {
            const char* bp = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(B) + b_base + (PS ? (long)kt * Npad * BK : img * b_img + tin * b_tin));
            rbh[q] = *reinterpret_cast<const h4*>(bp + (long)q * RPC * b_rs * 2 + lane_off_b * 2u);
        } else if constexpr (VB) {
            load_virtual(B, b_bs, img, p0, rb, q);
        } else if constexpr (BV) {      // rows >= b_valid do not exist in the tensor: loaded from row 0, zeroed
            const int row = tile_n * BN + l_row + RPC * q;
            const bool ok = row < b_valid;
            const f32x4 v = *reinterpret_cast<const f32x4*>(B + (long)img * b_bs + (long)(ok ? row : 0) * HW + p0 + l_c * 4);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rb[q] = ok ? v : z;
        } else {
            const char* bp = reinterpret_cast<const char*>(B + b_base + img * b_img + tin * b_tin);
            rb[q] = *reinterpret_cast<const f32x4*>(bp + (long)q * RPC * b_rs * 4 + lane_off_b * 4u);
        }
    };
    auto load_tile = [&](int kt, Stage& st) {
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) load_part(kt, st, q);
    };
    // Split of one pair of values into (hi, lo * 2^11) halves: same bits as sh_split on v * pre (the residual t - hi is exact).
    // PS -- a plain A written by the backward k_cnet, which stores T = g * a_scale * 2^11: hi = fp16(T * 2^-11) and
    // lo = fp16(T - 2^11 hi), each ONE v_fma_mix{lo,hi}_f16 that rounds straight into its half of the packed word: 2 instructions
    // per value, none of them a packed fp32 operation (v_pk_mul_f32 / v_pk_add_f32 next to a running MFMA stream are the most
    // expensive VALU instructions there are: scripts/ubench/mfma_valu_shadow.hip).  The 2^11 on lo is what keeps the range: early
    // in training g * a_scale is ~1e-5 and smaller (f.4 starts at zero), where a true-scale lo would be gone and hi itself
    // subnormal (test_tiny_gradients_behind_near_zero_tail_weights_purely_relative).
    auto split2 = [](float v0, float v1, float pre, unsigned& hi, unsigned& lo, auto ps) {
        if constexpr (decltype(ps)::value) {
            const float c_dn = SH_LO_INV, c_up = -SH_LO_SCALE;
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "s"(c_dn));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "s"(c_dn));
            asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "s"(c_up), "v"(v0));
            asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "s"(c_up), "v"(v1));
        } else {
            const f32x2_t vv = f32x2_t{v0, v1} * pre;
            const h2 x = __builtin_convertvector(vv, h2);
            const f32x2_t rr = f32x2_t{__builtin_fmaf((float)x[0], -1.0f, vv[0]), __builtin_fmaf((float)x[1], -1.0f, vv[1])} * SH_LO_SCALE;
            hi = __builtin_bit_cast(unsigned, x);
            lo = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, h2));
        }
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float rsum[A_F4];                  // rowsum != null: sum over this slice's pixels of the thread's A chunks (the bias gradient)
    const bool do_rsum = rowsum && tile_n == 0;      // (only the first column tile's workgroups hand their sums over)
#pragma unroll
    for (int j = 0; j < A_F4; ++j) rsum[j] = 0.f;
    // The next tile's way into LDS in pieces (step() runs them between its MFMAs): pair h of A chunk q -> packed halves in
    // (ahi, alo); the chunk's two 8-byte stores (4 consecutive pixels of a row = half a fragment group, 16 lanes = 128 contiguous
    // bytes); B chunk q (fp16 tape: as it is; fp32: split like A, exact scale 1).
    u32x2 ahi, alo, bhi, blo;
    auto conv_a = [&](Stage& st, int q, int h) {
        if (q >= A_F4) return;
        if (h == 0) rsum[q] += (st.ra[q][0] + st.ra[q][1]) + (st.ra[q][2] + st.ra[q][3]);     // (every workgroup: a branch here would cut the step into blocks)
#ifdef WG_ABL_NOCONV        // (timing experiments only: no split)
        ahi[h] = __float_as_uint(st.ra[q][2 * h]); alo[h] = __float_as_uint(st.ra[q][2 * h + 1]);
#else
        unsigned x, y;
        split2(st.ra[q][2 * h], st.ra[q][2 * h + 1], a_pre, x, y, std::integral_constant<bool, PS>{});
        ahi[h] = x; alo[h] = y;
#endif
    };
    auto store_a = [&](int buf, int q) {
        if (q >= A_F4) return;
        *reinterpret_cast<u32x2*>(&As[buf][0][l_c >> 1][l_row + RPC * q][(l_c & 1) * 4]) = ahi;
        *reinterpret_cast<u32x2*>(&As[buf][1][l_c >> 1][l_row + RPC * q][(l_c & 1) * 4]) = alo;
    };
    auto conv_b = [&](Stage& st, int q, int h) {
        if constexpr (!BH) {
            if (q < B_F4) {
                unsigned x, y;
                split2(st.rb[q][2 * h], st.rb[q][2 * h + 1], 1.0f, x, y, std::false_type{});
                bhi[h] = x; blo[h] = y;
            }
        }
    };
    auto store_b = [&](int buf, Stage& st, int q) {
        if (q >= B_F4) return;
        if constexpr (BH) {
            *reinterpret_cast<h4*>(&Bs[buf][0][l_c >> 1][l_row + RPC * q][(l_c & 1) * 4]) = st.rbh[q];
        } else {
            *reinterpret_cast<u32x2*>(&Bs[buf][0][l_c >> 1][l_row + RPC * q][(l_c & 1) * 4]) = bhi;
            *reinterpret_cast<u32x2*>(&Bs[buf][BH ? 0 : 1][l_c >> 1][l_row + RPC * q][(l_c & 1) * 4]) = blo;
        }
    };
    auto store_tile = [&](int buf, Stage& st) {       // (the first tile: nothing to hide behind)
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) {
            conv_a(st, q, 0); conv_a(st, q, 1); store_a(buf, q);
            conv_b(st, q, 0); conv_b(st, q, 1); store_b(buf, st, q);
        }
    };

    f32x16_t accm[TM][TN], accx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    const int nk = kt1 - kt0;
    if (nk > 0) {
        Stage sx;
#ifdef WG_ABL_SAMETILE      // (timing experiments only: every request hits the same k-tile -- L2 hits)
        auto tile_of = [&](int t) { return kt0 + (min(t, nk - 1) & 1); };
#else
        auto tile_of = [&](int t) { return kt0 + min(t, nk - 1); };     // (clamped: the steady loop does not branch around its requests)
#endif
        // One step = the 16 (fp16 B) or 24 MFMAs of tile t from LDS buffer t & 1, and BETWEEN them, one piece after each MFMA, in
        // this order: the split of tile t + 1 (stage `cur`) and its stores into the other LDS buffer, each chunk followed by
        // the request of the same chunk of tile t + 3.  A wave that multiplied first and converted afterwards spent 1 310 + 1 030 cycles per k-tile on the two
        // (scripts/stamps_wgrad.py; 512 would be the MFMAs alone): its VALU work ran at ~1/3 of its rate next to the partner
        // wave's MFMA stream and its own MFMAs at ~40 %.  In the shadow of the wave's OWN MFMA (32 clocks on the pipe, 4 to issue)
        // up to six plain VALU instructions are free (ubench: 39 clocks per MFMA + 6 v_fma_f32).  sched_barrier pins the order.
        constexpr int NM = (BH ? 2 : 3) * TM * TN * (BK / 16), QM = NM / NPARTS;       // MFMAs per step, per part (= per chunk q)
        // (the gathered-A and two-plane instances at 128 columns have no registers for the pieces or a second stage -- they
        // spilled, and a spill's reload waits for every request in flight: they multiply first, then split + store tile t + 1
        // from their ONE stage and request tile t + 2 into it)
        constexpr bool SPREAD = BN == 64 || (BH && !VA);
        constexpr int AHEAD = SPREAD ? 3 : 2;
        static_assert(NM % NPARTS == 0 && QM >= 2, "pieces per chunk over its MFMA slots");
#ifdef GLOWHIP_DEBUG_STAMPS
        unsigned long long tph[5] = {0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define WG_PH(i) do { if (BH && !VA) { const unsigned long long c_ = __builtin_readcyclecounter(); tph[i] += c_ - tprev; tprev = c_; } } while (0)
#else
#define WG_PH(i) do { } while (0)
#endif
        auto step = [&](int t, Stage& cur, auto last) {
            const bool req = !last.value || t + AHEAD < nk, st_on = !last.value || t + 1 < nk;
            const int ktn = tile_of(t + AHEAD), buf = t & 1, nbuf = buf ^ 1;
            WG_PH(3);
            // fragment reads: a k-step's at the latest point that still gives them four MFMAs to arrive -- ah of step ks + 1 when
            // the last group using ah of ks is done, the B fragments one MFMA later, al at the head of ks + 1 (it is used last):
            // 32 fragment registers live at the peak instead of 48 (with 128 accumulators and three stages there are no more)
            h8 ah[BK / 16][TM], al[BK / 16][TM], bh[BK / 16][TN], bl[BK / 16][BH ? 1 : TN];
            auto rd_ah = [&](int ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) ah[ks][i] = *reinterpret_cast<const h8*>(&As[buf][0][2 * ks + kl][wr * WM + i * 32 + ml][0]);
            };
            auto rd_al = [&](int ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) al[ks][i] = *reinterpret_cast<const h8*>(&As[buf][1][2 * ks + kl][wr * WM + i * 32 + ml][0]);
            };
            auto rd_b = [&](int ks) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[ks][j] = *reinterpret_cast<const h8*>(&Bs[buf][0][2 * ks + kl][wc * WN + j * 32 + ml][0]);
                    if constexpr (!BH) bl[ks][j] = *reinterpret_cast<const h8*>(&Bs[buf][BH ? 0 : 1][2 * ks + kl][wc * WN + j * 32 + ml][0]);
                }
            };
            constexpr int GM = TM * TN, KM = NM / (BK / 16);            // MFMAs per group, per k-step
            constexpr int AH_DONE = (BH ? 1 : 2) * GM;                  // slot (within a k-step) after which ah is dead
            rd_ah(0); rd_b(0); rd_al(0);
            int m = 0;
            auto side = [&](int mm) {
                if constexpr (!SPREAD) return;
                const int q = mm / QM, r = mm - q * QM;
                const int ks = mm / KM, rk = mm - ks * KM;
                auto at = [&](int piece) { return (QM >= 4 ? piece : piece * QM / 4) == r; };      // four pieces over QM slots
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < BK / 16) {
                    if (rk == AH_DONE - 1) rd_ah(ks + 1);
                    if (rk == AH_DONE) rd_b(ks + 1);
                    if (rk == KM - 1) rd_al(ks + 1);
                }
                if (at(0)) { if (st_on) conv_a(cur, q, 0); }
                if (at(1)) { if (st_on) conv_a(cur, q, 1); }
                if (at(2)) { if (st_on) { store_a(nbuf, q); conv_b(cur, q, 0); } }
                if (at(3)) { if (st_on) { conv_b(cur, q, 1); store_b(nbuf, cur, q); } if (req) load_part(ktn, cur, q); }
                __builtin_amdgcn_sched_barrier(0);
            };
#ifndef WG_ABL_NOMFMA
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                if constexpr (!SPREAD) { if (ks > 0) { rd_ah(ks); rd_b(ks); rd_al(ks); } }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) { accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], accm[i][j], 0, 0, 0); side(m++); }
                if constexpr (!BH) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) { accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], accx[i][j], 0, 0, 0); side(m++); }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) { accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], accx[i][j], 0, 0, 0); side(m++); }
            }
#else
#pragma unroll
            for (int mm = 0; mm < NM; ++mm) side(mm);
#endif
            WG_PH(0);
            if constexpr (!SPREAD) {
                __builtin_amdgcn_sched_barrier(0);
                if (st_on) store_tile(nbuf, cur);
                if (req) load_tile(ktn, cur);
            }
            WG_PH(1);
            __syncthreads();
            WG_PH(2);
        };
        int t = 0;
        if constexpr (SPREAD) {
            Stage sy;
            load_tile(tile_of(0), sx); load_tile(tile_of(1), sy);
            store_tile(0, sx);
            load_tile(tile_of(2), sx);
            __syncthreads();
            // step t splits tile t + 1: odd tiles live in sy, even ones in sx
            for (; t + 3 <= nk; t += 2) { step(t, sy, std::false_type{}); step(t + 1, sx, std::false_type{}); }
            if (t < nk) step(t, sy, std::true_type{});
            if (t + 1 < nk) step(t + 1, sx, std::true_type{});
        } else {
            load_tile(tile_of(0), sx);
            store_tile(0, sx);
            load_tile(tile_of(1), sx);
            __syncthreads();
            for (; t + 2 <= nk; ++t) step(t, sx, std::false_type{});
            if (t < nk) step(t, sx, std::true_type{});
        }
#ifdef GLOWHIP_DEBUG_STAMPS
        if (BH && !VA) { GH_STAMP_VAL(0, tph[0]); GH_STAMP_VAL(1, tph[1]); GH_STAMP_VAL(2, tph[2]); GH_STAMP_VAL(3, tph[3]); GH_STAMP_VAL(5, tph[4]); GH_STAMP_VAL(4, nk);
                         GH_STAMP_VAL(63, __builtin_amdgcn_s_getreg((31 << 11) | 4)); }
#endif
    }
    if (do_rsum && nk > 0) {      // the eight chunks of a row sit in eight consecutive lanes: one fp64 atomic per row and slice
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            float v = rsum[j];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            if (l_c == 0) atomicAdd(rowsum + tile_m * BM + l_row + RPC * j, (double)v * (double)(PS ? SH_LO_INV / a_scale : a_pre / a_scale));      // (a pre-scaled A: back to g)
        }
    }
    const float inv = 1.0f / a_scale;
    float* out = partial + ((long)split * Mpad + tile_m * BM) * Npad + tile_n * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                out[(long)row * Npad + wc * WN + j * 32 + ml] = (accm[i][j][r] + accx[i][j][r] * SH_LO_INV) * inv;
            }
}

template <int BN>
__global__ void __launch_bounds__(256, 2) k_wgrad_gemm_ps(WgArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[wgrad_sh_lds_bytes<BN, true>()];
    wgrad_ps_body<BN, false, false, false, true, true>(a, xcd_remap(blockIdx.x, gridDim.x), lds);
}

// The same with 256 output columns per workgroup and EIGHT waves (two rows x four columns of 64 x 64 wave tiles, one workgroup per
// CU: the same two waves per SIMD as two 128 x 128 workgroups): at 128 x 128 every operand panel is fetched from L2 by the four
// workgroups of its tile row / column -- 800 MB of L2 -> LDS traffic for the 201 MB of f.2's level-1 operands, 10 TB/s over the
// launch, which is what it waits for; here the fp32 gradient panel is shared by twice as many waves: fetched twice, the fp16 tape
// panel four times -- 536 MB.  (66 KiB of LDS: dynamic.)
__global__ void __launch_bounds__(512, 1) k_wgrad_gemm_ps512(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds_dyn[];
    wgrad_ps_body<256, false, false, false, true, true, 512>(a, xcd_remap(blockIdx.x, gridDim.x), lds_dyn);
}

// All three weight-gradient GEMMs of a FlowStep behind the backward k_cnet in ONE launch: blocks [0, a2.nblocks) are f.2's
// (wgrad_ps_body), the next a4.nblocks f.4's, the rest f.0's (both counts multiples of 8: xcd_remap; live2 / live4 of them have
// a tile).  What k_wgrad_gemm_pair did for two: ~512 workgroups shared out by MFMA work, each GEMM cut into as few pixel slices
// as its share allows -- 30 MB of partial tiles per level-1 step instead of 80 when each GEMM filled the chip alone.
template <int BN0>
__global__ void __launch_bounds__(256, 2) k_wgrad_gemm_trio(WgArgs a2, WgArgs a4, WgArgs a0, int live2, int live4) {
    constexpr int L2 = wgrad_sh_lds_bytes<128, true>(), L0 = wgrad_sh_lds_bytes<BN0, false>();
    __shared__ __attribute__((aligned(16))) char lds[L2 > L0 ? L2 : L0];
    const int b = blockIdx.x;
    if (b < a2.nblocks) {
        const int l = xcd_remap(b, a2.nblocks);
        if (l < live2) wgrad_ps_body<128, false, false, false, true, true>(a2, l, lds);
    } else if (b < a2.nblocks + a4.nblocks) {
        const int l = xcd_remap(b - a2.nblocks, a4.nblocks);
        if (l < live4) wgrad_sh_body<128, true, false, false, true>(a4, l, lds);
    } else {
        wgrad_sh_body<BN0, false, true, false, false>(a0, xcd_remap(b - a2.nblocks - a4.nblocks, a0.nblocks), lds);
    }
}

__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, int splits,
                                                      int Mpad, int Npad, int Mreal, int Nreal, int mode) {
    wgrad_reduce_body(partial, dw, splits, Mpad, Npad, Mreal, Nreal, mode, blockIdx.x);
}

// the reductions of up to three weight-gradient GEMMs in ONE launch (blockIdx.y = job): a FlowStep's three were three launches
// of ~7 us each, mostly launch latency
__global__ void __launch_bounds__(256) k_wgrad_reduce_batched(WgradReduceJobs j) {
    const WgradReduceJob& r = j.job[blockIdx.y];
    wgrad_reduce_body(r.partial, r.dw, r.splits, r.Mpad, r.Npad, r.Mreal, r.Nreal, r.mode, blockIdx.x);
}

int launch_wgrad_reduce_batched(const WgradReduceJobs& j, hipStream_t s) {
    if (j.n == 0) return GLOWHIP_OK;
    long blocks = 1;
    for (int i = 0; i < j.n; ++i) blocks = std::max(blocks, (long)cdiv((long)j.job[i].Mreal * ((j.job[i].Nreal + 3) / 4), 256));
    hipLaunchKernelGGL(k_wgrad_reduce_batched, dim3((unsigned)blocks, j.n), dim3(256), 0, s, j);
    GH_LAUNCH_CHECK("k_wgrad_reduce_batched");
    return GLOWHIP_OK;
}

bool wgrad_mfma_supported(int HW, int Mpad, int Npad) { return HW % 32 == 0 && Mpad % 128 == 0 && Npad % 64 == 0; }
static bool t1_ok(const WgradTaps& t, int HW) { return t.H * t.W == HW && t.W >= 4 && (t.W & (t.W - 1)) == 0; }

size_t wgrad_mfma_partial_floats(int Mpad, int Npad, int N, int HW) {
    const int tiles = (Mpad / 128) * (Npad % 128 == 0 ? Npad / 128 : Npad / 64);
    const int total = (int)((long)N * HW / 32);
    int splits = std::max(1, std::min(total, (512 + tiles - 1) / tiles));
    return (size_t)splits * Mpad * Npad;
}

bool wgrad_pair_ok(int HW, int m4, int hid, int n0) {
    return wgrad_mfma_supported(HW, m4, hid) && wgrad_mfma_supported(HW, hid, n0) && hid % 128 == 0 && (n0 == 64 || n0 % 128 == 0);
}

// f.4's and f.0's weight-gradient GEMMs of one FlowStep (both behind the backward k_cnet, both with a gathered 3x3 operand) as ONE
// launch (k_wgrad_gemm_pair): the ~512 workgroups that fill the chip are shared out by MFMA work, so each GEMM is cut into half
// as many pixel slices as it would take alone -- half the partial tiles to write and to reduce.  The caller asks wgrad_pair_ok first;
// shapes outside it are an error here (GLOWHIP_EINVAL, nothing launched).
//   f.4: A = gathered g_pre (t4: Cout x 9 rows -> m4), B = h2 fp16 (hid rows);   f.0: A = g_u0 (hid rows, pre-scaled), B = gathered y1 (t0 -> n0)
int launch_wgrad_pair(const float* gpre, long gpre_bs, const void* h2_half, float* partial4, float* dw4, int m4, int m4_real,
                      const float* gu0, const float* y1, long y1_bs, float* partial0, float* dw0, int n0, int n0_real,
                      int N, int HW, int hid, float sh_scale, double* rowsum0, const WgradTaps& t4, const WgradTaps& t0, int tiled4,
                      int tiled0, WgradReduceJob* rj4, WgradReduceJob* rj0, hipStream_t s) {
    const bool ok = sh_scale > 0.f && N > 0 && wgrad_pair_ok(HW, m4, hid, n0) && t4.operand == 0 && t1_ok(t4, HW) && t0.operand == 1 && t1_ok(t0, HW) &&
                    tiled4 == 2 && tiled0 == 5;
    GH_REQUIRE(ok, "wgrad grouped launch: shapes outside wgrad_pair_ok / operands not as documented");
    const int total = (int)((long)N * HW / 32);
    const int bn0 = n0 == 64 ? 64 : 128;
    const int tiles4 = (m4 / 128) * (hid / 128), tiles0 = (hid / 128) * (n0 / bn0);
    // MFMAs per k-tile: f.4 two per product (fp16 B), f.0 three
    const double w4 = 2.0 * m4 * hid, w0 = 3.0 * hid * n0;
    auto slices = [&](int tiles, double share, int* per) {
        int sp = std::max(1, std::min(total, (int)(512.0 * share / tiles + 0.5)));
        *per = (total + sp - 1) / sp;
        return (total + *per - 1) / *per;
    };
    int per4, per0;
    const int sp4 = slices(tiles4, w4 / (w4 + w0), &per4), sp0 = slices(tiles0, w0 / (w4 + w0), &per0);
    const int nb4 = (tiles4 * sp4 + 7) / 8 * 8, nb0 = tiles0 * sp0;      // (f.0's blocks start on XCD 0: xcd_remap)
    WgArgs a4{gpre, gpre_bs, reinterpret_cast<const float*>(h2_half), (long)hid * HW, partial4, HW, m4, hid, total, per4, sh_scale, sh_scale, nullptr,
              t4.C, t4.H, t4.W, t4.sign, hid, tiled4, tiles4 * sp4};
    WgArgs a0{gu0, (long)hid * HW, y1, y1_bs, partial0, HW, hid, n0, total, per0, sh_scale, SH_LO_INV, rowsum0,
              t0.C, t0.H, t0.W, t0.sign, n0, tiled0, nb0};
    // (blocks tiles4 * sp4 .. nb4 of f.4's range have no tile: the body's split index runs past its k-tiles and they only write
    // nothing -- see the guard in the wrapper)
    a4.nblocks = nb4;
    if (bn0 == 64) hipLaunchKernelGGL(k_wgrad_gemm_pair<64>, dim3(nb4 + nb0), dim3(256), 0, s, a4, a0, tiles4 * sp4);
    else hipLaunchKernelGGL(k_wgrad_gemm_pair<128>, dim3(nb4 + nb0), dim3(256), 0, s, a4, a0, tiles4 * sp4);
    GH_LAUNCH_CHECK("k_wgrad_gemm_pair");
    *rj4 = WgradReduceJob{partial4, dw4, sp4, m4, hid, m4_real, hid, 1};
    *rj0 = WgradReduceJob{partial0, dw0, sp0, hid, n0, hid, n0_real, 0};
    return GLOWHIP_OK;
}

// The same with f.2's GEMM (A = g_u2, B = h1 fp16, both pixel-tile-major; wgrad_ps_body) as the third: k_wgrad_gemm_trio.
int launch_wgrad_trio(const float* gu2, const void* h1_half, float* partial2, float* dw2, double* rowsum2,
                      const float* gpre, long gpre_bs, const void* h2_half, float* partial4, float* dw4, int m4, int m4_real,
                      const float* gu0, const float* y1, long y1_bs, float* partial0, float* dw0, int n0, int n0_real,
                      int N, int HW, int hid, float sh_scale, double* rowsum0, const WgradTaps& t4, const WgradTaps& t0,
                      WgradReduceJob* rj2, WgradReduceJob* rj4, WgradReduceJob* rj0, hipStream_t s) {
    const bool ok = sh_scale > 0.f && N > 0 && wgrad_pair_ok(HW, m4, hid, n0) && t4.operand == 0 && t1_ok(t4, HW) && t0.operand == 1 && t1_ok(t0, HW);
    GH_REQUIRE(ok, "wgrad grouped launch: shapes outside wgrad_pair_ok / operands not as documented");
    const int total = (int)((long)N * HW / 32);
    const int bn0 = n0 == 64 ? 64 : 128;
    const int tiles2 = (hid / 128) * (hid / 128), tiles4 = (m4 / 128) * (hid / 128), tiles0 = (hid / 128) * (n0 / bn0);
    // Shares by MFMAs per k-tile: two per product with an fp16 B (f.2, f.4), three for f.0.  (By measured cost per (workgroup,
    // k-tile) -- 1.37 us for f.2's kernel, 2.4 - 2.7 us for the gathering ones -- the launch got slower: f.2's steps slow down
    // next to gathering workgroups on the same CU.  With long pixel axes (level 1 of config B: 2 048 k-tiles) the launch is slower
    // than f.2 and the pair one after the other either way, 185 vs 87 + 73 us: the caller takes it only for short ones.)
    const double w2 = 2.0 * hid * hid, w4 = 2.0 * m4 * hid, w0 = 3.0 * hid * n0, wsum = w2 + w4 + w0;
    auto slices = [&](int tiles, double share, int* per) {
        int sp = std::max(1, std::min(total, (int)(512.0 * share / tiles + 0.5)));
        *per = (total + sp - 1) / sp;
        return (total + *per - 1) / *per;
    };
    int per2, per4, per0;
    const int sp2 = slices(tiles2, w2 / wsum, &per2), sp4 = slices(tiles4, w4 / wsum, &per4), sp0 = slices(tiles0, w0 / wsum, &per0);
    const int live2 = tiles2 * sp2, live4 = tiles4 * sp4, nb2 = (live2 + 7) / 8 * 8, nb4 = (live4 + 7) / 8 * 8, nb0 = tiles0 * sp0;
    const long hb = (long)hid * HW;
    WgArgs a2{gu2, hb, reinterpret_cast<const float*>(h1_half), hb, partial2, HW, hid, hid, total, per2, sh_scale, SH_LO_INV, rowsum2,
              0, 0, 1, 0, hid, 7, nb2};
    WgArgs a4{gpre, gpre_bs, reinterpret_cast<const float*>(h2_half), hb, partial4, HW, m4, hid, total, per4, sh_scale, sh_scale, nullptr,
              t4.C, t4.H, t4.W, t4.sign, hid, 2, nb4};
    WgArgs a0{gu0, hb, y1, y1_bs, partial0, HW, hid, n0, total, per0, sh_scale, SH_LO_INV, rowsum0,
              t0.C, t0.H, t0.W, t0.sign, n0, 5, nb0};
    if (bn0 == 64) hipLaunchKernelGGL(k_wgrad_gemm_trio<64>, dim3(nb2 + nb4 + nb0), dim3(256), 0, s, a2, a4, a0, live2, live4);
    else hipLaunchKernelGGL(k_wgrad_gemm_trio<128>, dim3(nb2 + nb4 + nb0), dim3(256), 0, s, a2, a4, a0, live2, live4);
    GH_LAUNCH_CHECK("k_wgrad_gemm_trio");
    *rj2 = WgradReduceJob{partial2, dw2, sp2, hid, hid, hid, hid, 0};
    *rj4 = WgradReduceJob{partial4, dw4, sp4, m4, hid, m4_real, hid, 1};
    *rj0 = WgradReduceJob{partial0, dw0, sp0, hid, n0, hid, n0_real, 0};
    return GLOWHIP_OK;
}

int launch_wgrad_mfma(const float* A, long a_bs, const float* B, long b_bs, float* partial, float* dw, int N, int HW,
                      int Mpad, int Npad, int Mreal, int Nreal, int mode, hipStream_t s, float sh_scale, double* rowsum,
                      const WgradTaps* taps, WgradReduceJob* defer, int b_valid, int b_half, int tiled) {
    const bool ps = tiled & 4;          // bit 2: A holds g * sh_scale * 2^11 (the backward k_cnet's g_u2 / g_u0)
    const float a_pre = ps ? SH_LO_INV : sh_scale;
    GH_REQUIRE(!ps || !(taps && taps->operand == 0), "wgrad_mfma: a pre-scaled A is a plain one");
    GH_REQUIRE(!tiled || (sh_scale > 0.f && !(tiled & ~7) && !((tiled & 1) && taps && taps->operand == 0) &&
                          !((tiled & 2) && ((taps && taps->operand == 1) || b_valid > 0))),
               "wgrad_mfma: the pixel-tile-major layout is for plain operands of the split-half kernel");
    GH_REQUIRE(!rowsum || sh_scale > 0.f, "wgrad_mfma: row sums only on the split-half kernel");
    GH_REQUIRE(!b_half || (sh_scale > 0.f && b_valid <= 0 && !(taps && taps->operand == 1)), "wgrad_mfma: an fp16 operand B needs the split-half kernel and a plain B");
    GH_REQUIRE(b_valid <= 0 || (sh_scale > 0.f && taps && taps->operand == 0 && Npad == 64), "wgrad_mfma: b_valid only with gathered A and a 64-column B");
    GH_REQUIRE(!taps || (sh_scale > 0.f && taps->H * taps->W == HW && taps->W >= 4 && (taps->W & (taps->W - 1)) == 0 &&
                         (taps->operand == 0 || taps->operand == 1)),
               "wgrad_mfma: virtual operand needs the split-half kernel and a power-of-two width >= 4");
    GH_REQUIRE(wgrad_mfma_supported(HW, Mpad, Npad), "wgrad_mfma: unsupported shape");
    if (N == 0) return GLOWHIP_OK;
    const bool bn128 = Npad % 128 == 0;
    const int total = (int)((long)N * HW / 32);
    // f.2 behind the backward k_cnet at a level with enough pixels: 256-column tiles, one eight-wave workgroup per CU (k_wgrad_gemm_ps512)
    const bool wide = b_half && ps && Npad % 256 == 0 && total >= 512 && !g_wgrad_narrow;
    const int tiles = (Mpad / 128) * (wide ? Npad / 256 : (bn128 ? Npad / 128 : Npad / 64));
    int splits = std::max(1, std::min(total, ((wide ? 256 : 512) + tiles - 1) / tiles));   // 2 workgroups per CU (wide: 1, of twice the waves)
    const int per = (total + splits - 1) / splits;
    splits = (total + per - 1) / per;
    const int vC = taps ? taps->C : 0, vH = taps ? taps->H : 0, vW = taps ? taps->W : 1, vs = taps ? taps->sign : 0;
    const WgArgs wa{A, a_bs, B, b_bs, partial, HW, Mpad, Npad, total, per, sh_scale, a_pre, rowsum, vC, vH, vW, vs,
                    b_valid > 0 ? b_valid : Npad, tiled, tiles * splits};
#define GH_WG(bn, va, vb) hipLaunchKernelGGL((k_wgrad_gemm_sh<bn, va, vb>), dim3(tiles * splits), dim3(256), 0, s, wa)
#define GH_WGH(bn, va) hipLaunchKernelGGL((k_wgrad_gemm_sh<bn, va, false, false, true>), dim3(tiles * splits), dim3(256), 0, s, wa)
    if (b_half && ps && wide) {        // ... with 256-column tiles and eight waves, one workgroup per CU
        GH_REQUIRE((tiled & 3) == 3, "wgrad_mfma: f.2's kernel reads pixel-tile-major operands");
        constexpr int lds256 = wgrad_sh_lds_bytes<256, true>();
        (void)hipFuncSetAttribute((const void*)k_wgrad_gemm_ps512, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        hipLaunchKernelGGL(k_wgrad_gemm_ps512, dim3(tiles * splits), dim3(512), lds256, s, wa);
    } else
    if (b_half && ps) {                // f.2 behind the backward k_cnet: its own kernel
        GH_REQUIRE((tiled & 3) == 3, "wgrad_mfma: f.2's kernel reads pixel-tile-major operands");
        if (bn128) hipLaunchKernelGGL(k_wgrad_gemm_ps<128>, dim3(tiles * splits), dim3(256), 0, s, wa);
        else hipLaunchKernelGGL(k_wgrad_gemm_ps<64>, dim3(tiles * splits), dim3(256), 0, s, wa);
    } else
    if (b_half) {                      // B = h1 / h2 as fp16 from the tape
        const bool va = taps && taps->operand == 0;
        if (bn128) { if (va) GH_WGH(128, true); else GH_WGH(128, false); }
        else { if (va) GH_WGH(64, true); else GH_WGH(64, false); }
    } else
    if (sh_scale > 0.f && bn128) {     // f16 matrix pipe, split-half operands (sh_scale = power-of-two pre-scale of the gradient operand)
        if (taps && taps->operand == 0) GH_WG(128, true, false);
        else if (taps) GH_WG(128, false, true);
        else GH_WG(128, false, false);
    } else if (sh_scale > 0.f) {
        if (taps && taps->operand == 0 && b_valid > 0)
            hipLaunchKernelGGL((k_wgrad_gemm_sh<64, true, false, true>), dim3(tiles * splits), dim3(256), 0, s, wa);
        else if (taps && taps->operand == 0) GH_WG(64, true, false);
        else if (taps) GH_WG(64, false, true);
        else GH_WG(64, false, false);
    }
#undef GH_WG
#undef GH_WGH
    else if (bn128)
        hipLaunchKernelGGL(k_wgrad_gemm<128>, dim3(tiles * splits), dim3(256), 0, s, A, a_bs, B, b_bs, partial, HW, Mpad, Npad,
                           total, per);
    else
        hipLaunchKernelGGL(k_wgrad_gemm<64>, dim3(tiles * splits), dim3(256), 0, s, A, a_bs, B, b_bs, partial, HW, Mpad, Npad,
                           total, per);
    GH_LAUNCH_CHECK("k_wgrad_gemm");
    if (defer) {       // the caller reduces later (launch_wgrad_reduce_batched): `partial` must stay untouched until then
        *defer = WgradReduceJob{partial, dw, splits, Mpad, Npad, Mreal, Nreal, mode};
        return GLOWHIP_OK;
    }
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv((long)Mreal * ((Nreal + 3) / 4), 256)), dim3(256), 0, s, partial, dw, splits, Mpad, Npad,
                       Mreal, Nreal, mode);
    GH_LAUNCH_CHECK("k_wgrad_reduce");
    return GLOWHIP_OK;
}

}  // namespace glowhip
