// conv_mfma_first.hip -- f.0 of the coupling network: 3x3 convolution from a few input channels (C/2 = 6, 12,
// 24, ...) to `hidden` output channels with ActNorm + ReLU fused (network/module.py:252-259, 314-315).
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32 like k_conv_wide, but the pixel operand is NOT staged per K-tile:
// the workgroup's input window (all Cin channels x its rows + 1-pixel halo, zero padded) is loaded ONCE into
// LDS and stays there; a filter tap is an LDS address offset (no im2col expansion, no gather loads).  Only the
// weight operand streams: K is ordered (6-channel chunk, tap, channel) so that one LDS tile [54][128] holds all
// 9 taps of 6 input channels; tiles are double buffered and a workgroup walks MB output-channel tiles x Cin/6
// chunks with the pixel window stationary.  A k-pair of one MFMA = two adjacent channels of the same tap, so a
// fragment address is  lane_base(pixel, k&1) + scalar(chunk, tap, channel).
#include "conv_mfma.h"
#include "sh.h"

namespace glowhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FIRST_CK = 6;              // input channels per weight tile
constexpr int FIRST_BK = 9 * FIRST_CK;   // 54 k-rows per weight tile
constexpr int FIRST_BM = 128;            // output channels per tile

struct FirstGeom {
    int RS, CHS, TR, W4;   // LDS row stride / channel stride (floats), image rows per block, W/4
    int MB;                // output-channel tiles per workgroup
};

__host__ __device__ constexpr int first_chs(int TR, int W) {   // == 16 (mod 32): the two k-halves of a ds_read_b32
    return (TR + 2) * (W + 8) + ((16 - ((TR + 2) * (W + 8)) % 32) + 32) % 32;   // half-wave hit disjoint banks
}

bool conv_mfma_first_supported(int Cin, int H, int W, int Cout) {
    if (Cin % FIRST_CK != 0 || Cin > 192) return false;
    if (Cout % FIRST_BM != 0) return false;
    if (W != 8 && W != 16 && W != 32 && W != 64 && W != 128) return false;   // instantiated widths
    const int HW = H * W;
    const int BN = (HW % 128 == 0) ? 128 : 64;
    if (HW % BN != 0 || BN % W != 0) return false;
    // weight double buffer + stationary window must fit the 160 KiB LDS
    return ((size_t)2 * FIRST_BK * FIRST_BM + (size_t)Cin * first_chs(BN / W, W)) * sizeof(float) <= 150 * 1024;
}

// packed image: [9*Cin][Cout] weights pre-multiplied by exp(3 logs[o]), then Cout floats bias[o]*exp(3 logs[o])
size_t conv_mfma_first_packed_bytes(int Cin, int Cout) { return ((size_t)9 * Cin + 1) * Cout * sizeof(float); }

// BN: pixels per workgroup (128: waves 2x2 of 64x64; 64: waves 2x2 of 64x32).  WF: image width, compile time, so
// that every tap/channel LDS offset of the unrolled k-loop is an instruction immediate (with a runtime width the
// 27 offsets lived in VGPRs: 253 VGPRs + spills into AGPRs, one wave per SIMD).
template <int BN, int WF>
__global__ void __launch_bounds__(256, 2)     // (at 193 + 64 = 257 registers the BN = 128 instances ran one wave per SIMD)
k_conv_first(const float* __restrict__ X, long x_bs, const float* __restrict__ Wf, const float* __restrict__ bs,
             float* __restrict__ Y, int N, int Cin, int H, int W, int M, FirstGeom gr, int relu) {
    constexpr int BM = FIRST_BM, BK = FIRST_BK;
    constexpr int RS = WF + 8, TR = BN / WF, W4 = WF / 4, CHS = first_chs(BN / WF, WF);
    struct { int RS, CHS, TR, W4, MB; } g = {RS, CHS, TR, W4, gr.MB};
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_F4 = BK * BM / 4;                    // 1728 float4 per weight tile
    constexpr int A_IT = (A_F4 + 255) / 256;             // 7
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                                     // [2][BK][BM]
    float* Xs = lds + 2 * BK * BM;                       // [Cin][CHS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int kl = lane >> 5, ml = lane & 31;
    const int HW = H * W;
    const int mgroups = (M / BM) / g.MB;
    const int mg = blockIdx.x % mgroups;                 // consecutive blocks share the pixel window (L2 locality)
    const long gp0 = (long)(blockIdx.x / mgroups) * BN;
    const long n = gp0 / HW;
    const int p0 = (int)(gp0 - n * HW);
    const int y0 = p0 / W;
    const float* xin = X + n * x_bs;
    const int nch = Cin / FIRST_CK;
    const int steps = g.MB * nch;

    // ---- weight tile prefetch registers + first tile
    f32x4 rA[A_IT];
    auto load_A = [&](int st) {
        const int mt = mg * g.MB + st / nch, ch = st % nch;
        const float* src = Wf + (long)ch * BK * M + (long)mt * BM;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int e = it * 256 + tid;
            const int row = e / (BM / 4), c4 = e % (BM / 4);
            rA[it] = (e < A_F4) ? *reinterpret_cast<const f32x4*>(src + (long)row * M + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_A = [&](int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int e = it * 256 + tid;
            if (e < A_F4) *reinterpret_cast<f32x4*>(As + buf * BK * BM + e * 4) = rA[it];
        }
    };
    load_A(0);

    // ---- stationary pixel window: zero fill, then rows y0-1 .. y0+TR of every channel (aligned float4 rows)
    for (int e = tid; e < Cin * g.CHS; e += 256) Xs[e] = 0.f;
    __syncthreads();
    {
        const int rows = g.TR + 2, per_ch = rows * g.W4, count = Cin * per_ch;
        for (int e = tid; e < count; e += 256) {
            const int c = e / per_ch, rem = e - c * per_ch;
            const int r = rem / g.W4, x4 = rem - r * g.W4;
            const int yy = y0 - 1 + r;
            if (yy >= 0 && yy < H)
                *reinterpret_cast<f32x4*>(Xs + c * g.CHS + r * g.RS + 4 + x4 * 4) =
                    *reinterpret_cast<const f32x4*>(xin + (long)c * HW + (long)yy * W + x4 * 4);
        }
    }
    store_A(0);
    __syncthreads();

    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int q = wc * WN + j * 32 + ml;
        const int r = q / W, x = q - r * W;
        boff[j] = (r + 1) * g.RS + x + 4 + kl * g.CHS;
    }

    f32x16 acc[TM][TN];
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        const int mt = mg * g.MB + st / nch, ch = st % nch;
        if (ch == 0) {
            // ActNorm folded into the GEMM: the packed weights carry exp(3 logs[o]) and the accumulators start at
            // bias[o]*exp(3 logs[o]), so the epilogue is ReLU + store (no per-row parameter loads kept live)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float b0 = bs[mt * BM + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl];
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] = b0;
                }
        }
        if (st + 1 < steps) load_A(st + 1);
        const float* Ab = As + buf * BK * BM;
        const float* Xc = Xs + ch * FIRST_CK * CHS;
        // 27 k-steps: tap-major, channel pairs inner; fragments software-pipelined one step ahead
        float a[2][TM], b[2][TN];
        auto fetch = [&](int ks, int slot) {
            const int tap = ks / 3, cl = (ks % 3) * 2;
            const int koff = cl * CHS + (tap / 3 - 1) * RS + (tap % 3 - 1);   // compile-time after unrolling
#pragma unroll
            for (int i = 0; i < TM; ++i) a[slot][i] = Ab[(tap * FIRST_CK + cl + kl) * BM + wr * WM + i * 32 + ml];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[slot][j] = Xc[boff[j] + koff];
        };
        fetch(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < BK / 2) fetch(ks + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            if (ks + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        if (st + 1 < steps) store_A(buf ^ 1);
        if (ch == nch - 1) {
            // epilogue of this output-channel tile: ActNorm + ReLU, C[row = o][col = pixel]
            {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int p = p0 + wc * WN + j * 32 + ml;
                float* yn = Y + n * (long)M * HW + p;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = mt * BM + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                        yn[(long)o * HW] = relu ? relu_(acc[i][j][r]) : acc[i][j][r];
                    }
                }
            }
            }
        }
        __syncthreads();
    }
}

int launch_conv_mfma_first(const float* x, long x_bs, const float* wf_, const float* bias_scaled, float* y, int N,
                           int Cin, int H, int W, int Cout, hipStream_t s, int relu) {
    GH_REQUIRE(conv_mfma_first_supported(Cin, H, W, Cout), "conv_mfma_first: unsupported shape");
    if (N == 0) return GLOWHIP_OK;
    const int HW = H * W;
    const long total_px = (long)N * HW;
    const int mtiles = Cout / FIRST_BM;
    // pixel tile: 128 when that still gives >= 512 workgroups (with MB=1), else 64; then fold output-channel
    // tiles into a workgroup (MB) while >= 512 workgroups remain, so the pixel window is reused
    int BN = (HW % 128 == 0 && 128 % W == 0) ? 128 : 64;
    if (BN == 128 && HW % 64 == 0 && 64 % W == 0 && (total_px / 128) * mtiles < 512 &&
        ((size_t)2 * FIRST_BK * FIRST_BM + (size_t)Cin * first_chs(64 / W, W)) * sizeof(float) <= 150 * 1024)
        BN = 64;
    int MB = 1;
    while (MB * 2 <= mtiles && mtiles % (MB * 2) == 0 && (total_px / BN) * (mtiles / (MB * 2)) >= 512) MB *= 2;
    FirstGeom g;
    g.RS = W + 8; g.TR = BN / W; g.W4 = W / 4; g.CHS = first_chs(g.TR, W); g.MB = MB;
    const size_t lds = ((size_t)2 * FIRST_BK * FIRST_BM + (size_t)Cin * g.CHS) * sizeof(float);
    const unsigned grid = (unsigned)((total_px / BN) * (mtiles / MB));
#define GH_FIRST_CASE(bn, wf)                                                                                        \
    if (BN == bn && W == wf) {                                                                                       \
        (void)hipFuncSetAttribute((const void*)k_conv_first<bn, wf>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)lds);                                                                         \
        hipLaunchKernelGGL((k_conv_first<bn, wf>), dim3(grid), dim3(256), lds, s, x, x_bs, wf_, bias_scaled, y, N,   \
                           Cin, H, W, Cout, g, relu);                                                                \
        GH_LAUNCH_CHECK("k_conv_first");                                                                             \
        return GLOWHIP_OK;                                                                                           \
    }
    GH_FIRST_CASE(128, 32) GH_FIRST_CASE(64, 32) GH_FIRST_CASE(128, 16) GH_FIRST_CASE(64, 16) GH_FIRST_CASE(64, 8)
    GH_FIRST_CASE(128, 8) GH_FIRST_CASE(128, 64) GH_FIRST_CASE(64, 64) GH_FIRST_CASE(128, 128)
#undef GH_FIRST_CASE
    set_error("conv_mfma_first: no kernel for BN=%d W=%d", BN, W);
    return GLOWHIP_EINVAL;
}

}  // namespace glowhip
