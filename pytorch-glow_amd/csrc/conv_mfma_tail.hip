// conv_mfma_tail.hip -- 3x3 convolution to a few output channels with the coupling / prior arithmetic
// fused into the epilogue (see the header comment of conv_mfma.hip for the design).
//
// MFMA: v_mfma_f32_16x16x4_f32, A = weights [16 out-rows][4 in-channels], B = activations
// [4 in-channels][16 pixels]; C[row = out-row][col = pixel]: lane l holds rows (l>>4)*4 + {0..3} of pixel
// l&15.  Out-rows are PERMUTED at pack time so that a lane's four rows are {shift_c0, shift_c1, scale_c0,
// scale_c1} (or {mean, mean, logs, logs}) of two coupling channels: the whole affine update is lane-local.
//
// LDS image of the input: per 32-channel chunk a zero-padded halo tile [32][TR+2][W+8] (data at columns
// 4..W+3 => 16-byte aligned rows, pad columns 3 and W+4 stay zero), channel stride == 16 (mod 32) floats so
// the two 16-lane pixel runs of a ds_read_b32 half-wave fall on disjoint banks.  A tap is an LDS address
// offset; nothing is expanded (no im2col), each input element is fetched from HBM once per block.
#include "conv_mfma_tail_dma.h"
#include "sh.h"

namespace glowhip {

// Pixel-tile choice.  A block covers TP = 16*NTW*WN consecutive pixels = whole image rows of ONE image; its 4
// waves are arranged WN (pixel tiles) x WK (K-split: wave wk takes channel groups c4 = wk, wk+WK, ...), partial
// sums of the WK waves are reduced through LDS before the epilogue.  Smaller tiles + K-split keep >= 2 blocks per
// CU on the deep levels (8x8 images: 4096 pixels per step) where a 128-pixel tile would occupy 1/4 of the chip.
static int g_force_tp = 0;      // testing hooks (glowhip_debug_force_tail_tile): pixels per block, 0 = automatic
static int g_force_msplit = -1; // -1 automatic, 0 never, 1 always split the out-channel tiles over blockIdx.y
static bool g_disable_tail_dma = false;
// register-staged kernels: the halo tile must fit the 6-float4 staging registers
static bool tp_ok_reg(int tp, int H, int W) {
    if (tp % W != 0 || (H * W) % tp != 0) return false;
    return TAIL_CK * (tp / W + 2) * (W / 4) <= 6 * 256;
}
// LDS-DMA kernels: instantiated widths, 16-channel chunks, 3 stages within the 160 KiB LDS
static bool tp_ok_dma(int tp, int H, int W, int Cin) {
    if (g_disable_tail_dma || Cin % 16 != 0 || Cin < 32) return false;
    if (W != 8 && W != 16 && W != 32 && W != 64 && W != 128) return false;
    if (tp % W != 0 || (H * W) % tp != 0) return false;
    const int xf = (16 * tail_chs(tp / W, W) + 255) / 256 * 256;
    return (size_t)(3 * (xf + 4 * 9 * 64) + 264) * sizeof(float) <= 150 * 1024;
}
static bool tp_ok(int tp, int H, int W, int Cin) { return tp_ok_reg(tp, H, W) || tp_ok_dma(tp, H, W, Cin); }
// Choose pixels-per-block and whether to split the out-channel tiles over blockIdx.y with a two-term cost model
// per CU: matrix-pipe cycles vs. weight-streaming cycles.  Every block streams the weights of its out-channel
// tiles once (Cin*9*16*4 B per tile; a CU sustains ~10 B/clk of such L2->LDS refills), so many small pixel tiles
// are stream-bound (the deep levels), while few large ones leave CUs idle.
struct TailChoice { int tp, msplit; };
static TailChoice tail_choose(int H, int W, long total_px, int Cin, int mt_total) {
    const int cand[4] = {128, 64, 32, 16};
    TailChoice best{0, 0};
    double best_cost = 1e30;
    for (int ms = 0; ms <= 1; ++ms) {
        if (ms == 1 && mt_total == 1) continue;
        if (g_force_msplit >= 0 && ms != g_force_msplit && mt_total > 1) continue;
        for (int i = 0; i < 4; ++i) {
            const int tp = cand[i];
            // a DMA-only tile needs the out-channel split (MT == 1 kernels); more than 3 tiles per block do not exist
            const bool reg = tp_ok_reg(tp, H, W), dmaok = tp_ok_dma(tp, H, W, Cin);
            if (!(reg || dmaok)) continue;
            const int mt_blk = ms ? 1 : mt_total;
            if (mt_blk > 3 || (!reg && mt_blk != 1)) continue;
            if (g_force_tp && tp != g_force_tp && tp_ok(g_force_tp, H, W, Cin)) continue;
            const int wk = tp >= 64 ? 1 : (tp == 32 ? 2 : 4), ntw = tp == 128 ? 2 : 1;
            const int mt_b = ms ? 1 : mt_total;
            const double blocks = (double)(total_px / tp) * (ms ? mt_total : 1);
            const double per_cu = blocks <= 256 ? 1.0 : blocks / 256.0;
            const double mfma = per_cu * ntw * mt_b * ((Cin + 3) / 4 * 9.0 / wk) * 32.0;
            const double stream = per_cu * (double)Cin * 9 * mt_b * 64 / 10.0;
            double cost = mfma > stream ? mfma : stream;
            if (blocks < 512) cost *= 1.15;            // a lone block per CU cannot hide its own refill latency
            if (blocks < 256) cost *= 2.0;             // ... and idle CUs on top (measured: 8x8 level, 192 blocks)
            cost += 1e-3 * stream;                       // tie-break: less streaming
            if (cost < best_cost) { best_cost = cost; best = TailChoice{tp, ms}; }
        }
    }
    return best;
}

static bool tail_paired(int mode) { return mode != TAIL_PLAIN && mode != TAIL_ADD_FWD && mode != TAIL_ADD_REV; }

bool conv_mfma_tail_supported(int Cin, int H, int W, int Cout) {
    if (W % 4 != 0 || W < 8) return false;
    if (Cout > 1024 || Cout < 1 || Cin < 1) return false;
    return tp_ok(128, H, W, Cin) || tp_ok(64, H, W, Cin) || tp_ok(32, H, W, Cin) || tp_ok(16, H, W, Cin);
}

size_t conv_mfma_tail_packed_bytes(int Cin, int Cout) {
    // sized for the larger of the two row layouts (paired rows >= unpaired rows for even Cout)
    const int mt = tail_mt(Cout, 1) > tail_mt(Cout, 0) ? tail_mt(Cout, 1) : tail_mt(Cout, 0);
    return (size_t)tail_chunks(Cin) * (TAIL_CK / 4) * 9 * mt * 64 * sizeof(float);
}

// wp[(((chunk*8 + c4)*9 + tap)*MT + mt)*64 + kq*16 + i] = w[o(mt*16+i)][chunk*32 + c4*4 + kq][tap]
__global__ void __launch_bounds__(256) k_pack_tail(const float* __restrict__ w, int Cin, int Cout, int paired, int MT,
                                                   long total, float* __restrict__ wp) {
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int i = (int)(e & 15), kq = (int)((e >> 4) & 3);
    long t = e >> 6;
    const int mt = (int)(t % MT); t /= MT;
    const int tap = (int)(t % 9); t /= 9;
    const int c4 = (int)(t % (TAIL_CK / 4));
    const int chunk = (int)(t / (TAIL_CK / 4));
    const int ci = chunk * TAIL_CK + c4 * 4 + kq;
    const int o = tail_row_channel(mt * 16 + i, Cout, paired);
    wp[e] = (o >= 0 && ci < Cin) ? w[((long)o * Cin + ci) * 9 + tap] : 0.f;
}

int conv_mfma_tail_pack(const float* w, int Cin, int Cout, int paired, float* wp, hipStream_t s) {
    const int MT = tail_mt(Cout, paired);
    const long total = (long)tail_chunks(Cin) * (TAIL_CK / 4) * 9 * MT * 64;
    hipLaunchKernelGGL(k_pack_tail, dim3(cdiv(total, 256)), dim3(256), 0, s, w, Cin, Cout, paired, MT, total, wp);
    GH_LAUNCH_CHECK("k_pack_tail");
    return GLOWHIP_OK;
}

struct TailGeom {
    int RS;    // LDS row stride (floats) = W + 8
    int CHS;   // LDS channel stride (floats), == 16 mod 32
    int TR;    // image rows per block tile
    int W4;    // W / 4
};

// WFIX > 0: image width known at compile time (the BASELINE level geometries) => every LDS offset of the
// unrolled tap/channel pipeline is an instruction immediate instead of a live VGPR; WFIX == 0: runtime geometry.
template <int MT, int NTW, int WN, int WK, int WFIX>
__global__ void __launch_bounds__(256) k_conv_tail(TailConvArgs a, TailGeom gr, int paired) {
    static_assert(WN * WK == 4, "4 waves per block");
    constexpr int TP = 16 * NTW * WN;                  // pixels per block
    TailGeom g = gr;
    if (WFIX > 0) {
        g.RS = WFIX + 8;
        g.TR = TP / (WFIX > 0 ? WFIX : 1);
        g.W4 = WFIX / 4;
        g.CHS = tail_chs(TP / (WFIX > 0 ? WFIX : 1), WFIX);
    }
    constexpr int A_FLOATS = (TAIL_CK / 4) * 9 * MT * 64;
    constexpr int A_F4 = A_FLOATS / 4;
    constexpr int A_IT = (A_F4 + 255) / 256;
    constexpr int X_IT_MAX = 6;                        // ceil(32 * (TR+2) * W4 / 256) <= 6 for every supported geometry
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                                   // [TAIL_CK][CHS]
    float* As = lds + TAIL_CK * g.CHS;                 // [8][9][MT][64]
    __shared__ double red[4];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid % WN, wk = wid / WN;            // pixel-tile column / K-split slice of this wave
    const int HW = a.H * a.W;
    const long gp0 = (long)blockIdx.x * TP;
    const int mt_total = gridDim.y * MT, mt0 = blockIdx.y * MT;
    const long n = gp0 / HW;
    const int p0 = (int)(gp0 - n * HW);
    const int y0 = p0 / a.W;
    const float* xin = a.x + n * a.x_bs;

    // zero the X image once: pad columns / out-of-image rows are never written afterwards
    for (int e = tid; e < TAIL_CK * g.CHS; e += 256) Xs[e] = 0.f;

    const int rows = g.TR + 2;
    const int x_per_ch = rows * g.W4;
    const int x_count = TAIL_CK * x_per_ch;
    const int nchunks = (a.Cin + TAIL_CK - 1) / TAIL_CK;

    f32x4 rx[X_IT_MAX];  // native vectors (HIP's float4 struct arrays are not reliably kept in registers)
    f32x4 rA[A_IT];
    int x_dst[X_IT_MAX];
    long x_src[X_IT_MAX];
    bool x_rowok[X_IT_MAX];
    int x_c[X_IT_MAX];
#pragma unroll
    for (int it = 0; it < X_IT_MAX; ++it) {
        const int e = it * 256 + tid;
        const int c = e / x_per_ch, rem = e - c * x_per_ch;
        const int r = rem / g.W4, x4 = rem - r * g.W4;
        const int yy = y0 - 1 + r;
        x_c[it] = (e < x_count) ? c : TAIL_CK;  // TAIL_CK => never loaded
        x_rowok[it] = (e < x_count) && yy >= 0 && yy < a.H;
        x_src[it] = (long)c * HW + (long)yy * a.W + x4 * 4;
        x_dst[it] = c * g.CHS + r * g.RS + 4 + x4 * 4;
    }
    auto load_chunk = [&](int ch) {
        const int cbase = ch * TAIL_CK;
#pragma unroll
        for (int it = 0; it < X_IT_MAX; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (x_rowok[it] && cbase + x_c[it] < a.Cin)
                v = *reinterpret_cast<const f32x4*>(xin + (long)cbase * HW + x_src[it]);
            rx[it] = v;
        }
        // packed weights are [c4][tap][mt_total][64]; this block takes MT consecutive m-tiles from mt0
        // (M-split over blockIdx.y: a block streams only its own share of the weights)
        const float* asrc = a.wp + (long)ch * (TAIL_CK / 4) * 9 * mt_total * 64;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int e = it * 256 + tid;                 // float4 index into the block's [8*9][MT*64] image
            const int row = e / (MT * 16), c = e - row * (MT * 16);
            rA[it] = (e < A_F4) ? *reinterpret_cast<const f32x4*>(asrc + ((long)row * mt_total + mt0) * 64 + c * 4)
                                : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_chunk = [&]() {
        // rows outside the image / channels beyond Cin hold zeros in rx: written too, so the halo stays clean
#pragma unroll
        for (int it = 0; it < X_IT_MAX; ++it)
            if (x_c[it] < TAIL_CK) *reinterpret_cast<f32x4*>(Xs + x_dst[it]) = rx[it];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int e = it * 256 + tid;
            if (e < A_F4) reinterpret_cast<f32x4*>(As)[e] = rA[it];
        }
    };

    // per-lane LDS offsets of this wave's pixels
    int boff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int q = (wn * NTW + nt) * 16 + (lane & 15);
        const int r = q / a.W, x = q - r * a.W;
        boff[nt] = (r + 1) * g.RS + x + 4 + (lane >> 4) * g.CHS;
    }

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    __syncthreads();  // zero-fill done
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) load_chunk(ch + 1);
        const int left = a.Cin - ch * TAIL_CK;
        const int nc4 = left >= TAIL_CK ? TAIL_CK / 4 : (left + 3) / 4;
        {   // plain tap x channel-group loops: this kernel now only serves small / odd shapes (compile time matters
            // more than the last 20 % here; the hot shapes run k_conv_tail_dma)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3 - 1) * g.RS + (tap % 3 - 1);
                for (int c4 = wk; c4 < nc4; c4 += WK) {
                    float av[MT], bv[NTW];
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[m] = As[((c4 * 9 + tap) * MT + m) * 64 + lane];
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) bv[nt] = Xs[boff[nt] + c4 * 4 * g.CHS + toff];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[nt], acc[m][nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (ch + 1 < nchunks) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- K-split reduction (WK > 1)
    if (WK > 1) {
        // the chunk loop ended with a barrier: the operand images are dead, reuse LDS as [WK-1][WN][MT*NTW*4][64]
        float* part = lds;
        if (wk > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        part[(((wk - 1) * WN + wn) * (MT * NTW * 4) + (m * NTW + nt) * 4 + r) * 64 + lane] = acc[m][nt][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int k = 1; k < WK; ++k)   // fixed order => deterministic
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[m][nt][r] += part[(((k - 1) * WN + wn) * (MT * NTW * 4) + (m * NTW + nt) * 4 + r) * 64 + lane];
        }
    }

    // ---------------------------------------------------------------- epilogue (waves with wk == 0)
    double ld = 0.0;
    if (wk == 0) ld = tail_epilogue<MT, NTW>(a, acc, n, p0, HW, wn, mt0, paired, lane);
    if (a.acc && (a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV || a.mode == TAIL_SPLIT_FWD)) {
        const double tot = block_sum<256>(ld, red);
        if (tid == 0) fix_atomic_add(a.acc, n, a.N, tot);
    }
}

template <int MT, int NTW, int WN, int WK, int WFIX = 0>
static int launch_tail_cfg(const TailConvArgs& a, const TailGeom& g, int paired, hipStream_t s, int msplit = 1) {
    constexpr int TP = 16 * NTW * WN;
    const long total_px = (long)a.N * a.H * a.W;
    size_t lds = ((size_t)TAIL_CK * g.CHS + (size_t)(TAIL_CK / 4) * 9 * MT * 64) * sizeof(float);
    const size_t red = (size_t)(WK - 1) * WN * MT * NTW * 4 * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (lds > 32 * 1024)
        (void)hipFuncSetAttribute((const void*)k_conv_tail<MT, NTW, WN, WK, WFIX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL((k_conv_tail<MT, NTW, WN, WK, WFIX>), dim3((unsigned)(total_px / TP), msplit), dim3(256), lds, s, a,
                       g, paired);
    GH_LAUNCH_CHECK("k_conv_tail");
    return GLOWHIP_OK;
}

int launch_conv_mfma_tail(const TailConvArgs& a, hipStream_t s) {
    GH_REQUIRE(conv_mfma_tail_supported(a.Cin, a.H, a.W, a.Cout), "conv_mfma_tail: unsupported shape");
    if (a.N == 0) return GLOWHIP_OK;
    const int paired = tail_paired(a.mode);
    GH_REQUIRE(!paired || a.Cout % 2 == 0, "conv_mfma_tail: paired mode needs an even Cout");
    const int MTT = tail_mt(a.Cout, paired);   // out-channel tiles in total
    const TailChoice tc = tail_choose(a.H, a.W, (long)a.N * a.H * a.W, a.Cin, MTT);
    const int TP = tc.tp;
    GH_REQUIRE(TP > 0, "conv_mfma_tail: no pixel tile for %dx%d", a.H, a.W);
    GH_REQUIRE(tp_ok_reg(TP, a.H, a.W) || a.zeros, "conv_mfma_tail: this shape needs the LDS-DMA kernel (zero block missing)");
    TailGeom g;
    g.RS = a.W + 8;
    g.TR = TP / a.W;
    g.W4 = a.W / 4;
    g.CHS = tail_chs(g.TR, a.W);
    const int MT = tc.msplit ? 1 : MTT;          // tiles per block
    const int Y = tc.msplit ? MTT : 1;           // blockIdx.y extent
    // LDS-DMA pipeline for the level shapes of the 64x64 / L=3 model (needs a global zero block and Cin % 16 == 0)
    if (MT == 1 && a.zeros && tp_ok_dma(TP, a.H, a.W, a.Cin)) {
        return a.W >= 32 ? launch_tail_dma_wide(a, paired, s, TP, Y) : launch_tail_dma_narrow(a, paired, s, TP, Y);
    }
#define GH_TAIL_TP(mt)                                                               \
    if (MT == mt) {                                                                  \
        if (TP == 128) return launch_tail_cfg<mt, 2, 4, 1>(a, g, paired, s, Y);      \
        if (TP == 64) return launch_tail_cfg<mt, 1, 4, 1>(a, g, paired, s, Y);       \
        if (TP == 32) return launch_tail_cfg<mt, 1, 2, 2>(a, g, paired, s, Y);       \
        if (TP == 16) return launch_tail_cfg<mt, 1, 1, 4>(a, g, paired, s, Y);       \
    }
    GH_TAIL_TP(1) GH_TAIL_TP(2) GH_TAIL_TP(3)
#undef GH_TAIL_TP
    set_error("conv_mfma_tail: no kernel for MT=%d TP=%d", MT, TP);
    return GLOWHIP_EINVAL;
}

// testing hook: tp = pixels per block (0 automatic); bit 8 set => force msplit on (0x100 | tp), bit 9 => force off
void conv_mfma_tail_force_tile(int v) {
    g_force_tp = v & 0xff;
    g_force_msplit = (v & 0x100) ? 1 : ((v & 0x200) ? 0 : -1);
    g_disable_tail_dma = (v & 0x400) != 0;   // | 0x400: register-staged kernels only
    // 0x800: the split-half path off (every coupling network on the exact-fp32 kernels, training included); 0x8000: no mixer of the
    // next step inside the finishing kernel and no squeeze folded into a mixer (the fused and the separate forms must agree bit for bit)
    // 0x100000: FUSED finishing on (k_cnet1w finishing the step itself instead of a k_cfinish launch; off by default: measured slower)
    // 0x200000: log|det W| of the small invconv matrices on the workgroup-wide LU instead of one wave per matrix (lu.hip; same bits)
    plan_disable_sh(((v & 0x800) ? 1 : 0) | ((v & 0x8000) ? 16 : 0) | ((v & 0x100000) ? 32 : 0) | ((v & 0x200000) ? 64 : 0));
    cnet_force((v >> 22) & 7, ((v >> 25) & 15) | ((v & 0x10000) ? 16 : 0) | ((v & 0x20000) ? 32 : 0) | ((v & 0x40000) ? 64 : 0));   // 0x40000: no backward instance of k_cnet1w;   // 0x10000: no k_cnet1w (one wave per SIMD); 0x20000: its row-split instance where it applies (off by default: measured slower);   // bits 22..24: row splits; bit 25: 128-pixel tiles only, bit 26: 64-pixel tiles, bit 27: finishing chained into the next k_cnet, bit 28: finishing kernel without the XCD-affine chunk order
    plan_train_disable_sh((v & 0x800) ? 1 : 0);
    wgrad_force_narrow((v & 0x80000) ? 1 : 0);           // 0x80000: f.2's weight-gradient GEMM on 128-column tiles everywhere
    plan_pack_one_stream((v & 0x20000000) ? 1 : 0);      // bit 29: glowhip_plan_pack entirely on the caller's stream (A/B)
    plan_train_disable_cnet((v & 0x40000000) ? 1 : 0);   // bit 30: training forward on the per-layer kernels (no taping k_cnet)
    plan_train_disable_cnet_bwd(((unsigned)v & 0x80000000u) ? 1 : 0);   // bit 31: input-gradient chain on the per-layer kernels
}

}  // namespace glowhip
