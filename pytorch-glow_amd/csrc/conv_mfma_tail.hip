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
#include <type_traits>

#include "conv_mfma.h"

namespace glowhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int tail_chs(int TR, int W) {   // channel stride == 16 (mod 32) floats
    return (TR + 2) * (W + 8) + ((16 - ((TR + 2) * (W + 8)) % 32) + 32) % 32;
}

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

__device__ __forceinline__ float gauss_logp1_(float mean, float logs, float x) {
    const float d = x - mean;
    return -0.5f * (LOG_2PI_F + 2.0f * logs + (d * d) / expf(2.0f * logs));
}

// Epilogue shared by the tail kernels: (conv + bias) * exp(3 logs), then the coupling / prior arithmetic on the
// lane-local {even, odd} channel pairs.  Returns this lane's contribution to the per-sample log-det term.
template <int MT, int NTW>
__device__ __forceinline__ double tail_epilogue(const TailConvArgs& a, const f32x4 (&acc)[MT][NTW], long n, int p0,
                                                int HW, int wn, int mt0, int paired, int lane) {
    double ld = 0.0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mrow = (mt0 + m) * 16 + (lane >> 4) * 4;
        float hb[4], hs[4];
        int oc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            oc[r] = tail_row_channel(mrow + r, a.Cout, paired);
            hb[r] = oc[r] >= 0 ? a.bias[oc[r]] : 0.f;
            hs[r] = oc[r] >= 0 ? a.scale[oc[r]] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int p = p0 + (wn * NTW + nt) * 16 + (lane & 15);
            float hv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (acc[m][nt][r] + hb[r]) * hs[r];
            if (paired) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (oc[e] < 0) continue;
                    const int c = oc[e] >> 1;  // coupling channel
                    const float A_ = hv[e], B_ = hv[2 + e];
                    const long zi = n * a.z2_in_bs + (long)c * HW + p;
                    const long zo = n * a.z2_out_bs + (long)c * HW + p;
                    if (a.mode == TAIL_AFFINE_FWD) {
                        const float sc = sigmoidf_(B_ + 2.0f);
                        a.z2_out[zo] = (a.z2_in[zi] + A_) * sc;
                        ld += (double)logf(sc);
                    } else if (a.mode == TAIL_AFFINE_REV) {
                        const float sc = sigmoidf_(B_ + 2.0f);
                        a.z2_out[zo] = a.z2_in[zi] / sc - A_;
                        ld -= (double)logf(sc);
                    } else if (a.mode == TAIL_SPLIT_FWD) {
                        ld += (double)gauss_logp1_(A_, B_, a.z2_in[zi]);
                    } else {  // TAIL_SPLIT_REV
                        a.z2_out[zo] = A_ + expf(B_) * a.z2_in[zi];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (oc[r] < 0) continue;
                    const long zo = n * a.z2_out_bs + (long)oc[r] * HW + p;
                    if (a.mode == TAIL_PLAIN) a.z2_out[zo] = hv[r];
                    else {
                        const float z2 = a.z2_in[n * a.z2_in_bs + (long)oc[r] * HW + p];
                        a.z2_out[zo] = a.mode == TAIL_ADD_FWD ? z2 + hv[r] : z2 - hv[r];
                    }
                }
            }
        }
    }
    return ld;
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
        if (nc4 == TAIL_CK / 4) {
            // full chunk: 9 taps x (8/WK) channel groups, flattened and software-pipelined -- the LDS reads of
            // step s+1 are issued before the MFMAs of step s (pinned with sched_group_barrier)
            constexpr int CPW = (TAIL_CK / 4) / WK;   // channel groups per wave per tap
            constexpr int STEPS = 9 * CPW;
            float av[2][MT], bv[2][NTW];
            auto fetch = [&](int st, int slot) {
                const int tap = st / CPW, c4 = wk + (st % CPW) * WK;
                const int toff = (tap / 3 - 1) * g.RS + (tap % 3 - 1);
#pragma unroll
                for (int m = 0; m < MT; ++m) av[slot][m] = As[((c4 * 9 + tap) * MT + m) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bv[slot][nt] = Xs[boff[nt] + c4 * 4 * g.CHS + toff];
            };
            fetch(0, 0);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int cur = st & 1;
                if (st + 1 < STEPS) fetch(st + 1, cur ^ 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][m], bv[cur][nt], acc[m][nt], 0, 0, 0);
                if (st + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
            }
        } else {
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
        if (tid == 0) fix_atomic_add(a.acc + n, tot);
    }
}

// ================================================================================================
// k_conv_tail_dma: the same computation with both operand images streamed by the LDS-DMA path
// (global_load_lds_dwordx4) into a 3-stage ring of 16-channel chunks -- no staging VGPRs, no ds_write bursts,
// one raw barrier per chunk placed MID-chunk (the matrix pipe still has queued MFMAs), refill pieces issued one
// per k-step behind an executing MFMA.  The X window of a stage is written as a LINEAR LDS image (a DMA piece
// fills 1 KiB = 64 consecutive 16-byte slots); slots that are padding columns / rows outside the image read a
// 16-byte zero block in global memory instead of being skipped, so the zero padding is re-established by the
// DMA itself.  Needs Cin % 16 == 0 and compile-time geometry (WFIX).
// ================================================================================================
template <int MT, int NTW, int WN, int WK, int WFIX>
__global__ void __launch_bounds__(256) k_conv_tail_dma(TailConvArgs a, int paired) {
    static_assert(WN * WK == 4 && WFIX > 0, "4 waves per block, compile-time width");
    constexpr int TP = 16 * NTW * WN, CK = 16, NST = 3;
    constexpr int RS = WFIX + 8, TR = TP / WFIX, W4 = WFIX / 4, CHS = tail_chs(TR, WFIX);
    constexpr int X_FLOATS = (CK * CHS + 255) / 256 * 256, PX = X_FLOATS / 256;
    constexpr int A_FLOATS = (CK / 4) * 9 * MT * 64, PA = A_FLOATS / 256;
    constexpr int P = PX + PA, PPW = (P + 3) / 4;          // DMA pieces per stage / per wave (dummies pad to PPW)
    constexpr int STAGE = X_FLOATS + A_FLOATS;
    constexpr int CPW = (CK / 4) / WK, STEPS = 9 * CPW, MID = STEPS / 2;
    static_assert(PPW <= STEPS - MID, "not enough k-steps to spread the refill");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // NST*STAGE | 256 dummy | 8 floats (red)
    float* dummy = lds + NST * STAGE;
    double* red = reinterpret_cast<double*>(dummy + 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % WN, wk = wid / WN;
    const int HW = a.H * a.W;
    const long gp0 = (long)blockIdx.x * TP;
    const long n = gp0 / HW;
    const int p0 = (int)(gp0 - n * HW);
    const int y0 = p0 / a.W;
    const float* xin = a.x + n * a.x_bs;
    const int mt_total = gridDim.y * MT, mt0 = blockIdx.y * MT;
    const int nchunks = a.Cin / CK;

    // per-lane source of every DMA piece this wave issues (piece q = wid + 4*i of each stage)
    const float* src0[PPW];
    int stride[PPW];     // floats per chunk
    int ldso[PPW];       // LDS float offset inside the stage (wave-uniform), or -1 for the dummy area
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wid + 4 * i;
        if (q < PX) {
            const int slot = q * 64 + lane;
            const int c = slot / (CHS / 4), rem = slot - c * (CHS / 4);
            const int r = rem / (RS / 4), col4 = rem - r * (RS / 4);
            const int yy = y0 - 1 + r;
            const bool ok = c < CK && r < TR + 2 && col4 >= 1 && col4 <= W4 && yy >= 0 && yy < a.H;
            src0[i] = ok ? xin + (long)c * HW + (long)yy * a.W + (col4 - 1) * 4 : a.zeros;
            stride[i] = ok ? CK * HW : 0;
            ldso[i] = q * 256;
        } else if (q < P) {
            const int f = (q - PX) * 256 + lane * 4;
            const int row = f / (MT * 64), col = f - row * (MT * 64);
            src0[i] = a.wp + ((long)row * mt_total + mt0) * 64 + col;
            stride[i] = (CK / 4) * 9 * mt_total * 64;
            ldso[i] = X_FLOATS + (q - PX) * 256;
        } else {
            src0[i] = a.zeros;
            stride[i] = 0;
            ldso[i] = -1;
        }
    }
    auto issue_piece = [&](int ch, int i) {
        float* dst = ldso[i] >= 0 ? lds + (ch % NST) * STAGE + ldso[i] : dummy;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0[i] + (long)ch * stride[i]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    int boff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int q = (wn * NTW + nt) * 16 + (lane & 15);
        const int r = q / WFIX, x = q - r * WFIX;
        boff[nt] = (r + 1) * RS + x + 4 + (lane >> 4) * CHS;
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(0, i);
    if (nchunks > 1) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(1, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    float av[2][MT], bv[2][NTW];
    auto fetch = [&](const float* Xs, const float* As, int st, int slot) {
        const int tap = st / CPW, c4 = wk + (st % CPW) * WK;
        const int toff = (tap / 3 - 1) * RS + (tap % 3 - 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) av[slot][m] = As[((c4 * 9 + tap) * MT + m) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bv[slot][nt] = Xs[boff[nt] + c4 * 4 * CHS + toff];
    };
    fetch(lds, lds + X_FLOATS, 0, 0);
    // chunk body; PAR = parity of the fragment slot its first k-step reads (alternates from chunk to chunk when
    // STEPS is odd -- slot indices must stay compile-time constants, so the two parities are two instantiations)
    auto chunk_body = [&](auto par_c, int ch) {
        constexpr int PAR = decltype(par_c)::value;
        const float* Xs = lds + (ch % NST) * STAGE;
        const float* As = Xs + X_FLOATS;
        const float* Xn = lds + ((ch + 1) % NST) * STAGE;
        const float* An = Xn + X_FLOATS;
        const bool has_next = ch + 1 < nchunks;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = (st + PAR) & 1;
            if (st == MID && has_next) {
                // chunk ch+1 was requested half a chunk ago or earlier; nothing younger is in flight yet
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // ... and every wave is past chunk ch-1: its stage is free
                asm volatile("" ::: "memory");
            }
            if (st + 1 < STEPS) fetch(Xs, As, st + 1, cur ^ 1);
            else if (has_next) fetch(Xn, An, 0, cur ^ 1);
            const bool dma = st >= MID && st - MID < PPW && ch + 2 < nchunks;
            if (dma) issue_piece(ch + 2, st - MID);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][m], bv[cur][nt], acc[m][nt], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
            if (st >= MID && st - MID < PPW) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
        }
    };
    if constexpr (STEPS % 2 == 0) {
        for (int ch = 0; ch < nchunks; ++ch) chunk_body(std::integral_constant<int, 0>{}, ch);
    } else {
        for (int ch = 0; ch < nchunks; ch += 2) {
            chunk_body(std::integral_constant<int, 0>{}, ch);
            if (ch + 1 < nchunks) chunk_body(std::integral_constant<int, 1>{}, ch + 1);
        }
    }
    __syncthreads();

    if (WK > 1) {
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
            for (int k = 1; k < WK; ++k)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[m][nt][r] += part[(((k - 1) * WN + wn) * (MT * NTW * 4) + (m * NTW + nt) * 4 + r) * 64 + lane];
        }
    }
    double ld = 0.0;
    if (wk == 0) ld = tail_epilogue<MT, NTW>(a, acc, n, p0, HW, wn, mt0, paired, lane);
    if (a.acc && (a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV || a.mode == TAIL_SPLIT_FWD)) {
        const double tot = block_sum<256>(ld, red);
        if (tid == 0) fix_atomic_add(a.acc + n, tot);
    }
}

template <int MT, int NTW, int WN, int WK, int WFIX>
static int launch_tail_dma(const TailConvArgs& a, int paired, hipStream_t s, int msplit) {
    constexpr int TP = 16 * NTW * WN;
    constexpr int CHS = tail_chs(TP / WFIX, WFIX);
    constexpr int X_FLOATS = (16 * CHS + 255) / 256 * 256, A_FLOATS = 4 * 9 * MT * 64;
    size_t lds = ((size_t)3 * (X_FLOATS + A_FLOATS) + 256 + 8) * sizeof(float);
    const size_t red = (size_t)(WK - 1) * WN * MT * NTW * 4 * 64 * sizeof(float);
    if (red > lds) lds = red;
    const long total_px = (long)a.N * a.H * a.W;
    (void)hipFuncSetAttribute((const void*)k_conv_tail_dma<MT, NTW, WN, WK, WFIX>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_conv_tail_dma<MT, NTW, WN, WK, WFIX>), dim3((unsigned)(total_px / TP), msplit), dim3(256), lds, s,
                       a, paired);
    GH_LAUNCH_CHECK("k_conv_tail_dma");
    return GLOWHIP_OK;
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
        if (a.W == 128 && TP == 128) return launch_tail_dma<1, 2, 4, 1, 128>(a, paired, s, Y);
        if (a.W == 64 && TP == 128) return launch_tail_dma<1, 2, 4, 1, 64>(a, paired, s, Y);
        if (a.W == 64 && TP == 64) return launch_tail_dma<1, 1, 4, 1, 64>(a, paired, s, Y);
        if (a.W == 32 && TP == 128) return launch_tail_dma<1, 2, 4, 1, 32>(a, paired, s, Y);
        if (a.W == 32 && TP == 64) return launch_tail_dma<1, 1, 4, 1, 32>(a, paired, s, Y);
        if (a.W == 32 && TP == 32) return launch_tail_dma<1, 1, 2, 2, 32>(a, paired, s, Y);
#define GH_TAIL_DMA(w)                                                                    \
        if (a.W == w) {                                                                   \
            if (TP == 128) return launch_tail_dma<1, 2, 4, 1, w>(a, paired, s, Y);        \
            if (TP == 64) return launch_tail_dma<1, 1, 4, 1, w>(a, paired, s, Y);         \
            if (TP == 32) return launch_tail_dma<1, 1, 2, 2, w>(a, paired, s, Y);         \
            if (TP == 16) return launch_tail_dma<1, 1, 1, 4, w>(a, paired, s, Y);         \
        }
        GH_TAIL_DMA(16) GH_TAIL_DMA(8)
#undef GH_TAIL_DMA
    }
    // compile-time geometry for the three level shapes of the 64x64 / L=3 model (BASELINE configs B/C)
    if (MT == 1 && TP == 128 && a.W == 32) return launch_tail_cfg<1, 2, 4, 1, 32>(a, g, paired, s, Y);
    if (MT == 1 && TP == 64 && a.W == 16) return launch_tail_cfg<1, 1, 4, 1, 16>(a, g, paired, s, Y);
    if (MT == 1 && TP == 64 && a.W == 8) return launch_tail_cfg<1, 1, 4, 1, 8>(a, g, paired, s, Y);
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
}

}  // namespace glowhip
