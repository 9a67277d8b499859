// tail_sh.hip -- f.4 (3x3 convolution hidden -> Cout, `Conv2dZeros`, network/module.py:263-297) with the coupling arithmetic
// and the per-sample log-det fused (network/model.py:105-113, 139-150), on split-half operands (sh.h).
//
// The filter taps are moved to the OUTPUT side: the kernel computes the plain GEMM
//     T[tap*Cout + co][px] = sum_ci W[co][ci][tap] * h2[ci][px]           (M = 9*Cout rows, K = hidden, no shifts)
// over the workgroup's pixel window (its R image rows + one halo row above and below), and the convolution is the
// shifted sum  out[co][y][x] = sum_tap T[tap*Cout + co][y+dy][x+dx]  taken from LDS in the epilogue.  Why: with the f16
// matrix pipe 16x faster than the fp32 one, LDS bandwidth is the scarce resource; a tap loop would read every activation
// fragment nine times (once per shift), this form reads it once and gives the GEMM a tall M (108..432 rows instead of
// 12..48).  Cost: the halo rows are computed redundantly ((R+2)/R more MFMA work, none of it HBM traffic).
//
// Main loop = gemm_sh.hip's: 32x32x16 f16 MFMAs (main + cross accumulators), operands streamed by LDS-DMA into a
// 4-stage ring of 16-deep k-tiles, counted vmcnt + one barrier per stage.  Window slots outside the image read a 16-byte
// zero block, so the zero padding comes out of the DMA itself.  A wave owns MW x NW 32x32 tiles.
#include "sh.h"
#include <algorithm>

#include "conv_mfma.h"

GH_STAMPS_DEFINE(tail)

namespace glowhip {


// T row stride (floats) for M9 = 9*Cg GEMM rows: multiple of 4 (16-byte stores), an ODD multiple so that the pixels of
// a wave land on different banks
__host__ __device__ inline int tail_sh_trow(int M9) {
    int r = (M9 + 3) / 4;
    if ((r & 1) == 0) ++r;
    return r * 4;
}

constexpr int TSH_ST = 4;       // ring stages (16 input channels each)
constexpr int TSH_MAXE = 4;     // epilogue items ((channel [pair], pixel)) per thread, at most

// NWV waves per workgroup (4 or 8).  Eight waves halve the LDS-DMA pieces each wave has to issue per stage (an issue costs
// 100-200 cycles and is serial per wave: with four waves the stage period was ~1100 cycles against 576 cycles of MFMA).
template <int MW, int NW, int PPW, int NWV>
__global__ void __launch_bounds__(64 * NWV) k_tail_sh(TailShArgs a, int WGM, int R, int Mpad, int Nwpad, int wshift, int groups) {
    constexpr int NTHR = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_t[];
    const int W = a.W, H = a.H, HW = H * W, K = a.Cin;
    const int Cout = a.Cout / groups;                  // output channels of this workgroup's group (blockIdx.y)
    const int c0 = blockIdx.y * Cout;
    const int Nw = (R + 2) * W;
    const int a_halfs = 4 * Mpad * 8;                  // [plane 2][chunk 2][Mpad][8]
    const int stage_halfs = a_halfs + 4 * Nwpad * 8;
    const int M9 = 9 * Cout;
    const int Mrow = tail_sh_trow(M9);                 // T row stride (floats): multiple of 4, odd multiple => spread banks
    const size_t ring_bytes = (size_t)TSH_ST * stage_halfs * sizeof(_Float16);
    const size_t t_bytes = (size_t)Nw * Mrow * sizeof(float);
    char* tailp = (char*)smem_t + (ring_bytes > t_bytes ? ring_bytes : t_bytes);
    _Float16* dummy = (_Float16*)tailp;                // 1 KiB landing area of padding DMA pieces
    double* red = (double*)(tailp + 1024);

    GH_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WGN = NWV / WGM;
    const int wm = wid / WGN, wn = wid - wm * WGN;
    const int kl = lane >> 5, ml = lane & 31;
    const int bpi = H / R;                             // workgroups per image
    // XCD-aware order: vertically adjacent row blocks share their halo rows, so give each XCD (private L2) a contiguous run
    // of row blocks instead of every 8th one
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const long n = lb / bpi;
    const int y0 = (int)(lb - n * bpi) * R;
    const long w_plane = (long)K * Mpad;
    const int pa = Mpad >> 6, pb = Nwpad >> 6, PT = 4 * (pa + pb);

    // ---- epilogue work items of this thread and their z2 inputs, requested NOW (their latency hides behind the GEMM)
    const int tile_px = R * W;
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    const int nch = paired ? Cout / 2 : Cout;
    const int nitems = nch * tile_px;
    float zin[TSH_MAXE], z1in[TSH_MAXE];
#pragma unroll
    for (int i = 0; i < TSH_MAXE; ++i) {
        const int e = tid + i * NTHR;
        zin[i] = 0.f; z1in[i] = 0.f;
        if (e < nitems) {
            const int c = e / tile_px, q = e - c * tile_px;
            const int cg = (paired ? c0 / 2 : c0) + c;
            zin[i] = a.z2_in[n * a.z2_in_bs + (long)cg * HW + (long)y0 * W + q];
            if (a.mix_C) z1in[i] = a.mix_z1[n * a.mix_z1_bs + (long)c * HW + (long)y0 * W + q];
        }
    }
    // fused mixer of the next step: its matrix (or gather table) into LDS now, its inputs staged after the tap sums
    float* mixv = reinterpret_cast<float*>(tailp + 1024 + 64);   // [C][tile_px], then [C*C] matrix
    float* mixm = mixv + a.mix_C * tile_px;
    if (a.mix_C && a.mix_matrix)
        for (int e = tid; e < a.mix_C * a.mix_C; e += NTHR) mixm[e] = a.mix_matrix[e];

    // ---- DMA pieces of this wave: q = wid + NWV*i.  A pieces first ((plane, chunk, 64-row group)), then B pieces
    const _Float16* src[PPW];
    long adv[PPW];          // halfs per stage (two 8-channel chunks)
    int ldso[PPW];          // wave-uniform LDS offset (halfs) inside the stage, or -1: dummy
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wid + NWV * i;
        if (q < 4 * pa) {
            const int pc = q / pa, rp = q - pc * pa;   // pc = plane*2 + chunk
            src[i] = (const _Float16*)a.wsh + ((long)blockIdx.y * 2 + (pc >> 1)) * w_plane + ((long)(pc & 1) * Mpad + rp * 64 + lane) * 8;
            adv[i] = (long)2 * Mpad * 8;
            ldso[i] = (pc * Mpad + rp * 64) * 8;
        } else if (q < PT) {
            const int qb = q - 4 * pa;
            const int pc = qb / pb, sp = qb - pc * pb;
            const int slot = sp * 64 + lane;           // window pixel
            const int yy = y0 - 1 + (slot >> wshift);
            const bool ok = slot < Nw && yy >= 0 && yy < H;
            const long gpx = n * HW + (long)(y0 - 1) * W + slot;
            src[i] = ok ? a.x_sh + sh_off(K >> 3, pc >> 1, pc & 1, gpx) : (const _Float16*)a.zeros;
            adv[i] = ok ? 2 * SH_CHUNK_STEP : 0;
            ldso[i] = a_halfs + (pc * Nwpad + sp * 64) * 8;
        } else {
            src[i] = (const _Float16*)a.zeros; adv[i] = 0; ldso[i] = -1;
        }
    }
    auto issue_stage = [&](int kt) {
        _Float16* st = smem_t + (kt % TSH_ST) * stage_halfs;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            _Float16* dst = ldso[i] >= 0 ? st + ldso[i] : dummy;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            src[i] += adv[i];
        }
    };

    f32x16_t accm[MW][NW], accx[MW][NW];
#pragma unroll
    for (int t = 0; t < MW; ++t)
#pragma unroll
        for (int u = 0; u < NW; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[t][u][r] = 0.f; accx[t][u][r] = 0.f; }

    const int nkt = K / 16;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nkt) issue_stage(t);

    GH_STAMP(1);
    const int a_off = (kl * Mpad + wm * MW * 32 + ml) * 8;         // + t*256 ; lo plane: + 2*Mpad*8
    const int b_off = a_halfs + (kl * Nwpad + wn * NW * 32 + ml) * 8;
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt has landed once at most the two younger stages (PPW pieces each) are outstanding; the z2 prefetch loads
        // are older than every stage, so they never hold a counted wait back
        if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt < 8) GH_STAMP(8 + kt);
        const _Float16* st = smem_t + (kt % TSH_ST) * stage_halfs;
        h8 ah[MW], al[MW], bh[NW], bl[NW];
#pragma unroll
        for (int t = 0; t < MW; ++t) {
            ah[t] = *reinterpret_cast<const h8*>(st + a_off + t * 256);
            al[t] = *reinterpret_cast<const h8*>(st + a_off + t * 256 + 2 * Mpad * 8);
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            bh[u] = *reinterpret_cast<const h8*>(st + b_off + u * 256);
            bl[u] = *reinterpret_cast<const h8*>(st + b_off + u * 256 + 2 * Nwpad * 8);
        }
        if (kt + 3 < nkt) issue_stage(kt + 3);
#pragma unroll
        for (int t = 0; t < MW; ++t)
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                accm[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], accm[t][u], 0, 0, 0);
                accx[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], accx[t][u], 0, 0, 0);
                accx[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], accx[t][u], 0, 0, 0);
            }
    }
    GH_STAMP(2);
    __syncthreads();   // every wave is done with the operand ring: it becomes the T staging area
    GH_STAMP(3);

    // ---- T[window pixel][m] -> LDS: a lane's 4 consecutive rows of one pixel are ONE 16-byte store
    float* T = reinterpret_cast<float*>(smem_t);
#pragma unroll
    for (int t = 0; t < MW; ++t)
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int px = (wn * NW + u) * 32 + ml;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = (wm * MW + t) * 32 + 8 * g + 4 * kl;
                if (m < Mrow && px < Nw) {
                    f32x4_t v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = accm[t][u][4 * g + q] + accx[t][u][4 * g + q] * SH_LO_INV;
                    *reinterpret_cast<f32x4_t*>(T + px * Mrow + m) = v;
                }
            }
        }
    __syncthreads();

    GH_STAMP(4);
    // ---- shifted 9-tap sum + (.. + bias) * exp(3 logs) + coupling; one thread per (channel [pair], pixel)
    double ld = 0.0;
#pragma unroll
    for (int i = 0; i < TSH_MAXE; ++i) {
        const int e = tid + i * NTHR;
        if (e >= nitems) continue;
        const int c = e / tile_px, q = e - c * tile_px;
        const int r = q >> wshift, x = q & (W - 1);
        const int ce = paired ? 2 * c : c;
        float se = 0.f, so = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const int tap = dy * 3 + dx + 1;
                const float* tp = T + ((r + dy) * W + xx) * Mrow + tap * Cout + ce;
                se += tp[0];
                if (paired) so += tp[1];
            }
        const int cg = (paired ? c0 / 2 : c0) + c;     // coupling channel in the full tensor
        const long zo = n * a.z2_out_bs + (long)cg * HW + (long)y0 * W + q;
        const float A_ = (se + a.bias[c0 + ce]) * a.scale[c0 + ce];
        float zres;
        if (paired) {
            const float B_ = (so + a.bias[c0 + ce + 1]) * a.scale[c0 + ce + 1];
            const float sc = sigmoidf_(B_ + 2.0f);
            if (a.mode == TAIL_AFFINE_FWD) {
                zres = (zin[i] + A_) * sc;
                ld += (double)logf(sc);
            } else {
                zres = zin[i] / sc - A_;
                ld -= (double)logf(sc);
            }
        } else {
            zres = a.mode == TAIL_ADD_FWD ? zin[i] + A_ : zin[i] - A_;
        }
        if (a.mix_C) {   // ActNorm of the next step on both halves of this pixel's channel c, staged for the mixer
            const int Chh = a.mix_C >> 1;
            mixv[c * tile_px + q] = (z1in[i] + a.mix_bias[c]) * a.mix_scale[c];
            mixv[(Chh + c) * tile_px + q] = (zres + a.mix_bias[Chh + c]) * a.mix_scale[Chh + c];
        } else {
            a.z2_out[zo] = zres;
        }
    }
    if (a.mix_C) {
        __syncthreads();
        const int C = a.mix_C;
        for (int e = tid; e < C * tile_px; e += NTHR) {
            const int o = e / tile_px, q = e - o * tile_px;
            float r;
            if (a.mix_matrix) {   // same operation order as k_chanmix: r = fma(m[o][i], v[i], r), i ascending
                r = 0.f;
                const float* m = mixm + o * C;
                for (int i = 0; i < C; ++i) r = fmaf(m[i], mixv[i * tile_px + q], r);
            } else {
                r = mixv[(a.mix_gather ? a.mix_gather[o] : o) * tile_px + q];
            }
            a.mix_out[n * a.mix_out_bs + (long)o * HW + (long)y0 * W + q] = r;
        }
    }
    GH_STAMP(5);
    if (paired) {
        const double tot = block_sum<NTHR>(ld, red);
        if (tid == 0) fix_atomic_add(a.acc, n, a.N, tot);
    }
}

// ---- configuration: rows per workgroup by image width, channel groups over blockIdx.y, wave grid by tile counts
struct TailShCfg { int MW, NW, PPW, NWV, WGM, R, Mpad, Nwpad, wshift, groups; };

static int g_tail_sh_rows8 = 1;   // 32-pixel-wide levels: 8 image rows per workgroup (halo overhead 10/8 instead of 6/4)
static int g_tail_sh_waves = 0;   // testing hook: 4 / 8 = only that many waves per workgroup (0 = automatic)
void tail_sh_force_waves(int v) { g_tail_sh_waves = v & 12; g_tail_sh_rows8 = !(v & 1); }   // | 1: 4-row workgroups everywhere

// N = batch size when known (launch), 0 at plan time.  The packed weight image depends on Mpad only, which the 8-row variant
// must share with the 4-row one (checked by the caller).
static bool tail_sh_config(int Cin, int H, int W, int Cout, TailShCfg* out, int N = 0) {
    if (Cin % 16 != 0 || Cin < 48) return false;
    // 32-pixel-wide levels: 8 image rows per workgroup when that still gives most CUs a workgroup (halo overhead 10/8
    // instead of 6/4: measured 45 vs 58 us at level 1, B=64)
    int R = g_tail_sh_rows8 && W == 32 && H % 8 == 0 && Cout <= 12 && (long)N * (H / 8) >= 192 ? 8 : 4, wshift;
    if (W == 64) { wshift = 6; R = 2; }   // 128-pixel workgroups: the T staging area (window x 9*Cg floats) has to fit the LDS
    else if (W == 32) wshift = 5;
    else if (W == 16) wshift = 4;
    else if (W == 8) wshift = 3;
    else return false;
    if (H % R != 0) return false;
    // 12 output channels (108 GEMM rows) per workgroup on the deep levels: their many channels become many workgroups
    const int groups = (Cout % 12 == 0 && H * W <= 64) ? Cout / 12 : 1;   // (measured: at 16x16 one group of 24 channels is faster)
    const int Cg = Cout / groups;
    const int Nw = (R + 2) * W;
    const int Mt = (9 * Cg + 31) / 32, Nt = (Nw + 31) / 32;
    // instantiated (MW, NW, PPW, waves); earlier entries win ties
    static const int inst[][4] = {{1, 5, 4, 8}, {1, 3, 3, 8}, {1, 4, 3, 8}, {2, 1, 3, 4} /* 8x8 level: measured faster than {1,1,2,8} */, {1, 1, 2, 8}, {2, 3, 5, 4}, {2, 2, 4, 4}, {2, 2, 5, 4},
                                  {2, 3, 6, 4}, {4, 1, 9, 4}, {2, 1, 5, 4}};
    int best = -1;
    TailShCfg bc{};
    for (size_t i = 0; i < sizeof(inst) / sizeof(inst[0]) && best < 0; ++i) {
        const int NWV = inst[i][3];
        if (g_tail_sh_waves && g_tail_sh_waves != NWV) continue;
        for (int WGM = 1; WGM <= NWV && best < 0; WGM *= 2) {
            const int WGN = NWV / WGM;
            const int MW = (Mt + WGM - 1) / WGM, NW = (Nt + WGN - 1) / WGN;
            const int Mpad = (WGM * MW * 32 + 63) / 64 * 64;
            const int Nwpad = (std::max(Nw, WGN * NW * 32) + 63) / 64 * 64;
            const int PPW = (4 * (Mpad / 64 + Nwpad / 64) + NWV - 1) / NWV;
            if (inst[i][0] == MW && inst[i][1] == NW && inst[i][2] == PPW) {
                best = (int)i;
                bc = TailShCfg{MW, NW, PPW, NWV, WGM, R, Mpad, Nwpad, wshift, groups};
            }
        }
    }
    if (best < 0) return false;
    const size_t ring = (size_t)TSH_ST * 4 * (bc.Mpad + bc.Nwpad) * 8 * sizeof(_Float16);
    const size_t tb = (size_t)Nw * tail_sh_trow(9 * Cg) * sizeof(float);
    if (std::max(ring, tb) + 1024 + 64 > 160 * 1024) return false;
    if (out) *out = bc;
    return true;
}

bool tail_sh_supported(int Cin, int H, int W, int Cout) { return tail_sh_config(Cin, H, W, Cout, nullptr); }

static size_t tail_sh_mix_bytes(int C, int R, int W) { return ((size_t)C * R * W + (size_t)C * C) * sizeof(float); }

// the fused mixer needs every channel of z2 in one workgroup (one channel group) and its scratch inside the LDS
bool tail_sh_mix_supported(int Cin, int H, int W, int Cout, int C) {
    TailShCfg c;
    if (!tail_sh_config(Cin, H, W, Cout, &c) || c.groups != 1 || C > 64) return false;
    const int Nw = (c.R + 2) * W;
    const size_t ring = (size_t)TSH_ST * 4 * (c.Mpad + c.Nwpad) * 8 * sizeof(_Float16);
    const size_t tb = (size_t)Nw * tail_sh_trow(9 * Cout) * sizeof(float);
    return std::max(ring, tb) + 1024 + 64 + tail_sh_mix_bytes(C, c.R, W) <= 160 * 1024;
}

size_t tail_sh_packed_bytes(int Cin, int H, int W, int Cout) {
    TailShCfg c;
    if (!tail_sh_config(Cin, H, W, Cout, &c)) return 0;
    return (size_t)c.groups * 2 * Cin * c.Mpad * sizeof(_Float16);
}

int tail_sh_mpad(int Cin, int H, int W, int Cout, int* groups) {
    TailShCfg c;
    if (!tail_sh_config(Cin, H, W, Cout, &c)) return 0;
    if (groups) *groups = c.groups;
    return c.Mpad;
}

int launch_tail_sh(const TailShArgs& a, hipStream_t s) {
    TailShCfg c, c4;
    GH_REQUIRE(tail_sh_config(a.Cin, a.H, a.W, a.Cout, &c4), "tail_sh: unsupported shape");
    if (!tail_sh_config(a.Cin, a.H, a.W, a.Cout, &c, a.N) || c.Mpad != c4.Mpad || c.groups != c4.groups) c = c4;
    {   // the larger variant also has to fit the fused mixer's scratch
        const size_t ring8 = (size_t)TSH_ST * 4 * (c.Mpad + c.Nwpad) * 8 * sizeof(_Float16);
        const size_t tb8 = (size_t)(c.R + 2) * a.W * tail_sh_trow(9 * (a.Cout / c.groups)) * sizeof(float);
        if (std::max(ring8, tb8) + 1024 + 64 + (a.mix_C ? tail_sh_mix_bytes(a.mix_C, c.R, a.W) : 0) > 160 * 1024) c = c4;
    }
    GH_REQUIRE(a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV || a.mode == TAIL_ADD_FWD || a.mode == TAIL_ADD_REV,
               "tail_sh: coupling modes only");
    if (a.N == 0) return GLOWHIP_OK;
    const int Nw = (c.R + 2) * a.W, Cg = a.Cout / c.groups;
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    GH_REQUIRE((paired ? Cg / 2 : Cg) * c.R * a.W <= TSH_MAXE * 64 * c.NWV, "tail_sh: too many epilogue items per workgroup");
    const size_t ring = (size_t)TSH_ST * 4 * (c.Mpad + c.Nwpad) * 8 * sizeof(_Float16);
    const size_t tb = (size_t)Nw * tail_sh_trow(9 * Cg) * sizeof(float);
    GH_REQUIRE(a.mix_C == 0 || (tail_sh_mix_supported(a.Cin, a.H, a.W, a.Cout, a.mix_C) && a.mix_z1 && a.mix_out && a.mix_bias &&
                                a.mix_scale), "tail_sh: fused mixer unsupported for this shape");
    const size_t lds = std::max(ring, tb) + 1024 + 64 + (a.mix_C ? tail_sh_mix_bytes(a.mix_C, c.R, a.W) : 0);
    const unsigned grid = (unsigned)(a.N * (a.H / c.R));
#define GH_TSH_CASE(mw, nw, ppw, nwv)                                                                                 \
    if (c.MW == mw && c.NW == nw && c.PPW == ppw && c.NWV == nwv) {                                                   \
        (void)hipFuncSetAttribute((const void*)k_tail_sh<mw, nw, ppw, nwv>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                          \
        hipLaunchKernelGGL((k_tail_sh<mw, nw, ppw, nwv>), dim3(grid, c.groups), dim3(64 * nwv), lds, s, a, c.WGM, c.R, \
                           c.Mpad, c.Nwpad, c.wshift, c.groups);                                                      \
        GH_LAUNCH_CHECK("k_tail_sh");                                                                                 \
        return GLOWHIP_OK;                                                                                            \
    }
    GH_TSH_CASE(1, 5, 4, 8) GH_TSH_CASE(1, 3, 3, 8) GH_TSH_CASE(1, 4, 3, 8) GH_TSH_CASE(1, 1, 2, 8)
    GH_TSH_CASE(2, 3, 5, 4) GH_TSH_CASE(2, 3, 6, 4) GH_TSH_CASE(4, 1, 9, 4) GH_TSH_CASE(2, 2, 5, 4) GH_TSH_CASE(2, 2, 4, 4)
    GH_TSH_CASE(2, 1, 5, 4) GH_TSH_CASE(2, 1, 3, 4)
#undef GH_TSH_CASE
    set_error("tail_sh: no kernel instance");
    return GLOWHIP_EINVAL;
}

}  // namespace glowhip
