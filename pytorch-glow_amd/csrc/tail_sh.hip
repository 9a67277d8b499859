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

namespace glowhip {


__device__ __forceinline__ float tsh_gauss_logp1(float mean, float logs, float x) {
    const float d = x - mean;
    return -0.5f * (LOG_2PI_F + 2.0f * logs + (d * d) / expf(2.0f * logs));
}

// KS = 16-deep k-steps per ring stage, ST = ring stages; PPW1 = DMA pieces per wave and k-step
template <int MW, int NW, int PPW1, int KS, int ST>
__global__ void __launch_bounds__(256) k_tail_sh(TailShArgs a, int WGM, int R, int Mpad, int Nwpad, int wshift, int groups) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_t[];
    const int W = a.W, H = a.H, HW = H * W, K = a.Cin;
    const int Cout = a.Cout / groups;                  // output channels of this workgroup's group (blockIdx.y)
    const int c0 = blockIdx.y * Cout;
    const int Nw = (R + 2) * W;
    constexpr int PPW = PPW1 * KS, NCH = 2 * KS;       // pieces per wave and stage; 8-channel chunks per stage
    const int a_halfs = 2 * NCH * Mpad * 8;            // [plane 2][chunk NCH][Mpad][8]
    const int stage_halfs = a_halfs + 2 * NCH * Nwpad * 8;
    const int M9 = 9 * Cout, Nwt = Nw + 8;
    const size_t ring_bytes = (size_t)ST * stage_halfs * sizeof(_Float16);
    const size_t t_bytes = ((size_t)M9 * Nwt * sizeof(float) + 15) & ~(size_t)15;
    char* tailp = (char*)smem_t + (ring_bytes > t_bytes ? ring_bytes : t_bytes);
    _Float16* dummy = (_Float16*)tailp;                // 1 KiB landing area of padding DMA pieces
    double* red = (double*)(tailp + 1024);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WGN = 4 / WGM;
    const int wm = wid / WGN, wn = wid - wm * WGN;
    const int kl = lane >> 5, ml = lane & 31;
    const int bpi = H / R;                             // workgroups per image
    const long n = blockIdx.x / bpi;
    const int y0 = (int)(blockIdx.x - n * bpi) * R;
    const long P = a.P;
    const long w_plane = (long)K * Mpad, x_plane = P * (long)K;
    const int pa = Mpad >> 6, pb = Nwpad >> 6, PT = 2 * NCH * (pa + pb);

    // ---- DMA pieces of this wave: q = wid + 4*i.  A pieces first ((plane, chunk, 64-row group)), then B pieces
    const _Float16* src[PPW];
    long adv[PPW];          // halfs per stage (two 8-channel chunks)
    int ldso[PPW];          // wave-uniform LDS offset (halfs) inside the stage, or -1: dummy
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wid + 4 * i;
        if (q < 2 * NCH * pa) {
            const int pc = q / pa, rp = q - pc * pa;   // pc = plane*NCH + chunk
            const int pl = pc / NCH, chk = pc - pl * NCH;
            src[i] = (const _Float16*)a.wsh + ((long)blockIdx.y * 2 + pl) * w_plane + ((long)chk * Mpad + rp * 64 + lane) * 8;
            adv[i] = (long)NCH * Mpad * 8;
            ldso[i] = (pc * Mpad + rp * 64) * 8;
        } else if (q < PT) {
            const int qb = q - 2 * NCH * pa;
            const int pc = qb / pb, sp = qb - pc * pb;
            const int pl = pc / NCH, chk = pc - pl * NCH;
            const int slot = sp * 64 + lane;           // window pixel
            const int yy = y0 - 1 + (slot >> wshift);
            const bool ok = slot < Nw && yy >= 0 && yy < H;
            const long gpx = n * HW + (long)(y0 - 1) * W + slot;
            src[i] = ok ? a.x_sh + pl * x_plane + ((long)chk * P + gpx) * 8 : (const _Float16*)a.zeros;
            adv[i] = ok ? (long)NCH * P * 8 : 0;
            ldso[i] = a_halfs + (pc * Nwpad + sp * 64) * 8;
        } else {
            src[i] = (const _Float16*)a.zeros; adv[i] = 0; ldso[i] = -1;
        }
    }
    auto issue_stage = [&](int kt) {
        _Float16* st = smem_t + (kt % ST) * stage_halfs;
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

#ifdef GLOWHIP_EXP_TNOLOOP
    const int nkt = ST - 1;
#else
    const int nkt = K / (16 * KS);
#endif
#pragma unroll
    for (int t = 0; t < ST - 1; ++t)
        if (t < nkt) issue_stage(t);

    const int a_off = (kl * Mpad + wm * MW * 32 + ml) * 8;         // + ks*2*Mpad*8 + t*256 ; lo plane: + NCH*Mpad*8
    const int b_off = a_halfs + (kl * Nwpad + wn * NW * 32 + ml) * 8;
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt has landed once at most the ST-2 younger stages (PPW pieces each) are outstanding
        if (kt + ST - 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PPW) : "memory");
        else if (ST == 4 && kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const _Float16* st = smem_t + (kt % ST) * stage_halfs;
        if (kt + ST - 1 < nkt) issue_stage(kt + ST - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h8 ah[MW], al[MW], bh[NW], bl[NW];
#pragma unroll
            for (int t = 0; t < MW; ++t) {
                ah[t] = *reinterpret_cast<const h8*>(st + a_off + ks * 2 * Mpad * 8 + t * 256);
                al[t] = *reinterpret_cast<const h8*>(st + a_off + ks * 2 * Mpad * 8 + t * 256 + NCH * Mpad * 8);
            }
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                bh[u] = *reinterpret_cast<const h8*>(st + b_off + ks * 2 * Nwpad * 8 + u * 256);
                bl[u] = *reinterpret_cast<const h8*>(st + b_off + ks * 2 * Nwpad * 8 + u * 256 + NCH * Nwpad * 8);
            }
#ifdef GLOWHIP_EXP_TNOMFMA
#pragma unroll
            for (int t = 0; t < MW; ++t)
#pragma unroll
                for (int u = 0; u < NW; ++u) { accm[t][u][0] += (float)ah[t][0] * (float)bl[u][0]; accx[t][u][0] += (float)al[t][0] * (float)bh[u][0]; }
#else
#pragma unroll
            for (int t = 0; t < MW; ++t)
#pragma unroll
                for (int u = 0; u < NW; ++u) {
                    accm[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], accm[t][u], 0, 0, 0);
                    accx[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], accx[t][u], 0, 0, 0);
                    accx[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], accx[t][u], 0, 0, 0);
                }
#endif
        }
    }
    __syncthreads();   // every wave is done with the operand ring: it becomes the T staging area

    // ---- T[m][window pixel] -> LDS (row stride Nw + 8 floats: the two half-waves of a store land on disjoint banks)
    float* T = reinterpret_cast<float*>(smem_t);
#pragma unroll
    for (int t = 0; t < MW; ++t)
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int px = (wn * NW + u) * 32 + ml;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * MW + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                if (m < M9 && px < Nw) T[m * Nwt + px] = accm[t][u][r] + accx[t][u][r] * SH_LO_INV;
            }
        }
    __syncthreads();

    // ---- shifted 9-tap sum + (.. + bias) * exp(3 logs) + coupling; one thread per (channel [pair], pixel)
    const int tile_px = R * W;
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    const int nch = paired ? Cout / 2 : Cout;
    double ld = 0.0;
#ifdef GLOWHIP_EXP_TNOEPI
    for (int e = tid; e < nch * tile_px; e += 256 * 64) {
#else
    for (int e = tid; e < nch * tile_px; e += 256) {
#endif
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
                const float* tp = T + (tap * Cout + ce) * Nwt + (r + dy) * W + xx;
                se += tp[0];
                if (paired) so += tp[Nwt];
            }
        const int p = (y0 + r) * W + x;
        const int cg = (paired ? c0 / 2 : c0) + c;     // coupling channel in the full tensor
        const long zi = n * a.z2_in_bs + (long)cg * HW + p;
        const long zo = n * a.z2_out_bs + (long)cg * HW + p;
        const float A_ = (se + a.bias[c0 + ce]) * a.scale[c0 + ce];
        if (paired) {
            const float B_ = (so + a.bias[c0 + ce + 1]) * a.scale[c0 + ce + 1];
            const float sc = sigmoidf_(B_ + 2.0f);
            if (a.mode == TAIL_AFFINE_FWD) {
                a.z2_out[zo] = (a.z2_in[zi] + A_) * sc;
                ld += (double)logf(sc);
            } else {
                a.z2_out[zo] = a.z2_in[zi] / sc - A_;
                ld -= (double)logf(sc);
            }
        } else {
            const float z2 = a.z2_in[zi];
            a.z2_out[zo] = a.mode == TAIL_ADD_FWD ? z2 + A_ : z2 - A_;
        }
    }
    if (paired) {
        const double tot = block_sum<256>(ld, red);
        if (tid == 0) fix_atomic_add(a.acc + n, tot);
    }
}

// ---- configuration: rows per workgroup by image width, channel groups over blockIdx.y, wave grid by tile counts
struct TailShCfg { int MW, NW, PPW, WGM, R, Mpad, Nwpad, wshift, groups, KS, ST; };

static int g_tail_sh_ks = 0;   // testing hook: force k-steps per stage (0 = automatic)
void tail_sh_force_ks(int ks) { g_tail_sh_ks = ks; }

static bool tail_sh_config(int Cin, int H, int W, int Cout, TailShCfg* out) {
    if (Cin % 16 != 0 || Cin < 48) return false;
    int R = 4, wshift;
    if (W == 32) wshift = 5;
    else if (W == 16) wshift = 4;
    else if (W == 8) wshift = 3;
    else return false;
    if (H % R != 0) return false;
    // 12 output channels (108 GEMM rows) per workgroup: the deep levels' many channels become many workgroups, and
    // a workgroup's weight stream stays at 2 DMA pieces per wave and stage
    const int groups = (Cout % 12 == 0 && H * W <= 64) ? Cout / 12 : 1;   // (measured: at 16x16 one group of 24 channels is faster)
    const int Cg = Cout / groups;
    const int Nw = (R + 2) * W;
    const int Mt = (9 * Cg + 31) / 32, Nt = (Nw + 31) / 32;
    static const int inst[][3] = {{2, 3, 5}, {2, 2, 4}, {2, 1, 3}, {2, 2, 5}, {2, 3, 6}, {4, 1, 9}, {2, 1, 5}};
    int best = -1, best_cost = 1 << 30;
    TailShCfg bc{};
    for (int WGM = 1; WGM <= 4; WGM *= 2) {
        const int WGN = 4 / WGM;
        const int MW = (Mt + WGM - 1) / WGM, NW = (Nt + WGN - 1) / WGN;
        const int Mpad = (WGM * MW * 32 + 63) / 64 * 64;
        const int Nwpad = (std::max(Nw, WGN * NW * 32) + 63) / 64 * 64;
        const int PPW = Mpad / 64 + Nwpad / 64;
        for (size_t i = 0; i < sizeof(inst) / sizeof(inst[0]); ++i)
            if (inst[i][0] == MW && inst[i][1] == NW && inst[i][2] == PPW && MW * NW < best_cost) {
                best = (int)i; best_cost = MW * NW;
                bc = TailShCfg{MW, NW, PPW, WGM, R, Mpad, Nwpad, wshift, groups, 1, 4};
            }
    }
    if (best < 0) return false;
    // ring geometry: 16-deep stages x 4 by default; 32/64-deep variants exist for A/B runs (instantiated combinations only)
    int ks = g_tail_sh_ks ? g_tail_sh_ks : 1;   // measured: deeper stages (32/64) are slower on every level
    if (!((bc.MW == 2 && bc.NW == 1 && bc.PPW == 3 && ks == 4) || (bc.MW == 2 && bc.NW == 3 && ks == 2))) ks = 1;
    if (Cin % (16 * ks) != 0) ks = 1;
    bc.KS = ks; bc.ST = ks == 1 ? 4 : 3;
    const size_t ring = (size_t)bc.ST * 4 * bc.KS * (bc.Mpad + bc.Nwpad) * 8 * sizeof(_Float16);
    const size_t tb = align_up((size_t)9 * Cg * (Nw + 8) * sizeof(float), 16);
    if (std::max(ring, tb) + 1024 + 64 > 160 * 1024) return false;
    if (out) *out = bc;
    return true;
}

bool tail_sh_supported(int Cin, int H, int W, int Cout) { return tail_sh_config(Cin, H, W, Cout, nullptr); }

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
    TailShCfg c;
    GH_REQUIRE(tail_sh_config(a.Cin, a.H, a.W, a.Cout, &c), "tail_sh: unsupported shape");
    GH_REQUIRE(a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV || a.mode == TAIL_ADD_FWD || a.mode == TAIL_ADD_REV,
               "tail_sh: coupling modes only");
    if (a.N == 0) return GLOWHIP_OK;
    const int Nw = (c.R + 2) * a.W;
    const size_t ring = (size_t)c.ST * 4 * c.KS * (c.Mpad + c.Nwpad) * 8 * sizeof(_Float16);
    const size_t tb = align_up((size_t)9 * (a.Cout / c.groups) * (Nw + 8) * sizeof(float), 16);
    const size_t lds = std::max(ring, tb) + 1024 + 64;
    const unsigned grid = (unsigned)(a.N * (a.H / c.R));
#define GH_TSH_CASE(mw, nw, ppw, ks, st)                                                                              \
    if (c.MW == mw && c.NW == nw && c.PPW == ppw && c.KS == ks && c.ST == st) {                                       \
        (void)hipFuncSetAttribute((const void*)k_tail_sh<mw, nw, ppw, ks, st>,                                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        hipLaunchKernelGGL((k_tail_sh<mw, nw, ppw, ks, st>), dim3(grid, c.groups), dim3(256), lds, s, a, c.WGM, c.R,  \
                           c.Mpad, c.Nwpad, c.wshift, c.groups);                                                      \
        GH_LAUNCH_CHECK("k_tail_sh");                                                                                 \
        return GLOWHIP_OK;                                                                                            \
    }
    GH_TSH_CASE(2, 3, 5, 1, 4) GH_TSH_CASE(2, 3, 6, 1, 4) GH_TSH_CASE(4, 1, 9, 1, 4) GH_TSH_CASE(2, 2, 5, 1, 4)
    GH_TSH_CASE(2, 2, 4, 1, 4) GH_TSH_CASE(2, 1, 5, 1, 4) GH_TSH_CASE(2, 1, 3, 1, 4)
    GH_TSH_CASE(2, 1, 3, 4, 3) GH_TSH_CASE(2, 3, 5, 2, 3) GH_TSH_CASE(2, 3, 6, 2, 3)
#undef GH_TSH_CASE
    set_error("tail_sh: no kernel instance");
    return GLOWHIP_EINVAL;
}

}  // namespace glowhip
