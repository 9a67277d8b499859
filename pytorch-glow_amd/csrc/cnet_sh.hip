// cnet_sh.hip -- the whole coupling network  h = f(z1) = f.4(relu(f.2(relu(f.0(z1)))))  (network/module.py:300-319) as ONE
// kernel on true-scale split-half operands (SH2, sh.h), followed by a light finishing kernel that applies the coupling
// (network/model.py:105-113 / 131-150), the per-sample log-det and the adjacent channel mixer (ActNorm + invertible 1x1 conv or
// permutation, network/module.py:122-149, 344-369, 392-397).  Neither h1 nor h2 ever reaches HBM.
//
// Workgroup = 128 pixels x (hidden / MS) rows of h2, 8 waves, one per CU (LDS: 128 KiB activation buffer + the z1 window).
//   P0   z1 window (tile rows + one halo row/column each side, zero padded) -> (hi, lo) halves in LDS
//   P1   h1 = relu(conv3x3(z1; W0') + b0') by MFMA, 256 channels at a time (the LDS buffer holds 256 channels x 128 pixels as
//        (hi, lo) halves in the B-operand layout [plane][chunk][pixel][8]); weights as A fragments straight from L2
//   P2   h2 accumulators += W2'[:, those 256 channels] h1 -- B resident in LDS, A two k-steps ahead from L2; a wave owns
//        64 rows x 128 pixels (8 MFMA tiles, ONE fp32 accumulator each).  P1/P2 alternate over the two channel halves.
//        128 pixels per workgroup instead of 64 halve the weight bytes fetched per MFMA, which is what bounded k_f02_sh
//        (vector-memory return path, ~30 B/clk/CU); the single accumulator of SH2 is what makes the 128 x 512 tile fit.
//   P3   f.4 with the filter taps moved to the OUTPUT side (tail_sh.hip): T[tap*Cout + co][px] = sum_k W4[co][k][tap] h2[k][px],
//        h2 passed to the B side through the same LDS buffer (256 channels at a time), T accumulated in registers
//   P4   T -> LDS, shifted 9-tap sums.  Rows of the tile's own pixels go to `hpart`; what the tile's first / last image row
//        contributes to the rows just outside the tile goes to `hup` / `hdn` (no halo recompute, no atomics: deterministic).
// MS > 1 splits the h2 rows (= f.4's reduction axis) over MS workgroups per tile, each recomputing h1 -- for the levels whose
// pixel count alone cannot fill 256 CUs.  The finishing kernel sums the MS partials and the neighbours' halo rows.
#include "sh.h"
#include <algorithm>

#include "conv_mfma.h"

namespace glowhip {

constexpr int CN_PX = 128;                   // pixels per workgroup tile
constexpr int CN_HBUF = 128 * 1024;          // bytes of the h1 / h2 / T region
constexpr int CN_MAXMS = 4;

struct CnetGeo {
    int wshift, lsub, NI, R, WP, Wpx, nchunk, G, steps0, Mpad4, Mrow, NRT4, KS, npass, tiles;
    int winplane;     // halfs per window plane
    int HW;
};

__host__ __device__ inline int cnet_trow(int M9) {   // T row stride (floats): multiple of 4, an odd multiple (bank spread)
    int r = (M9 + 3) / 4;
    if ((r & 1) == 0) ++r;
    return r * 4;
}

template <int HID, int MS, int UPW>
__global__ void __launch_bounds__(512) k_cnet(CnetArgs a, CnetGeo g) {
    constexpr int NH = HID > 256 ? 2 : 1;            // channel halves of h1 (the LDS buffer holds 256 channels x 128 px)
    constexpr int HK = HID / NH;                     // channels per half
    constexpr int NCH = HK / 8;                      // 8-channel chunks per half
    constexpr int MR = HID / MS;                     // h2 rows of this workgroup
    constexpr int TP1 = (HK / 32) * 4 / 8;           // P1 tiles per wave and half: 4 / 2 / 1
    constexpr int TP2 = (MR / 32) * 4 / 8;           // P2 tiles per wave: 8 / 4 / 2 / 1
    static_assert(TP1 >= 1 && TP2 >= 1, "hidden / MS too small for 8 waves");
    constexpr int RT2 = TP2 >= 4 ? TP2 / 4 : 1;      // row tiles x pixel tiles of a wave's h2 block
    constexpr int PT2 = TP2 >= 4 ? 4 : TP2;
    constexpr int P1SUB = (TP2 == 8 && TP1 == 4) ? 2 : 1;   // P1 in two pixel sub-passes while 128 accumulator registers are live
    constexpr int PTS = TP1 / P1SUB;                 // pixel tiles per P1 sub-pass
    constexpr int NL = MR > 256 ? 2 : 1;             // loads of h2 into the LDS buffer for P3
    constexpr int LK = MR / NL;                      // channels per load

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_c[];
    _Float16* hbuf = smem_c;
    _Float16* win = smem_c + CN_HBUF / 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, ml = lane & 31;
    const int W = a.W, H = a.H, HW = g.HW;
    const int tb = blockIdx.x;
    const int ms_row0 = blockIdx.y * MR;
    const long gp0 = (long)tb * CN_PX;
    // tile origin: one image (NI = 1: R rows from y0) or NI whole images
    const long n0 = g.NI == 1 ? gp0 / HW : (long)tb * g.NI;
    const int y0 = g.NI == 1 ? (int)((gp0 - n0 * HW) >> g.wshift) : 0;
    const int submask = (1 << g.lsub) - 1;

    // ---- A-operand bases
    const _Float16* W0 = (const _Float16*)a.w0;
    const long w0_plane = (long)g.G * HID * 8;
    const float* rs0 = (const float*)((const char*)a.w0 + sh2_rowscale_off(g.G * 8, HID));
    const float* b0 = rs0 + HID;
    const _Float16* W2 = (const _Float16*)a.w2;
    constexpr long w2_plane = (long)HID * HID;
    const float* rs2 = (const float*)((const char*)a.w2 + sh2_rowscale_off(HID, HID));
    const float* b2 = rs2 + HID;

    // ---- wave's h2 block: row tiles [rt2, rt2 + RT2), pixel tiles [pt2, pt2 + PT2)
    const int rt2 = (wid * TP2) >> 2, pt2 = (wid * TP2) & 3;
    const _Float16* a2p = W2 + ((long)kl * HID + ms_row0 + rt2 * 32 + ml) * 8;      // + ks * 2*HID*8 ; + i*256 ; lo: + w2_plane
    h8 A2[3][2 * RT2];     // three k-steps of A fragments in flight: [set][i] hi, [set][RT2 + i] lo
    auto loadA2 = [&](int ks, h8 (&dst)[2 * RT2]) {
        const _Float16* p = a2p + (long)ks * (2 * HID * 8);
#pragma unroll
        for (int i = 0; i < RT2; ++i) {
            dst[i] = *reinterpret_cast<const h8*>(p + i * 256);
            dst[RT2 + i] = *reinterpret_cast<const h8*>(p + i * 256 + w2_plane);
        }
    };
    // requested before the window is built: the first two A sets of P2 (their L2 round trips overlap P0 and P1)
    loadA2(0, A2[0]);
    loadA2(1, A2[1]);

    // ---- P0: window -> (hi, lo) halves in LDS; slot e = (chunk, sub-tile, window pixel), 8 channels each
    {
        const int nslots = g.nchunk * g.NI * g.Wpx;
        for (int e = tid; e < nslots; e += 512) {
            const int ch = e / (g.NI * g.Wpx);
            const int rem = e - ch * (g.NI * g.Wpx);
            const int sub = rem / g.Wpx, wp = rem - sub * g.Wpx;
            const int r = wp / g.WP, c = wp - r * g.WP;
            const int yy = y0 - 1 + r, xx = c - 1;
            const long n = n0 + sub;
            const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W && n < a.N;
            const float* xin = a.x + n * a.x_bs;
            h8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ci = ch * 8 + q;
                const float v = (in && ci < a.Cin) ? xin[(long)ci * HW + yy * W + xx] * SH2_ACT_SCALE : 0.f;
                _Float16 x0, x1;
                sh2_split(v, x0, x1);
                hi[q] = x0; lo[q] = x1;
            }
            *reinterpret_cast<h8*>(win + (long)e * 8) = hi;
            *reinterpret_cast<h8*>(win + g.winplane + (long)e * 8) = lo;
        }
    }
    __syncthreads();

    // window offset (halfs) of tile pixel q for tap (0,0), chunk 0
    auto pix_base = [&](int q) {
        const int sub = q >> g.lsub, qq = q & submask;
        return (sub * g.Wpx + (qq >> g.wshift) * g.WP + (qq & (W - 1))) * 8;
    };

    f32x16_t acc2[RT2][PT2];
#pragma unroll
    for (int i = 0; i < RT2; ++i)
#pragma unroll
        for (int j = 0; j < PT2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;

#pragma unroll 1
    for (int hh = 0; hh < NH; ++hh) {
        // ---- P1: h1 channels [hh*HK, hh*HK + HK) of all 128 pixels -> hbuf
        {
            const int t0 = wid * TP1;
            const int rt1 = t0 >> 2, pt1 = t0 & 3;
            const int o0 = hh * HK + rt1 * 32;
            const _Float16* a0p = W0 + ((long)kl * HID + o0 + ml) * 8;     // + st * 2*HID*8 ; lo: + w0_plane
#pragma unroll 1
            for (int sp = 0; sp < P1SUB; ++sp) {
                int pb[PTS];
#pragma unroll
                for (int j = 0; j < PTS; ++j) pb[j] = pix_base((pt1 + sp * PTS + j) * 32 + ml);
                f32x16_t acc1[PTS];
#pragma unroll
                for (int j = 0; j < PTS; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
                h8 Ah0 = *reinterpret_cast<const h8*>(a0p), Al0 = *reinterpret_cast<const h8*>(a0p + w0_plane);
                h8 Ah1 = Ah0, Al1 = Al0;
                if (g.steps0 > 1) {
                    Ah1 = *reinterpret_cast<const h8*>(a0p + 2 * HID * 8);
                    Al1 = *reinterpret_cast<const h8*>(a0p + 2 * HID * 8 + w0_plane);
                }
#pragma unroll 1
                for (int st = 0; st < g.steps0; ++st) {
                    h8 Ah2 = Ah1, Al2 = Al1;
                    if (st + 2 < g.steps0) {
                        Ah2 = *reinterpret_cast<const h8*>(a0p + (long)(st + 2) * (2 * HID * 8));
                        Al2 = *reinterpret_cast<const h8*>(a0p + (long)(st + 2) * (2 * HID * 8) + w0_plane);
                    }
                    int gk = 2 * st + kl;
                    gk = gk < 9 * g.nchunk ? gk : 0;           // padded groups carry zero weights
                    const int tap = gk / g.nchunk, ch = gk - tap * g.nchunk;
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const int goff = (ch * g.NI * g.Wpx + dy * g.WP + dx) * 8;
#pragma unroll
                    for (int j = 0; j < PTS; ++j) {
                        const h8 bh = *reinterpret_cast<const h8*>(win + goff + pb[j]);
                        const h8 bl = *reinterpret_cast<const h8*>(win + g.winplane + goff + pb[j]);
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah0, bh, acc1[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah0, bl, acc1[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al0, bh, acc1[j], 0, 0, 0);
                    }
                    Ah0 = Ah1; Al0 = Al1; Ah1 = Ah2; Al1 = Al2;
                }
                // relu, split, store into the B-operand image: a lane's 4 consecutive channels = 8 bytes per plane
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int o = o0 + 8 * gq + 4 * kl;
                    const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(rs0 + o);
                    const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(b0 + o);
                    const int chunk = rt1 * 4 + gq;
#pragma unroll
                    for (int j = 0; j < PTS; ++j) {
                        h4 hi, lo;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float v = fmaxf(acc1[j][4 * gq + t] * rs[t] + bb[t], 0.f);
                            _Float16 x0, x1;
                            sh2_split(v, x0, x1);
                            hi[t] = x0; lo[t] = x1;
                        }
                        _Float16* dst = hbuf + ((long)chunk * CN_PX + (pt1 + sp * PTS + j) * 32 + ml) * 8 + 4 * kl;
                        *reinterpret_cast<h4*>(dst) = hi;
                        *reinterpret_cast<h4*>(dst + (long)NCH * CN_PX * 8) = lo;
                    }
                }
            }
        }
        __syncthreads();

        // ---- P2: acc2 += W2'[rows, half hh] h1[half hh]; B from LDS, A two k-steps ahead from L2
        {
            constexpr int NS = HK / 16;
            const int ks0 = hh * NS;
            const _Float16* bp = hbuf + ((long)kl * CN_PX + pt2 * 32 + ml) * 8;
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
                if (ks0 + s + 2 < HID / 16) loadA2(ks0 + s + 2, A2[2]);
                const _Float16* bs = bp + (long)s * (2 * CN_PX * 8);
                h8 bh[PT2], bl[PT2];
#pragma unroll
                for (int j = 0; j < PT2; ++j) {
                    bh[j] = *reinterpret_cast<const h8*>(bs + j * 256);
                    bl[j] = *reinterpret_cast<const h8*>(bs + j * 256 + (long)NCH * CN_PX * 8);
                }
#pragma unroll
                for (int i = 0; i < RT2; ++i)
#pragma unroll
                    for (int j = 0; j < PT2; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2[0][i], bh[j], acc2[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT2; ++i)
#pragma unroll
                    for (int j = 0; j < PT2; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2[0][i], bl[j], acc2[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT2; ++i)
#pragma unroll
                    for (int j = 0; j < PT2; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2[0][RT2 + i], bh[j], acc2[i][j], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2 * RT2; ++t) { A2[0][t] = A2[1][t]; A2[1][t] = A2[2][t]; }
            }
        }
        __syncthreads();     // every wave is done reading this half of h1
    }

    // ---- h2 = relu(acc2 * rowscale + bias) (times SH2_ACT_SCALE), in place
#pragma unroll
    for (int i = 0; i < RT2; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int o = ms_row0 + (rt2 + i) * 32 + 8 * gq + 4 * kl;
            const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(rs2 + o);
            const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(b2 + o);
#pragma unroll
            for (int j = 0; j < PT2; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc2[i][j][4 * gq + t] = fmaxf(acc2[i][j][4 * gq + t] * rs[t] + bb[t], 0.f);
        }

    if (a.y_sh) {   // testing: h2 as an (old-format) SH tensor, f.4 left to k_tail_sh
#pragma unroll
        for (int i = 0; i < RT2; ++i)
#pragma unroll
            for (int j = 0; j < PT2; ++j) {
                const long px = gp0 + (pt2 + j) * 32 + ml;
                if (px >= (long)a.N * HW) continue;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int o = ms_row0 + (rt2 + i) * 32 + 8 * gq + 4 * kl;
                    h4 hi, lo;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        _Float16 x0, x1;
                        sh_split(acc2[i][j][4 * gq + t] * SH2_ACT_INV, x0, x1);
                        hi[t] = x0; lo[t] = x1;
                    }
                    _Float16* dst = a.y_sh + sh_off(HID / 8, 0, o >> 3, px) + (o & 7);
                    *reinterpret_cast<h4*>(dst) = hi;
                    *reinterpret_cast<h4*>(dst + (long)(HID / 8) * SH_CHUNK_STEP) = lo;
                }
            }
        return;
    }

    // ---- P3: T[m][px] = sum_k W4t[m][k] h2[k][px] over this workgroup's h2 rows (= k range [ms_row0, ms_row0 + MR))
    // unit u = (row tile rt4 = u % NRT4, k part u / NRT4) x all 4 pixel tiles; wave w takes units w (and w + 8 when UPW = 2)
    const _Float16* W4 = (const _Float16*)a.w4;
    const long w4_plane = (long)HID * g.Mpad4;
    const float* rs4 = (const float*)((const char*)a.w4 + sh2_rowscale_off(HID, g.Mpad4));
    const int nunits = g.NRT4 * g.KS;
    f32x16_t accT[UPW][4];
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accT[u][j][r] = 0.f;
    constexpr int LCH = LK / 8;        // chunks per load
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        // the owners of rows [l*LK, (l+1)*LK) pass their h2 to the B side through hbuf
#pragma unroll
        for (int i = 0; i < RT2; ++i) {
            const int wr = (rt2 + i) * 32;                   // first workgroup-local row of this tile
            if (wr / LK != l) continue;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int chunk = (wr - l * LK) / 8 + gq;
#pragma unroll
                for (int j = 0; j < PT2; ++j) {
                    h4 hi, lo;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        _Float16 x0, x1;
                        sh2_split(acc2[i][j][4 * gq + t], x0, x1);
                        hi[t] = x0; lo[t] = x1;
                    }
                    _Float16* dst = hbuf + ((long)chunk * CN_PX + (pt2 + j) * 32 + ml) * 8 + 4 * kl;
                    *reinterpret_cast<h4*>(dst) = hi;
                    *reinterpret_cast<h4*>(dst + (long)LCH * CN_PX * 8) = lo;
                }
            }
        }
        __syncthreads();
        const int nsl = (LK / 16) / g.KS;                    // k-steps of this load per k part
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int unit = wid + 8 * u;
            if (unit >= nunits) continue;
            const int rt4 = unit % g.NRT4, kp = unit / g.NRT4;
            const int s0 = kp * nsl;
            const _Float16* ap = W4 + ((long)((ms_row0 + l * LK) / 8 + 2 * s0 + kl) * g.Mpad4 + rt4 * 32 + ml) * 8;
            const _Float16* bp = hbuf + ((long)(2 * s0 + kl) * CN_PX + ml) * 8;
            h8 ah0 = *reinterpret_cast<const h8*>(ap), al0 = *reinterpret_cast<const h8*>(ap + w4_plane);
            h8 ah1 = ah0, al1 = al0;
            if (nsl > 1) {
                ah1 = *reinterpret_cast<const h8*>(ap + (long)2 * g.Mpad4 * 8);
                al1 = *reinterpret_cast<const h8*>(ap + (long)2 * g.Mpad4 * 8 + w4_plane);
            }
#pragma unroll 1
            for (int s = 0; s < nsl; ++s) {
                h8 ah2 = ah1, al2 = al1;
                if (s + 2 < nsl) {
                    ah2 = *reinterpret_cast<const h8*>(ap + (long)(s + 2) * (2 * g.Mpad4 * 8));
                    al2 = *reinterpret_cast<const h8*>(ap + (long)(s + 2) * (2 * g.Mpad4 * 8) + w4_plane);
                }
                const _Float16* bs = bp + (long)s * (2 * CN_PX * 8);
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    h8 bh[2], bl[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        bh[jj] = *reinterpret_cast<const h8*>(bs + (2 * jp + jj) * 256);
                        bl[jj] = *reinterpret_cast<const h8*>(bs + (2 * jp + jj) * 256 + (long)LCH * CN_PX * 8);
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        accT[u][2 * jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[jj], accT[u][2 * jp + jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        accT[u][2 * jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[jj], accT[u][2 * jp + jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        accT[u][2 * jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[jj], accT[u][2 * jp + jj], 0, 0, 0);
                }
                ah0 = ah1; al0 = al1; ah1 = ah2; al1 = al2;
            }
        }
        __syncthreads();     // hbuf free again (next load / T staging)
    }

    // ---- P4: T -> LDS [pixel][Mrow] (fp32, row scale applied), k parts summed in a fixed order, then the 9-tap sums
    float* T = reinterpret_cast<float*>(hbuf);
    const int Mrow = g.Mrow, Cout = a.Cout;
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    const int nch = paired ? Cout / 2 : Cout;
    const int ppx = CN_PX / g.npass;                         // pixels per staging pass (whole sub-tiles when npass = 2)
    const long msN = (long)blockIdx.y * a.N;
    float* hpart = a.scratch;
    float* hup = a.scratch + (long)MS * a.N * Cout * HW;
    float* hdn = hup + (long)MS * g.tiles * Cout * W;
#pragma unroll 1
    for (int pass = 0; pass < g.npass; ++pass) {
#pragma unroll 1
        for (int kp = 0; kp < g.KS; ++kp) {
#pragma unroll
            for (int u = 0; u < UPW; ++u) {
                const int unit = wid + 8 * u;
                if (unit >= nunits || unit / g.NRT4 != kp) continue;
                const int rt4 = unit % g.NRT4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = j * 32 + ml;
                    if (q / ppx != pass) continue;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int m = rt4 * 32 + 8 * gq + 4 * kl;
                        if (m >= Mrow) continue;
                        const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(rs4 + m);
                        f32x4_t* dst = reinterpret_cast<f32x4_t*>(T + (long)(q - pass * ppx) * Mrow + m);
                        f32x4_t v;
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = accT[u][j][4 * gq + t] * rs[t];
                        if (kp > 0) {
                            const f32x4_t o = *dst;
#pragma unroll
                            for (int t = 0; t < 4; ++t) v[t] += o[t];
                        }
                        *dst = v;
                    }
                }
            }
            __syncthreads();
        }
        // own rows: out[c][r][x] = sum over taps whose source row r + dy - 1 lies inside the sub-tile
        const int items = nch * ppx;
        for (int e = tid; e < items; e += 512) {
            const int c = e / ppx, ql = e - c * ppx;
            const int q = pass * ppx + ql;
            const int sub = q >> g.lsub, qq = q & submask;
            const int r = qq >> g.wshift, x = qq & (W - 1);
            const long n = n0 + sub;
            if (n >= a.N) continue;
            const int ce = paired ? 2 * c : c;
            float se = 0.f, so = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int rs_ = r + dy - 1;
                if (rs_ < 0 || rs_ >= g.R) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xs = x + dx - 1;
                    if (xs < 0 || xs >= W) continue;
                    const int tap = dy * 3 + dx;              // out(r, x) += T[tap (dy, dx)][source (r + dy - 1, x + dx - 1)]
                    const float* tp = T + (long)(ql + (dy - 1) * W + (dx - 1)) * Mrow + tap * Cout + ce;
                    se += tp[0];
                    if (paired) so += tp[1];
                }
            }
            const long base = ((msN + n) * Cout + ce) * HW + (long)(y0 + r) * W + x;
            hpart[base] = se;
            if (paired) hpart[base + HW] = so;
        }
        // halo rows (NI = 1 only): what the tile's first row gives to image row y0 - 1, its last row to row y0 + R
        if (g.NI == 1 && g.R < H) {
            const int hitems = 2 * Cout * W;
            for (int e = tid; e < hitems; e += 512) {
                const int dn = e / (Cout * W);
                const int rem = e - dn * (Cout * W);
                const int co = rem / W, x = rem - co * W;
                if (dn ? (y0 + g.R >= H) : (y0 == 0)) continue;
                const int rsrc = dn ? g.R - 1 : 0;
                const int dyt = dn ? 0 : 2;                 // filter row applied by the outside pixel to this source row
                float sacc = 0.f;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xs = x + dx - 1;
                    if (xs < 0 || xs >= W) continue;
                    sacc += T[(long)(rsrc * W + xs) * Mrow + (dyt * 3 + dx) * Cout + co];
                }
                (dn ? hdn : hup)[(((long)blockIdx.y * g.tiles + tb) * Cout + co) * W + x] = sacc;
            }
        }
        if (pass + 1 < g.npass) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ finishing kernel
// One workgroup = 64 consecutive pixels of one image x all channels: sums the MS partials and the neighbour tiles' halo rows,
// (h + bias) * exp(3 logs), coupling, per-sample log-det, then the channel mixer on the finished pixels.
struct CfinArgs {
    CnetArgs a;
    int MS, tiles, R, NI, wshift, HW;
};

__global__ void __launch_bounds__(256) k_cfinish(CfinArgs f) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];   // [C][64] values, then [C*C] matrix
    __shared__ double red[4];
    const CnetArgs& a = f.a;
    const int tid = threadIdx.x;
    const int HW = f.HW, W = a.W, H = a.H, Cout = a.Cout;
    const long gp0 = (long)blockIdx.x * 64;
    const long n = gp0 / HW;
    const int p0 = (int)(gp0 - n * HW);
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    const int Ch = paired ? Cout / 2 : Cout;         // channels of z2 (= C/2)
    const int C = 2 * Ch;
    float* mixv = fsm;
    float* mixm = fsm + C * 64;
    if (a.mix_C && a.mix_matrix)
        for (int e = tid; e < C * C; e += 256) mixm[e] = a.mix_matrix[e];
    const float* hpart = a.scratch;
    const float* hup = a.scratch + (long)f.MS * a.N * Cout * HW;
    const float* hdn = hup + (long)f.MS * f.tiles * Cout * W;
    const float* zi = a.z_in + n * a.z_in_bs;
    float* zn = a.z_out + n * a.z_out_bs;
    double ld = 0.0;
    for (int e = tid; e < Ch * 64; e += 256) {
        const int c = e >> 6, q = e & 63;
        const int p = p0 + q;
        const int y = p >> f.wshift, x = p & (W - 1);
        const int ce = paired ? 2 * c : c;
        float se = 0.f, so = 0.f;
        for (int m = 0; m < f.MS; ++m) {
            const long base = (((long)m * a.N + n) * Cout + ce) * HW + p;
            se += hpart[base];
            if (paired) so += hpart[base + HW];
        }
        if (f.NI == 1 && f.R < H) {
            const int r = y & (f.R - 1);
            const long tile = (n * HW + (long)(y - r) * W) >> 7;       // 128-pixel tile holding row y
            if (r == 0 && y > 0)
                for (int m = 0; m < f.MS; ++m) {
                    const long hb = (((long)m * f.tiles + tile - 1) * Cout + ce) * W + x;
                    se += hdn[hb];
                    if (paired) so += hdn[hb + W];
                }
            if (r == f.R - 1 && y < H - 1)
                for (int m = 0; m < f.MS; ++m) {
                    const long hb = (((long)m * f.tiles + tile + 1) * Cout + ce) * W + x;
                    se += hup[hb];
                    if (paired) so += hup[hb + W];
                }
        }
        const float zin = zi[(long)(Ch + c) * HW + p];
        const float A_ = (se + a.bias[ce]) * a.scale[ce];
        float zres;
        if (paired) {
            const float B_ = (so + a.bias[ce + 1]) * a.scale[ce + 1];
            const float sc = sigmoidf_(B_ + 2.0f);
            if (a.mode == TAIL_AFFINE_FWD) {
                zres = (zin + A_) * sc;
                ld += (double)logf(sc);
            } else {
                zres = zin / sc - A_;
                ld -= (double)logf(sc);
            }
        } else {
            zres = a.mode == TAIL_ADD_FWD ? zin + A_ : zin - A_;
        }
        if (a.mix_C) {
            const float z1v = zi[(long)c * HW + p];
            if (!a.mix_reverse) {     // ActNorm of the next step on both halves, staged for its matrix / gather
                mixv[c * 64 + q] = (z1v + a.mix_bias[c]) * a.mix_scale[c];
                mixv[(Ch + c) * 64 + q] = (zres + a.mix_bias[Ch + c]) * a.mix_scale[Ch + c];
            } else {
                mixv[c * 64 + q] = z1v;
                mixv[(Ch + c) * 64 + q] = zres;
            }
        } else {
            zn[(long)(Ch + c) * HW + p] = zres;
        }
    }
    if (a.mix_C) {
        __syncthreads();
        for (int e = tid; e < C * 64; e += 256) {
            const int o = e >> 6, q = e & 63;
            float r;
            if (a.mix_matrix) {   // same operation order as k_chanmix: r = fma(m[o][i], v[i], r), i ascending
                r = 0.f;
                const float* m = mixm + o * C;
                for (int i = 0; i < C; ++i) r = fmaf(m[i], mixv[i * 64 + q], r);
            } else {
                r = mixv[(a.mix_gather ? a.mix_gather[o] : o) * 64 + q];
            }
            if (a.mix_reverse) r = r * a.mix_scale[o] - a.mix_bias[o];
            zn[(long)o * HW + p0 + q] = r;
        }
    }
    if (paired) {
        const double tot = block_sum<256>(ld, red);
        if (tid == 0) fix_atomic_add(a.acc + n, tot);
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int g_cnet_ms = 0;
void cnet_force(int ms, int flags) { g_cnet_ms = ms; (void)flags; }

int cnet_g0(int Cin) { return (9 * ((Cin + 7) / 8) + 1) & ~1; }
int cnet_mpad4(int Cout) { return (9 * Cout + 31) / 32 * 32; }

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static bool cnet_geo(int Cin, int H, int W, int hidden, int Cout, int N, CnetGeo* out) {
    if (!(hidden == 64 || hidden == 128 || hidden == 256 || hidden == 512)) return false;
    if (!pow2(W) || !pow2(H) || W < 4 || W > 128) return false;
    const int HW = H * W;
    if (HW < 64) return false;
    if (Cin < 1 || Cout < 1 || Cout > 56) return false;
    CnetGeo g{};
    g.HW = HW;
    g.wshift = __builtin_ctz(W);
    if (HW >= CN_PX) { g.NI = 1; g.R = CN_PX / W; g.lsub = 7; }
    else { g.NI = CN_PX / HW; g.R = H; g.lsub = __builtin_ctz(HW); }
    if (g.NI > 2) return false;
    g.WP = W + 2;
    g.Wpx = (g.R + 2) * g.WP;
    g.nchunk = (Cin + 7) / 8;
    g.G = cnet_g0(Cin);
    g.steps0 = g.G / 2;
    g.Mpad4 = cnet_mpad4(Cout);
    g.Mrow = cnet_trow(9 * Cout);
    g.NRT4 = g.Mpad4 / 32;
    if (g.NRT4 > 16) return false;
    g.KS = g.NRT4 <= 2 ? 4 : (g.NRT4 <= 4 ? 2 : 1);
    g.npass = (size_t)CN_PX * g.Mrow * sizeof(float) > (size_t)CN_HBUF ? 2 : 1;
    if (g.npass == 2 && (g.NI != 2 || (size_t)(CN_PX / 2) * g.Mrow * sizeof(float) > (size_t)CN_HBUF)) return false;
    g.winplane = g.nchunk * g.NI * g.Wpx * 8;
    if ((size_t)CN_HBUF + (size_t)2 * g.winplane * sizeof(_Float16) > 160 * 1024) return false;
    g.tiles = N > 0 ? (int)(((long)N * HW + CN_PX - 1) / CN_PX) : 0;
    if (out) *out = g;
    return true;
}

bool cnet_supported(int Cin, int H, int W, int hidden, int Cout) { return cnet_geo(Cin, H, W, hidden, Cout, 0, nullptr); }

size_t cnet_scratch_floats(int N, int H, int W, int Cout) {
    const long tiles = ((long)N * H * W + CN_PX - 1) / CN_PX;
    return (size_t)CN_MAXMS * ((size_t)N * Cout * H * W + (size_t)2 * tiles * Cout * W);
}

size_t cnet_scratch_floats_per_sample(int H, int W, int Cout) {
    const long tiles = ((long)H * W + CN_PX - 1) / CN_PX;
    return (size_t)CN_MAXMS * ((size_t)Cout * H * W + (size_t)2 * tiles * Cout * W);
}

template <int HID, int MS, int UPW>
static int launch_cnet_inst(const CnetArgs& a, const CnetGeo& g, hipStream_t s) {
    const size_t lds = (size_t)CN_HBUF + (size_t)2 * g.winplane * sizeof(_Float16);
    (void)hipFuncSetAttribute((const void*)k_cnet<HID, MS, UPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_cnet<HID, MS, UPW>), dim3(g.tiles, MS), dim3(512), lds, s, a, g);
    GH_LAUNCH_CHECK("k_cnet");
    return GLOWHIP_OK;
}

int launch_cnet(const CnetArgs& a, hipStream_t s) {
    CnetGeo g;
    GH_REQUIRE(cnet_geo(a.Cin, a.H, a.W, a.hidden, a.Cout, a.N, &g), "cnet: unsupported shape");
    GH_REQUIRE(a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV || a.mode == TAIL_ADD_FWD || a.mode == TAIL_ADD_REV,
               "cnet: coupling modes only");
    if (a.N == 0) return GLOWHIP_OK;
    // rows of h2 per workgroup: split them over MS workgroups when the pixel tiles alone leave most CUs idle
    int ms = 1;
    const int ms_max = std::min(CN_MAXMS, a.hidden / 64);
    while (ms < ms_max && g.tiles * ms < 160) ms *= 2;
    if (g_cnet_ms) ms = std::min(g_cnet_ms, ms_max);
    if (a.y_sh) ms = 1;
    // T units per wave: (row tiles of T) x (k parts) over 8 waves.  Two units per wave next to the 128 accumulator registers of
    // a 512-row h2 block would spill: that combination runs with the rows split in two
    const int upw = g.NRT4 * g.KS > 8 ? 2 : 1;
    if (upw == 2 && a.hidden == 512 && ms == 1 && !a.y_sh) ms = 2;
    int rc = GLOWHIP_EINVAL;
#define GH_CN(hid, m, u) if (a.hidden == hid && ms == m && upw == u) rc = launch_cnet_inst<hid, m, u>(a, g, s);
    GH_CN(512, 1, 1) GH_CN(512, 2, 1) GH_CN(512, 4, 1) GH_CN(256, 1, 1) GH_CN(256, 2, 1) GH_CN(256, 4, 1) GH_CN(128, 1, 1)
    GH_CN(128, 2, 1) GH_CN(64, 1, 1)
    GH_CN(512, 2, 2) GH_CN(512, 4, 2) GH_CN(256, 1, 2) GH_CN(256, 2, 2) GH_CN(256, 4, 2) GH_CN(128, 1, 2) GH_CN(128, 2, 2)
    GH_CN(64, 1, 2)
#undef GH_CN
    GH_TRY(rc);
    if (a.y_sh) return GLOWHIP_OK;
    CfinArgs f{a, ms, g.tiles, g.R, g.NI, g.wshift, g.HW};
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    const int C = 2 * (paired ? a.Cout / 2 : a.Cout);
    GH_REQUIRE(a.mix_C == 0 || a.mix_C == C, "cnet: mixer channel count %d != %d", a.mix_C, C);
    const size_t flds = ((size_t)C * 64 + (a.mix_C && a.mix_matrix ? (size_t)C * C : 0)) * sizeof(float);
    GH_REQUIRE(flds <= 64 * 1024, "cnet: finishing kernel LDS");
    if (flds > 32 * 1024) (void)hipFuncSetAttribute((const void*)k_cfinish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds);
    hipLaunchKernelGGL(k_cfinish, dim3((unsigned)((long)a.N * g.HW / 64)), dim3(256), flds, s, f);
    GH_LAUNCH_CHECK("k_cfinish");
    return GLOWHIP_OK;
}

}  // namespace glowhip
