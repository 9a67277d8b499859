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
//   P3   f.4 with the filter taps moved to the OUTPUT side: T[tap*Cout + co][px] = sum_k W4[co][k][tap] h2[k][px],
//        h2 passed to the B side through the same LDS buffer (256 channels at a time), T accumulated in registers
//   P4   T -> LDS, shifted 9-tap sums.  Rows of the tile's own pixels go to `hpart`; what the tile's first / last image row
//        contributes to the rows just outside the tile goes to `hup` / `hdn` (no halo recompute, no atomics: deterministic).
// MS > 1 splits the h2 rows (= f.4's reduction axis) over MS workgroups per tile, each recomputing h1 -- for the levels whose
// pixel count alone cannot fill 256 CUs.  The finishing kernel sums the MS partials and the neighbours' halo rows.
#include "sh.h"
#include <algorithm>
#include <type_traits>

#include "conv_mfma.h"
#include "cnet_fin.h"
#include "cnet_geo.h"

GH_STAMPS_DEFINE(cnet)
GH_WGTIMES_DEFINE(cnet)

#ifndef CN_SB_P1
#define CN_SB_P1 0
#endif
#ifndef CN_SB_P3
#define CN_SB_P3 0
#endif

namespace glowhip {

// (FinSrc, fin_src, fin_gather_t: cnet_fin.h -- shared with the backward's k_chanmix_bwd)

// PRE: the launch may finish the previous step while it builds its window (a.pre_on); without it none of that code is compiled in
// NG: f.4 output-channel groups (compile time: with one group nothing of the group loop survives -- as a run-time count it kept
// the h2 accumulators alive through P4 and cost the level-1 instance 88 bytes of spills)
// MODE 1 (TAPE), the training forward (plan_train.hip): h1 and h2 also go to memory as FP16, pixel-tile-major
// [pixel / 32][hidden][pixel % 32] over the batch's pixels (sh.h "T32": the 8 rows x 32 pixels a wave stores are one contiguous
// 512-byte block, and a weight-gradient GEMM's 128-row x 32-pixel operand panel is one contiguous block instead of 128 runs
// 2 - 4 KB apart) -- what the
// weight-gradient GEMMs read as their B operand (wgrad_mfma.hip BH: 2 bytes per value on the tape and in their loaders; the
// gradient operand keeps both planes) -- stored from the
// epilogues (a lane holds one pixel x 4 consecutive channels: a wave store is two 64-byte runs of one channel each), and their
// signs as bit masks (a lane's 16 accumulator registers of a 32-row tile = one 16-bit word, register k in bit 15 - k;
// mask[(row tile * P + pixel) * 2 + kl]).
// MODE 2 (BWD), the input-gradient chain of the same network (it has the same shape: 3x3 Cout -> hidden with f.4's transposed
// weights, 1x1 hidden -> hidden with f.2's, 3x3 hidden -> C/2 with f.0's): x = d L / d(f.4 output), the "activation" of the
// first two layers is  g_u = g_h * (h > 0) * exp(3 logs)  (the scale folded into the weight image's rows, the mask from the
// tape's bit masks), and g_u2 / g_u0 go to memory as fp32, pixel-tile-major like the tape, for the weight-gradient GEMMs.  Needs HW % 32 == 0.
template <int HID, int MS, int UPW, int PXT, bool PRE, int NG = 1, int MODE = 0>
__global__ void __launch_bounds__(512) k_cnet(CnetArgs a, CnetGeo g) {
    constexpr bool TAPE = MODE == 1, BWD = MODE == 2, STORE = MODE != 0;
#ifdef CN_DBG_NOF32
    constexpr bool F32ST = false;       // (timing experiments only: the fp32 tensors are not stored)
#else
    constexpr bool F32ST = STORE;
#endif
#ifdef CN_DBG_NOMASK
    constexpr bool MSKST = false;       // (timing experiments only: the sign words are not stored)
#else
    constexpr bool MSKST = TAPE;
#endif
    const long P_all = (long)a.N * g.HW;          // pixels of the batch (row stride of the bit masks)
    const bool pre_on = PRE && a.pre_on;
    constexpr int NPT = PXT / 32;                    // pixel tiles of the workgroup: 4 (128 pixels) or 2 (64 pixels)
    constexpr int ADEPTH = (HID / MS / 32) * NPT / 8 == 8 ? 1 : 2;       // prefetch distance of P2's A fragments (k-steps)
    constexpr bool MIXSPLIT = PXT == 64 && NG == 1;  // v_fma_mix_f32 in the split epilogues where registers are to spare (sh.h)
    constexpr int CAP = CN_HBUF / (PXT * 4);         // activation channels the LDS buffer holds as (hi, lo) halves: 256 / 512
    constexpr int NH = HID > CAP ? HID / CAP : 1;    // passes over h1
    constexpr int HK = HID / NH;                     // channels per pass
    constexpr int NCH = HK / 8;                      // 8-channel chunks per pass
    constexpr int MR = HID / MS;                     // h2 rows of this workgroup
    constexpr int TP1 = (HK / 32) * NPT / 8;         // P1 tiles per wave and pass
    constexpr int TP2 = (MR / 32) * NPT / 8;         // P2 tiles per wave
    static_assert(TP1 >= 1 && TP2 >= 1, "hidden / MS too small for 8 waves");
    constexpr int PT1 = TP1 < NPT ? TP1 : NPT, RT1 = TP1 / PT1;   // a wave's h1 block: RT1 row tiles x PT1 pixel tiles
    constexpr int PT2 = TP2 < NPT ? TP2 : NPT, RT2 = TP2 / PT2;   // a wave's h2 block
    constexpr int P1SUB = (TP2 == 8 && PT1 == 4) ? 2 : 1;   // P1 in two pixel sub-passes while 128 accumulator registers are live
    constexpr int PTS = PT1 / P1SUB;                 // pixel tiles per P1 sub-pass
    constexpr int NL = MR > CAP ? MR / CAP : 1;      // loads of h2 into the LDS buffer for P3
    constexpr int LK = MR / NL;                      // channels per load
    constexpr int LCH = LK / 8;                      // chunks per load
    constexpr int NS = HK / 16;                      // k-steps of P2 per pass
    constexpr int RTU = 4 / NPT;                     // row tiles of a T unit (a unit = 4 MFMA tiles: RTU row tiles x all pixel tiles)
    static_assert(NS % 3 == 1 || NH == 1, "the set rotation of P2 must end on set 0 for the second pass to restart there");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_c[];
    _Float16* hbuf = smem_c;
    _Float16* win = smem_c + CN_HBUF / 2;
    // small tables behind the window: row scale / bias of f.0 and f.2 (this workgroup's rows), row scale of the f.4 rows -- read
    // from LDS in the epilogues instead of from global memory
    float* t_rs0 = reinterpret_cast<float*>(win + 2 * g.winplane);
    float* t_b0 = t_rs0 + HID;
    float* t_rs2 = t_b0 + HID;
    float* t_b2 = t_rs2 + MR;
    float* t_rs4 = t_b2 + MR;
    long long* ld_slot = reinterpret_cast<long long*>((reinterpret_cast<size_t>(t_rs4 + g.Mpad4) + 7) & ~(size_t)7);   // [8 waves][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, ml = lane & 31;
    const int W = a.W, H = a.H, HW = g.HW;
    const int tb = blockIdx.x;
    const int ms_row0 = blockIdx.y * MR;
    const long gp0 = (long)tb * PXT;
    // tile origin: one image (NI = 1: R rows from y0) or NI whole images
    const long n0 = g.NI == 1 ? gp0 / HW : (long)tb * g.NI;
    const int y0 = g.NI == 1 ? (int)((gp0 - n0 * HW) >> g.wshift) : 0;
    const int submask = (1 << g.lsub) - 1;

    // ---- A-operand bases
    const _Float16* W0 = (const _Float16*)a.w0;
    const long w0_plane = (long)g.G * HID * 8;
    const float* rs0 = (const float*)((const char*)a.w0 + sh2_rowscale_off(g.G * 8, HID));
    const _Float16* W2 = (const _Float16*)a.w2;
    constexpr long w2_plane = (long)HID * HID;
    const float* rs2 = (const float*)((const char*)a.w2 + sh2_rowscale_off(HID, HID));
    const _Float16* W4 = (const _Float16*)a.w4;
    const long w4_plane = (long)HID * g.Mpad4;
    const float* rs4 = (const float*)((const char*)a.w4 + sh2_rowscale_off(HID, g.Mpad4));

    // ---- wave's h2 block: row tiles [rt2, rt2 + RT2), pixel tiles [pt2, pt2 + PT2)
    const int rt2 = (wid * TP2) / NPT, pt2 = (wid * TP2) % NPT;
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
    // ---- wave's h1 block per pass: row tiles [rt1, rt1 + RT1), pixel tiles [pt1, pt1 + PT1)
    const int rt1 = (wid * TP1) / NPT, pt1 = (wid * TP1) % NPT;
    const _Float16* a1p = W0 + ((long)kl * HID + rt1 * 32 + ml) * 8;               // + pass * HK*8 ; + st * 2*HID*8 ; lo: + w0_plane
    h8 A1[3][2 * RT1];     // [set][i] hi, [set][RT1 + i] lo
    auto loadA1 = [&](int hh, int st, h8 (&dst)[2 * RT1]) {
        const _Float16* p = a1p + (long)hh * (HK * 8) + (long)st * (2 * HID * 8);
#pragma unroll
        for (int i = 0; i < RT1; ++i) {
            dst[i] = *reinterpret_cast<const h8*>(p + i * 256);
            dst[RT1 + i] = *reinterpret_cast<const h8*>(p + i * 256 + w0_plane);
        }
    };
    GH_STAMP(0);
    GH_WG_BEGIN();
    GH_STAMP_VAL(63, __builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8]
    // requested before the window is built (their L2 round trips overlap P0): the first two A sets of P1 and of P2
    loadA1(0, 0, A1[0]);
    loadA1(0, g.steps0 > 1 ? 1 : 0, A1[1]);
    constexpr bool A2_LATE = TP2 == 8;    // 128 accumulator registers live across P1: P2's first A sets are requested after P1
    // (otherwise right after the window is built -- not here: 256 workgroups asking for 64 KB of f.2 weights each in the same
    // instant as their windows stretched the kernel's first trip to memory from 4.8 k to 6.7 k cycles at level 2, and P2 is
    // a whole P1 away)

    // ---- P0: tables, then the window -> (hi, lo) halves in LDS; slot e = (chunk, sub-tile, window pixel), 8 channels each
    const int nwin = g.NI * g.Wpx;
    const int nslots = g.nchunk * nwin;
    // window slot e -> (inside the image?, address of its first channel); the address is always valid (clamped)
    int slot_r, slot_c, slot_sub;      // window row / column / sub-image of the slot slot_src was last asked about
    auto slot_src = [&](int e, bool& in, int& ch) {
        // (three divisions by run-time constants, on the way to the kernel's FIRST loads: one v_mul_hi each instead of ~25 instructions)
        ch = (int)__umulhi((unsigned)e, g.m_nwin);
        const int rem = e - ch * nwin;
        const int sub = (int)__umulhi((unsigned)rem, g.m_Wpx), wp = rem - sub * g.Wpx;
        const int r = (int)__umulhi((unsigned)wp, g.m_WP), c = wp - r * g.WP;
        slot_r = r; slot_c = c; slot_sub = sub;
        const int yy = y0 - 1 + r, xx = c - 1;
        const long n = n0 + sub;
        in = yy >= 0 && yy < H && xx >= 0 && xx < W && n < a.N;
        return a.x + (n < a.N ? n : (long)a.N - 1) * a.x_bs + min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1);
    };
    // With `pre`: the window pixels that lie inside an image, as a compact list k = (sub-image * nrows + row) * W + x over the window
    // rows [rlo, rlo + nrows); KP = that count rounded up to whole waves.
    const int rlo = y0 == 0 ? 1 : 0;
    const int nrows = min(g.R + 2, H - y0 + 1) - rlo;
    const int nin = g.NI * nrows * W;
    const int KP = (nin + 63) & ~63;
    const int per_sub = nrows << g.wshift;
    // pixel k of the list -> image n, pixel index p, own pixel of this tile?
    auto pix = [&](int k, long& n, int& p, bool& own, int& sub) {
        sub = k >= per_sub ? 1 : 0;
        const int kk = k - sub * per_sub;
        const int r = rlo + (kk >> g.wshift);                     // window row
        n = n0 + sub;
        p = (y0 - 1 + r) * W + (kk & (W - 1));
        own = r >= 1 && r <= g.R;
        return k < nin && n < a.N;
    };
    // ---- the first round of global loads is REQUESTED before the tables are copied (one trip to memory for both): the window
    // values, or -- chained launch -- everything the finishing of the previous step reads
    constexpr int PU = 3;      // elements per thread and round of the finishing
    const FinSrc pf = pre_on ? fin_src(a.pre, a.N, H, W, HW, g.wshift) : FinSrc{};
    const int pCh = pf.paired ? pf.Cout / 2 : pf.Cout, pC = 2 * pCh;
    const int ptotal = pCh * KP;
    float pse[PU], pso[PU], pzin[PU], pz1[PU], pkb[PU][4], pkm[PU][4];      // pkb: f.4 bias / scale (even, odd); pkm: mixer bias / scale
    int pcc[PU], pkq[PU];
    auto pre_gather = [&](int e0, auto halo_c) {
        constexpr bool HALO = decltype(halo_c)::value;
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int e = min(e0 + 512 * u, ptotal - 1);
            const int c = __builtin_amdgcn_readfirstlane(e / KP);           // (a wave's 64 elements share the channel: 64 | KP)
            const int k = e - c * KP;
            long n; int p, sub; bool own;
            const bool ok = pix(k, n, p, own, sub);
            const long nc = ok ? n : n0;
            const int pc = ok ? p : (y0 - 1 + rlo) * W;                     // (clamped: a pixel of the list's first row)
            const float* zp = a.pre.z + nc * a.pre.z_bs;
            pz1[u] = zp[(long)c * HW + pc];
            pzin[u] = zp[(long)(pCh + c) * HW + pc];
            fin_gather_t<MS, HALO>(pf, nc, c, pc, pse[u], pso[u]);       // (pre.MS == MS: checked by the launcher)
            pcc[u] = c; pkq[u] = k;
            const int ce = pf.paired ? 2 * c : c, co = ce + (pf.paired ? 1 : 0);
            pkb[u][0] = pf.bias[ce]; pkb[u][1] = pf.scale[ce]; pkb[u][2] = pf.bias[co]; pkb[u][3] = pf.scale[co];
            const bool an = !a.pre_mix.reverse && a.pre_mix.C;
            pkm[u][0] = an ? a.pre_mix.bias[c] : 0.f; pkm[u][1] = an ? a.pre_mix.scale[c] : 1.f;
            pkm[u][2] = an ? a.pre_mix.bias[pCh + c] : 0.f; pkm[u][3] = an ? a.pre_mix.scale[pCh + c] : 1.f;
        }
    };
    float v0[8];
    if (!pre_on) {     // (chained launches build the window from the state they finish themselves)
        bool in; int ch;
        const float* xin = slot_src(min(tid, nslots - 1), in, ch);
#pragma unroll
        for (int q = 0; q < 8; ++q) v0[q] = xin[(long)min(ch * 8 + q, a.Cin - 1) * HW];
    } else if (pf.halos) {
        pre_gather(tid, std::true_type{});
    } else {
        pre_gather(tid, std::false_type{});
    }
    // The activations travel NEGATED from the first epilogue on (sh.h nrelu_bits): h1 and h2 sit in LDS as -h1, -h2, the h2 and T
    // accumulators hold the negated sums.  So the tables carry the signs: f.0 (-rs0, -b0) turns the true accumulator into -t;
    // f.2 (rs2, -b2) turns the negated accumulator into -t; f.4's row scale -rs4 turns the negated T back.  canon_nan: a NaN
    // from memory gets the sign bit the hardware's own NaNs have.
    // ALL table values are requested here, in the same round as the window values and the first A sets, from clamped addresses, and
    // stored afterwards: as three copy loops every iteration was a round trip of its own (load, s_waitcnt vmcnt(0) -- which also
    // waited for everything requested before -- store): four to five serialised trips to L2 in front of the window, 4.8 k - 6.7 k
    // cycles by the stamps.
    constexpr int N0 = (2 * HID + 511) / 512, N2 = (MR + 511) / 512;
    float tv0[N0], tv2[N2], tvb[N2], tv4;
#pragma unroll
    for (int i = 0; i < N0; ++i) tv0[i] = rs0[min(tid + 512 * i, 2 * HID - 1)];                  // rs0 | b0 are adjacent in the image
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const int e = min(tid + 512 * i, MR - 1);
        tv2[i] = rs2[ms_row0 + e]; tvb[i] = rs2[HID + ms_row0 + e];
    }
    tv4 = rs4[min(tid, g.Mpad4 - 1)];             // (Mpad4 <= 512; group 0: the others are loaded in the group loop)
    __builtin_amdgcn_sched_barrier(0);            // (hipcc sinks the last request behind the first store otherwise: one more trip)
#pragma unroll
    for (int i = 0; i < N0; ++i)
        if (tid + 512 * i < 2 * HID) t_rs0[tid + 512 * i] = canon_nan(-tv0[i]);
#pragma unroll
    for (int i = 0; i < N2; ++i)
        if (tid + 512 * i < MR) { t_rs2[tid + 512 * i] = canon_nan(tv2[i]); t_b2[tid + 512 * i] = canon_nan(-tvb[i]); }
    if (tid < g.Mpad4) t_rs4[tid] = canon_nan(-tv4);
    float* nz = nullptr;     // with `pre`: z1 of the freshly finished state, fp32 [Cin][KP]
    if (pre_on) {
        // ---- finish the PREVIOUS step on the window pixels: coupling (+ log-det for the tile's own pixels), then the channel mixer;
        // the result of the own pixels goes to pre_z_new, z1 of all of them to `nz` (hbuf is free at this point).  Same arithmetic
        // as k_cfinish, bit for bit.  Built for latency: ONE round of global loads (requested above), the mixer's matrix staged in
        // LDS meanwhile and read as wave-uniform 16-byte broadcasts, log-det terms summed per wave (fixed point: any grouping
        // gives the same bits).
        const FinSrc& f = pf;
        const int Ch = pCh, C = pC;
        float* pv = reinterpret_cast<float*>(hbuf);          // [C][KP] staged values
        nz = pv + C * KP;
        const int CP = (C + 3) & ~3;                          // matrix row stride (16-byte aligned rows)
        float* pm = nz + a.Cin * KP;                          // [C][CP]
        if (a.pre_mix.matrix)
            for (int e = tid; e < C * C; e += 512) { const int o = e / C; pm[o * CP + (e - o * C)] = a.pre_mix.matrix[e]; }
        GH_STAMP(22);
        long long ldq[2] = {0, 0};
        for (int e0 = tid; e0 < ptotal; e0 += 512 * PU) {
            if (e0 != tid) {                                  // (a second round only for shapes beyond the product's)
                if (f.halos) pre_gather(e0, std::true_type{}); else pre_gather(e0, std::false_type{});
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (e0 + 512 * u >= ptotal) continue;
                const int c = pcc[u], k = pkq[u];
                long n; int p, sub; bool own;
                const bool ok = pix(k, n, p, own, sub);
                float v1 = 0.f, v2 = 0.f;
                if (ok) {
                    long long lq = 0;
                    float bad = 0.f;
                    const float zres = fin_apply_k(f, pse[u], pso[u], pzin[u], pkb[u][0], pkb[u][1], pkb[u][2], pkb[u][3], lq, bad);
                    if (own && blockIdx.y == 0) {            // own pixel (not a halo row), once per tile: counts for the log-det
                        if (bad != 0.f) fix_flag_nonfinite(a.acc, n, a.N, bad);
                        ldq[sub] += lq;
                    }
                    if (!a.pre_mix.reverse && a.pre_mix.C) {             // ActNorm of the mixer on both halves, staged for its matrix / gather
                        v1 = (pz1[u] + pkm[u][0]) * pkm[u][1];
                        v2 = (zres + pkm[u][2]) * pkm[u][3];
                    } else { v1 = pz1[u]; v2 = zres; }
                }
                pv[c * KP + k] = v1;
                pv[(Ch + c) * KP + k] = v2;
            }
        }
        if (f.paired && blockIdx.y == 0) {                    // per-wave sums of the fixed-point terms, one atomic per wave and image
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                if (sb >= g.NI) continue;
                long long v = ldq[sb];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                if (lane == 0) ld_slot[wid * 2 + sb] = v;     // added to the sample's accumulator at the END of the kernel: an
                                                              // atomic in flight here would sit in front of every later wait
            }
        }
        GH_STAMP(23);
        __syncthreads();
        GH_STAMP(24);
        // mixer: item = (group of OB output channels, 64 pixels); the matrix rows of a wave's output channels are wave-uniform
        constexpr int OB = 6;
        const int npb = KP >> 6, nog = (C + OB - 1) / OB;
        for (int it = wid; it < nog * npb; it += 8) {
            const int og = it / npb, pb = it - og * npb;
            const int k = pb * 64 + lane;
            float r[OB], osc[OB], obi[OB];
            const bool rev = a.pre_mix.reverse && a.pre_mix.C;
#pragma unroll
            for (int j = 0; j < OB; ++j) {        // inverse ActNorm of the outputs (reverse flow): requested before the products
                const int o = min(og * OB + j, C - 1);
                osc[j] = rev ? a.pre_mix.scale[o] : 1.f;
                obi[j] = rev ? a.pre_mix.bias[o] : 0.f;
            }
            if (a.pre_mix.matrix) {   // same operation order as k_chanmix / k_cfinish: r = fma(m[o][i], v[i], r), i ascending
#pragma unroll
                for (int j = 0; j < OB; ++j) r[j] = 0.f;
                const float* mrow[OB];
#pragma unroll
                for (int j = 0; j < OB; ++j) mrow[j] = pm + min(og * OB + j, C - 1) * CP;
                int i = 0;
                for (; i + 4 <= C; i += 4) {
                    float vi[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) vi[t] = pv[(i + t) * KP + k];
#pragma unroll
                    for (int j = 0; j < OB; ++j) {
                        const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(mrow[j] + i);
#pragma unroll
                        for (int t = 0; t < 4; ++t) r[j] = fmaf(m4[t], vi[t], r[j]);
                    }
                }
                for (; i < C; ++i) {
                    const float vi = pv[i * KP + k];
#pragma unroll
                    for (int j = 0; j < OB; ++j) r[j] = fmaf(mrow[j][i], vi, r[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < OB; ++j) {
                    const int o = min(og * OB + j, C - 1);
                    r[j] = pv[(a.pre_mix.gather ? a.pre_mix.gather[o] : o) * KP + k];
                }
            }
            long n; int p, sub; bool own;
            const bool ok = pix(k, n, p, own, sub);
            const bool wr = ok && own && blockIdx.y == 0;
            float* zo = a.pre_z_new + (wr ? n : n0) * a.pre_z_new_bs + (wr ? p : 0);
#pragma unroll
            for (int j = 0; j < OB; ++j) {
                const int o = og * OB + j;
                if (rev) r[j] = r[j] * osc[j] - obi[j];
                if (o < C && wr) zo[(long)o * HW] = r[j];
            }
#pragma unroll
            for (int j = 0; j < OB; ++j) {
                const int o = og * OB + j;
                if (o < a.Cin && o < C) nz[o * KP + k] = ok ? r[j] : 0.f;
            }
        }
        GH_STAMP(25);
        __syncthreads();
    }
    GH_STAMP(26);
    for (int e = tid; e < nslots; e += 512) {
        bool in; int ch;
        const float* xin = slot_src(e, in, ch);
        float v[8];
        if (nz) {
            const int k = in ? ((slot_sub * nrows + slot_r - rlo) << g.wshift) + slot_c - 1 : 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = nz[min(ch * 8 + q, a.Cin - 1) * KP + k];
        } else if (e == tid) {      // (nz == nullptr <=> !pre_on: v0 was loaded)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = v0[q];
        } else {
            // every load is issued unconditionally from a clamped (valid) address and masked afterwards: eight independent
            // loads in flight per slot instead of eight round trips behind one another
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = xin[(long)min(ch * 8 + q, a.Cin - 1) * HW];
        }
        h8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float vv = (in && ch * 8 + q < a.Cin) ? canon_nan(v[q] * (STORE ? a.in_scale : SH2_ACT_SCALE)) : 0.f;
            _Float16 x0, x1;
            sh2_split(vv, x0, x1);
            hi[q] = x0; lo[q] = x1;
        }
        *reinterpret_cast<h8*>(win + (long)e * 8) = hi;
        *reinterpret_cast<h8*>(win + g.winplane + (long)e * 8) = lo;
    }
    __syncthreads();
    GH_STAMP(1);
    if (!A2_LATE) {
        loadA2(0, A2[0]);
        loadA2(1, A2[1]);
    }

    // window offset (halfs) of tile pixel q for tap (0,0), chunk 0
    auto pix_base = [&](int q) {
        const int sub = q >> g.lsub, qq = q & submask;
        return (sub * g.Wpx + (qq >> g.wshift) * g.WP + (qq & (W - 1))) * 8;
    };

    f32x16_t acc2[RT2][PT2];
    // P1 of one channel pass; PTSV = pixel tiles per sub-pass (fewer while the h2 accumulators are live)
    // syncc: the barrier "every wave is done reading the previous pass of h1" sits INSIDE this pass, between the MFMAs of its first
    // pixel sub-pass (which only read the window) and the first store into the h1 buffer -- a wave that is done with P2 early
    // (the older wave of each SIMD, ~8 k cycles) issues those MFMAs into the gaps of its partner's P2 instead of waiting
    auto p1_pass = [&](int hh, auto ptsc, auto syncc) {
        constexpr int PTSV = decltype(ptsc)::value;
        constexpr bool SYNC = decltype(syncc)::value;
        constexpr int SUBV = PT1 / PTSV;
        // ---- P1: h1 channels [hh*HK, hh*HK + HK) of all pixels -> hbuf.  Its first two A sets are already in flight.
#pragma unroll 1
        for (int sp = 0; sp < SUBV; ++sp) {
            int pb[PTSV];
#pragma unroll
            for (int j = 0; j < PTSV; ++j) pb[j] = pix_base((pt1 + sp * PTSV + j) * 32 + ml);
            f32x16_t acc1[RT1][PTSV];
#pragma unroll
            for (int i = 0; i < RT1; ++i)
#pragma unroll
                for (int j = 0; j < PTSV; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
            unsigned mw1[BWD ? RT1 : 1][BWD ? PTSV : 1];     // BWD: sign words of h2 for this block, requested before its MFMAs
            if (BWD) {
#pragma unroll
                for (int i = 0; i < RT1; ++i)
#pragma unroll
                    for (int j = 0; j < PTSV; ++j) {
                        const long px = min(gp0 + (pt1 + sp * PTSV + j) * 32 + ml, P_all - 1);
                        mw1[BWD ? i : 0][BWD ? j : 0] = a.mask2[((long)(hh * (HK / 32) + rt1 + i) * P_all + px) * 2 + kl];
                    }
            }
            // k groups of f.0 are ordered (8-channel chunk, tap), tap fastest: group gk -> chunk gk / 9, tap gk % 9 (divisions by
            // constants); groups past 9 * nchunk carry zero weights and read window offset 0.  With at most two pixel tiles the B
            // fragments of step st + 1 -- tap-shifted window addresses -- are requested while the MFMAs of step st issue.  The A
            // sets rotate by NAME (loop unrolled by three): a rotation by register copies makes the compiler wait for the newest
            // global load at the end of every step (vmcnt(0)).
            int gk = kl;
            auto cur_goff = [&]() {
                const int ch = gk / 9, tap = gk - ch * 9;
                const int dy = tap / 3, dx = tap - dy * 3;
                return ch < g.nchunk ? (ch * g.NI * g.Wpx + dy * g.WP + dx) * 8 : 0;
            };
            constexpr bool BPRE = PTSV <= 2;    // B fragments one step ahead (register budget: only with at most two tiles)
            h8 Bc[2 * PTSV], Bn[BPRE ? 2 * PTSV : 1];     // [j] hi, [PTSV + j] lo
            auto loadB = [&](h8* dst) {
                const int goff = cur_goff();
#pragma unroll
                for (int j = 0; j < PTSV; ++j) {
                    dst[j] = *reinterpret_cast<const h8*>(win + goff + pb[j]);
                    dst[PTSV + j] = *reinterpret_cast<const h8*>(win + g.winplane + goff + pb[j]);
                }
            };
            if (BPRE) loadB(Bc);
            // ld: std::true_type in the steady loop (the load is always a real one there: no branch, counted waits), a run-time flag in
            // the tail -- the last two k-steps have nothing left to request.  (They used to re-request the last set, clamped: the
            // loads were still in flight when the phase ended and whatever reused their registers next waited a full round trip
            // for data nobody reads: stamps showed 1.8 k - 4.4 k cycles of it after P3.)
            auto step1 = [&](int st, const h8 (&use)[2 * RT1], h8 (&fill)[2 * RT1], auto ld) {
                if constexpr (std::is_same_v<decltype(ld), std::true_type>) loadA1(hh, st + 2, fill);
                else if (ld) loadA1(hh, st + 2, fill);
#if CN_SB_P1
                __builtin_amdgcn_sched_barrier(0);               // (as in P2: keep the request here)
#endif
                if (BPRE) { gk += 2; loadB(Bn); } else { loadB(Bc); gk += 2; }
                // three sweeps over the tiles: consecutive MFMAs never share an accumulator
#pragma unroll
                for (int i = 0; i < RT1; ++i)
#pragma unroll
                    for (int j = 0; j < PTSV; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use[i], Bc[j], acc1[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT1; ++i)
#pragma unroll
                    for (int j = 0; j < PTSV; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use[i], Bc[PTSV + j], acc1[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT1; ++i)
#pragma unroll
                    for (int j = 0; j < PTSV; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use[RT1 + i], Bc[j], acc1[i][j], 0, 0, 0);
                if (BPRE) {
#pragma unroll
                    for (int j = 0; j < 2 * PTSV; ++j) Bc[j] = Bn[j];
                }
            };
            int st = 0;
#pragma unroll 1
            for (; st + 5 <= g.steps0; st += 3) {                // every request of these triples is a real one
                step1(st, A1[0], A1[2], std::true_type{});
                step1(st + 1, A1[1], A1[0], std::true_type{});
                step1(st + 2, A1[2], A1[1], std::true_type{});
            }
            // the two to four k-steps left over (the reduction is padded to whole k-steps only, not to whole triples: level 1
            // has 5 k-steps for its 54 real k values, a sixth would be a sixth of the phase's MFMAs for nothing)
            if (st < g.steps0) step1(st, A1[0], A1[2], st + 2 < g.steps0);
            if (st + 1 < g.steps0) step1(st + 1, A1[1], A1[0], st + 3 < g.steps0);
            if (st + 2 < g.steps0) step1(st + 2, A1[2], A1[1], st + 4 < g.steps0);
            if (st + 3 < g.steps0) step1(st + 3, A1[0], A1[2], false);
            // A sets of the next P1 pass (next pixel sub-pass of this channel pass, or the next channel pass): in flight during
            // the epilogue, the barrier and P2
            {
                const int nh = sp + 1 < SUBV ? hh : hh + 1;
                if (nh < NH) {
                    loadA1(nh, 0, A1[0]);
                    loadA1(nh, g.steps0 > 1 ? 1 : 0, A1[1]);
                }
            }
            if (SYNC && sp == 0) {
                GH_STAMP(4 + 4 * (hh - 1));
                __syncthreads();     // every wave is done reading the previous pass of h1
                GH_STAMP(5 + 4 * (hh - 1));
            }
            // -relu (sh.h nrelu_bits), split, store into the B-operand image: a lane's 4 consecutive channels = 8 bytes per plane
            // one 32 x 32 tile's group gq of four rows; mbw: the tile's sign word (read in BWD, built in TAPE)
            h4 pg_hi, pg_lo;       // the even row group's halves of the (tile, pixel tile) p1_group is walking
            auto p1_group = [&](int i, int j, int gq, unsigned& mbw) {
                const int o = hh * HK + (rt1 + i) * 32 + 8 * gq + 4 * kl;
                const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(t_rs0 + o);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(t_b0 + o);
                const int chunk = (rt1 + i) * 4 + (gq & 2) + kl;      // k-permuted position of rows 8 gq + 4 kl + t (sh.h sh2_kperm_src)
                h4 hi, lo;
                f32x4_t v;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float tt = fmaf(acc1[i][j][4 * gq + t], rs[t], bb[t]);
                    if (BWD) v[t] = (mbw >> (15 - (4 * gq + t))) & 1u ? tt : 0.f;      // (-g_u2, scaled: tables as in the forward)
                    else v[t] = nrelu_bits(tt);
                    // sign word: value k = 4 gq + t ends up in bit 15 - k (one v_alignbit per value: the word shifted left by
                    // one, the sign bit of -h -- set exactly where h > 0 -- shifted in)
                    if (TAPE) mbw = __builtin_amdgcn_alignbit(mbw, __float_as_uint(v[t]), 31);
                }
                sh2_split4<MIXSPLIT>(v, hi, lo);
                if ((gq & 1) == 0) { pg_hi = hi; pg_lo = lo; }
                else {           // (one 16-byte store per pair of row groups, as in the product epilogue below)
                    const h8 hi8 = {pg_hi[0], pg_hi[1], pg_hi[2], pg_hi[3], hi[0], hi[1], hi[2], hi[3]};
                    const h8 lo8 = {pg_lo[0], pg_lo[1], pg_lo[2], pg_lo[3], lo[0], lo[1], lo[2], lo[3]};
                    _Float16* dst = hbuf + ((long)chunk * PXT + (pt1 + sp * PTSV + j) * 32 + ml) * 8;
                    *reinterpret_cast<h8*>(dst) = hi8;
                    *reinterpret_cast<h8*>(dst + (long)NCH * PXT * 8) = lo8;
                }
                if (F32ST) {     // (with a row split every workgroup computes all of these rows: each stores its own rows' share)
                    const long px0 = gp0 + (pt1 + sp * PTSV + j) * 32;
                    if (px0 < P_all && (MS == 1 || (o >= ms_row0 && o < ms_row0 + MR))) {
                        const long tb_off = (px0 >> 5) * (HID * 32);     // pixel-tile-major: [pixel / 32][row][pixel % 32] (sh.h)
                        if (TAPE) {      // h1 as fp16
                            _Float16* tb = reinterpret_cast<_Float16*>(a.tape_h1) + tb_off;
#pragma unroll
                            for (int t = 0; t < 4; ++t) tb[((o + t) << 5) + ml] = (_Float16)(-v[t] * a.out_scale);
                        } else {         // g_u2 as fp32
                            float* tb = a.tape_h1 + tb_off;
#pragma unroll
                            for (int t = 0; t < 4; ++t) __builtin_nontemporal_store(-v[t] * a.out_scale, tb + ((o + t) << 5) + ml);
                        }
                    }
                }
            };
            if (!STORE) {        // product kernel: row groups outermost (the tables are read once per group)
                h4 st_hi[PTSV], st_lo[PTSV];
#pragma unroll
                for (int i = 0; i < RT1; ++i)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int o = hh * HK + (rt1 + i) * 32 + 8 * gq + 4 * kl;
                        const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(t_rs0 + o);
                        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(t_b0 + o);
                        const int chunk = (rt1 + i) * 4 + (gq & 2) + kl;      // k-permuted position (sh.h sh2_kperm_src)
#pragma unroll
                        for (int j = 0; j < PTSV; ++j) {
                            h4 hi, lo;
                            f32x4_t v;
#pragma unroll
                            for (int t = 0; t < 4; ++t) v[t] = nrelu_bits(fmaf(acc1[i][j][4 * gq + t], rs[t], bb[t]));
                            sh2_split4<MIXSPLIT>(v, hi, lo);
                            // (row groups 2 s and 2 s + 1 are the two 8-byte halves of the lane's 16-byte slot: ONE 16-byte store per
                            // plane -- as two 8-byte stores 16 bytes apart from lane to lane they conflict two-way on the LDS banks)
                            if ((gq & 1) == 0) { st_hi[j] = hi; st_lo[j] = lo; continue; }
                            const h8 hi8 = {st_hi[j][0], st_hi[j][1], st_hi[j][2], st_hi[j][3], hi[0], hi[1], hi[2], hi[3]};
                            const h8 lo8 = {st_lo[j][0], st_lo[j][1], st_lo[j][2], st_lo[j][3], lo[0], lo[1], lo[2], lo[3]};
                            _Float16* dst = hbuf + ((long)chunk * PXT + (pt1 + sp * PTSV + j) * 32 + ml) * 8;
                            *reinterpret_cast<h8*>(dst) = hi8;
                            *reinterpret_cast<h8*>(dst + (long)NCH * PXT * 8) = lo8;
                        }
                    }
            } else {             // taping / backward: tile by tile, one sign word live at a time
#pragma unroll
                for (int i = 0; i < RT1; ++i)
#pragma unroll
                    for (int j = 0; j < PTSV; ++j) {
                        unsigned mbw = BWD ? mw1[BWD ? i : 0][BWD ? j : 0] : 0u;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) p1_group(i, j, gq, mbw);
                        if (MSKST) {
                            const int R = hh * (HK / 32) + rt1 + i;
                            const long px0 = gp0 + (pt1 + sp * PTSV + j) * 32;
                            if (px0 < P_all && (MS == 1 || (R * 32 >= ms_row0 && R * 32 < ms_row0 + MR)))
                                a.mask1[((long)R * P_all + px0 + ml) * 2 + kl] = (unsigned short)mbw;
                        }
                    }
            }
        }
    };
    auto p2_pass = [&](int hh) {
        // ---- P2: acc2 += W2'[rows, pass hh] h1[pass hh]; B from LDS, A two k-steps ahead from L2.
        // Operand timing inside a k-step (per-wave stamps: every k-step used to open with the full LDS round trip of its eight B
        // reads in front of its first MFMA -- nothing hides that for the wave that has a SIMD to itself, ~20 % of P2):
        //   * the hi-plane B fragments of step s + 1 are requested before the THIRD sweep of step s (which still multiplies step
        //     s' own hi plane: the next ones go to the other half of a two-deep, named register pair);
        //   * the lo-plane B fragments of step s are requested at its start and first used by the SECOND sweep, 8 MFMAs later.
        // The k-loop is fully unrolled: A set (s % 3) and B pair (s & 1) are register NAMES.
        {
            const int ks0 = hh * NS;
            const _Float16* bp = hbuf + ((long)kl * PXT + pt2 * 32 + ml) * 8;
            h8 bhq[2][PT2];
            auto load_bh = [&](int s, h8 (&dst)[PT2]) {
                const _Float16* bs = bp + (long)s * (2 * PXT * 8);
#pragma unroll
                for (int j = 0; j < PT2; ++j) dst[j] = *reinterpret_cast<const h8*>(bs + j * 256);
            };
            load_bh(0, bhq[0]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // A fragments two k-steps ahead in three named sets -- one k-step ahead in two where the 128 accumulator registers
                // of a 512-row x 128-pixel block leave no room for the third (ADEPTH)
                h8 (&use)[2 * RT2] = A2[ADEPTH == 2 ? s % 3 : s & 1];
                h8 (&fill)[2 * RT2] = A2[ADEPTH == 2 ? (s + 2) % 3 : (s + 1) & 1];
                h8 (&bh)[PT2] = bhq[s & 1];
                if (s + ADEPTH < NS) loadA2(ks0 + s + ADEPTH, fill);      // (s is a compile-time constant: no branch; the last ADEPTH
                                                                          //  k-steps have nothing left to request)
                // the scheduler must not sink these loads towards their use two k-steps later (it does, to save registers, and
                // the wave then waits a full L2 round trip per k-step: measured 57% of the MFMA rate for a wave on its own)
                __builtin_amdgcn_sched_barrier(0);
                const _Float16* bs = bp + (long)s * (2 * PXT * 8);
                h8 bl[PT2];
#pragma unroll
                for (int j = 0; j < PT2; ++j) bl[j] = *reinterpret_cast<const h8*>(bs + j * 256 + (long)NCH * PXT * 8);
#pragma unroll
                for (int i = 0; i < RT2; ++i)
#pragma unroll
                    for (int j = 0; j < PT2; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use[i], bh[j], acc2[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT2; ++i)
#pragma unroll
                    for (int j = 0; j < PT2; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use[i], bl[j], acc2[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < NS) load_bh(s + 1, bhq[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RT2; ++i)
#pragma unroll
                    for (int j = 0; j < PT2; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use[RT2 + i], bh[j], acc2[i][j], 0, 0, 0);
            }
            if (!A2_LATE && hh + 1 < NH) {   // A sets of the next pass' first two k-steps: in flight while P1 rebuilds the LDS buffer
                loadA2((hh + 1) * NS, A2[0]);
                loadA2((hh + 1) * NS + 1, A2[1]);
            }
        }
    };
    // With a 512-row x 128-pixel h2 block (128 accumulator registers) the FIRST channel pass of P1 still runs before those
    // accumulators exist: all four pixel tiles per k-step there (12 MFMAs behind each A set), two at a time afterwards.
    constexpr bool PEEL = TP2 == 8 && PT1 == 4 && NH == 2;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        if (PEEL && hh == 0) p1_pass(hh, std::integral_constant<int, PT1>{}, std::false_type{});
        else if (hh == 0) p1_pass(hh, std::integral_constant<int, PTS>{}, std::false_type{});
        else p1_pass(hh, std::integral_constant<int, PTS>{}, std::true_type{});
        if (hh == 0) {
#pragma unroll
            for (int i = 0; i < RT2; ++i)
#pragma unroll
                for (int j = 0; j < PT2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
        }
        if (A2_LATE) {
            loadA2(hh * NS, A2[0]);
            loadA2(hh * NS + 1, A2[1]);
        }
        GH_STAMP(2 + 4 * hh);
        __syncthreads();
        GH_STAMP(3 + 4 * hh);
        p2_pass(hh);
        if (hh == NH - 1) {  // (between passes that barrier is inside the next P1 pass)
            GH_STAMP(4 + 4 * hh);
            __syncthreads();     // every wave is done reading this pass of h1
            GH_STAMP(5 + 4 * hh);
        }
    }

    // ---- P3 set-up: T units of this wave.  unit = (RTU row tiles ru of T, k part kp) x all pixel tiles = 4 MFMA tiles; wave w
    // takes units w (and w + 8 when UPW = 2).  The first two A sets of the first unit are requested NOW.
    const int nunits = g.NU4 * g.KS;
    const int nsl = (LK / 16) / g.KS;                        // k-steps of one h2 load per k part
    h8 A4[3][2 * RTU];
    const long w4_grp = (long)(sh2_image_bytes(HID, g.Mpad4) / sizeof(_Float16));     // halfs from one group's image to the next
    int grp = 0;                                             // output-channel group of f.4 being computed
    auto a4_base = [&](int unit, int l) {
        const int kp = unit / g.NU4;
        return W4 + (NG > 1 ? grp * w4_grp : 0) + ((long)((ms_row0 + l * LK) / 8 + 2 * kp * nsl + kl) * g.Mpad4 + ml) * 8;
    };
    auto loadA4 = [&](const _Float16* ap, int ru, int st, h8 (&dst)[2 * RTU]) {
#pragma unroll
        for (int i = 0; i < RTU; ++i) {
            const int rt4 = min(ru * RTU + i, g.NRT4 - 1);       // a unit's surplus row tile re-reads the last one (never staged)
            dst[i] = *reinterpret_cast<const h8*>(ap + (long)st * (2 * g.Mpad4 * 8) + rt4 * 256);
            dst[RTU + i] = *reinterpret_cast<const h8*>(ap + (long)st * (2 * g.Mpad4 * 8) + rt4 * 256 + w4_plane);
        }
    };
    if (wid < nunits) {
        const _Float16* ap = a4_base(wid, 0);
        loadA4(ap, wid % g.NU4, 0, A4[0]);
        loadA4(ap, wid % g.NU4, nsl > 1 ? 1 : 0, A4[1]);
    }

    // ---- -h2 = -relu(t2), t2 = true sum * rowscale + bias (times SH2_ACT_SCALE), in place (the accumulators hold the negated sum)
    auto p2_group = [&](int i, int j, int gq, unsigned& mbw) {
        const int o = (rt2 + i) * 32 + 8 * gq + 4 * kl;           // workgroup-local row
        const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(t_rs2 + o);
        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(t_b2 + o);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float tt = fmaf(acc2[i][j][4 * gq + t], rs[t], bb[t]);
            if (BWD) acc2[i][j][4 * gq + t] = (mbw >> (15 - (4 * gq + t))) & 1u ? tt : 0.f;
            else acc2[i][j][4 * gq + t] = nrelu_bits(tt);
            if (TAPE) mbw = __builtin_amdgcn_alignbit(mbw, __float_as_uint(acc2[i][j][4 * gq + t]), 31);
        }
        if (F32ST) {
            const long px0 = gp0 + (pt2 + j) * 32;
            if (px0 < P_all) {
                const long tb_off = (px0 >> 5) * (HID * 32);
                if (TAPE) {
                    _Float16* tb = reinterpret_cast<_Float16*>(a.tape_h2) + tb_off;
#pragma unroll
                    for (int t = 0; t < 4; ++t) tb[((ms_row0 + o + t) << 5) + ml] = (_Float16)(-acc2[i][j][4 * gq + t] * a.out_scale);
                } else {
                    float* tb = a.tape_h2 + tb_off;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        __builtin_nontemporal_store(-acc2[i][j][4 * gq + t] * a.out_scale, tb + ((ms_row0 + o + t) << 5) + ml);
                }
            }
        }
    };
    if (!STORE) {
#pragma unroll
        for (int i = 0; i < RT2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int o = (rt2 + i) * 32 + 8 * gq + 4 * kl;           // workgroup-local row
                const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(t_rs2 + o);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(t_b2 + o);
#pragma unroll
                for (int j = 0; j < PT2; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc2[i][j][4 * gq + t] = nrelu_bits(fmaf(acc2[i][j][4 * gq + t], rs[t], bb[t]));
            }
    } else {
        // taping / backward: tile by tile -- BWD requests a tile's sign word of h1 one tile ahead
        unsigned mnext = 0u;
        auto load_m = [&](int i, int j) {
            const long px = min(gp0 + (pt2 + j) * 32 + ml, P_all - 1);
            return (unsigned)a.mask1[((long)(ms_row0 / 32 + rt2 + i) * P_all + px) * 2 + kl];
        };
        if (BWD) mnext = load_m(0, 0);
#pragma unroll
        for (int i = 0; i < RT2; ++i)
#pragma unroll
            for (int j = 0; j < PT2; ++j) {
                unsigned mbw = BWD ? mnext : 0u;
                if (BWD && (i * PT2 + j + 1 < RT2 * PT2)) mnext = load_m((i * PT2 + j + 1) / PT2, (i * PT2 + j + 1) % PT2);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) p2_group(i, j, gq, mbw);
                if (MSKST) {
                    const long px0 = gp0 + (pt2 + j) * 32;
                    if (px0 < P_all) a.mask2[((long)(ms_row0 / 32 + rt2 + i) * P_all + px0 + ml) * 2 + kl] = (unsigned short)mbw;
                }
            }
    }
    GH_STAMP(10);

    // ---- P3: T[m][px] = sum_k W4t[m][k] h2[k][px] over this workgroup's h2 rows (= k range [ms_row0, ms_row0 + MR))
    f32x16_t accT[UPW][4];     // tile 4-index = (row tile within the unit) * NPT + pixel tile
#pragma unroll 1
    for (grp = 0; grp < NG; ++grp) {
    if (NG > 1 && grp > 0) {     // next group of output channels: its row scales and first A sets (the tap sums of the previous group are done
                       // with T = hbuf after this barrier; h2 is still in the accumulator registers and is handed over again)
        __syncthreads();
        const float* rs4g = (const float*)((const char*)a.w4 + (size_t)grp * sh2_image_bytes(HID, g.Mpad4) + sh2_rowscale_off(HID, g.Mpad4));
        for (int e = tid; e < g.Mpad4; e += 512) t_rs4[e] = canon_nan(-rs4g[e]);
        if (wid < nunits) {
            const _Float16* ap = a4_base(wid, 0);
            loadA4(ap, wid % g.NU4, 0, A4[0]);
            loadA4(ap, wid % g.NU4, nsl > 1 ? 1 : 0, A4[1]);
        }
    }
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accT[u][j][r] = 0.f;
    GH_STAMP(27);
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        // the owners of rows [l*LK, (l+1)*LK) pass their h2 to the B side through hbuf
#pragma unroll
        for (int i = 0; i < RT2; ++i) {
            const int wr = (rt2 + i) * 32;                   // first workgroup-local row of this tile
            if (wr / LK != l) continue;
            h4 ho_hi[PT2], ho_lo[PT2];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int chunk = (wr - l * LK) / 8 + (gq & 2) + kl;      // k-permuted position (sh.h sh2_kperm_src)
#pragma unroll
                for (int j = 0; j < PT2; ++j) {
                    h4 hi, lo;
                    const f32x4_t v = {acc2[i][j][4 * gq], acc2[i][j][4 * gq + 1], acc2[i][j][4 * gq + 2], acc2[i][j][4 * gq + 3]};
                    sh2_split4<MIXSPLIT>(v, hi, lo);
                    if ((gq & 1) == 0) { ho_hi[j] = hi; ho_lo[j] = lo; continue; }      // (one 16-byte store per pair of row groups, as in P1)
                    const h8 hi8 = {ho_hi[j][0], ho_hi[j][1], ho_hi[j][2], ho_hi[j][3], hi[0], hi[1], hi[2], hi[3]};
                    const h8 lo8 = {ho_lo[j][0], ho_lo[j][1], ho_lo[j][2], ho_lo[j][3], lo[0], lo[1], lo[2], lo[3]};
                    _Float16* dst = hbuf + ((long)chunk * PXT + (pt2 + j) * 32 + ml) * 8;
                    *reinterpret_cast<h8*>(dst) = hi8;
                    *reinterpret_cast<h8*>(dst + (long)LCH * PXT * 8) = lo8;
                }
            }
        }
        GH_STAMP(11 + 4 * l);
        __syncthreads();
        GH_STAMP(12 + 4 * l);
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int unit = wid + 8 * u;
            if (unit >= nunits) continue;
            const int kp = unit / g.NU4, ru = unit - kp * g.NU4;
            const _Float16* ap = a4_base(unit, l);
            const _Float16* bp = hbuf + ((long)(2 * kp * nsl + kl) * PXT + ml) * 8;
            // (B fragments as in P2 where one round of reads serves all pixel tiles: the hi plane of step st + 1 is requested before
            // the third sweep of step st, the lo plane of step st at its start -- BPRE4; with 128 accumulator registers live the
            // tiles go two at a time and every half-step reads its own fragments)
            constexpr int JW = TP2 == 8 ? 2 : NPT;
            constexpr bool BPRE4 = JW == NPT;
            h8 bhn[BPRE4 ? NPT : 1];
            if (BPRE4) {
#pragma unroll
                for (int jj = 0; jj < NPT; ++jj) bhn[jj] = *reinterpret_cast<const h8*>(bp + jj * 256);
            }
            auto step4 = [&](int st, const h8 (&use)[2 * RTU], h8 (&fill)[2 * RTU], auto ld) {      // ld: as in P1
                if constexpr (std::is_same_v<decltype(ld), std::true_type>) loadA4(ap, ru, st + 2, fill);
                else if (ld) loadA4(ap, ru, st + 2, fill);
#if CN_SB_P3
                __builtin_amdgcn_sched_barrier(0);
#endif
                const _Float16* bs = bp + (long)st * (2 * PXT * 8);
#pragma unroll
                for (int jp = 0; jp < NPT / JW; ++jp) {
                    h8 bh[JW], bl[JW];
#pragma unroll
                    for (int jj = 0; jj < JW; ++jj) {
                        if (BPRE4) bh[jj] = bhn[jj];
                        else bh[jj] = *reinterpret_cast<const h8*>(bs + (JW * jp + jj) * 256);
                        bl[jj] = *reinterpret_cast<const h8*>(bs + (JW * jp + jj) * 256 + (long)LCH * PXT * 8);
                    }
#pragma unroll
                    for (int i = 0; i < RTU; ++i)
#pragma unroll
                        for (int jj = 0; jj < JW; ++jj)
                            accT[u][i * NPT + JW * jp + jj] =
                                __builtin_amdgcn_mfma_f32_32x32x16_f16(use[i], bh[jj], accT[u][i * NPT + JW * jp + jj], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < RTU; ++i)
#pragma unroll
                        for (int jj = 0; jj < JW; ++jj)
                            accT[u][i * NPT + JW * jp + jj] =
                                __builtin_amdgcn_mfma_f32_32x32x16_f16(use[i], bl[jj], accT[u][i * NPT + JW * jp + jj], 0, 0, 0);
                    if (BPRE4) {      // next step's hi plane (clamped: the last step re-reads its own)
                        const _Float16* bn = bp + (long)min(st + 1, nsl - 1) * (2 * PXT * 8);
#pragma unroll
                        for (int jj = 0; jj < JW; ++jj) bhn[jj] = *reinterpret_cast<const h8*>(bn + jj * 256);
                    }
#pragma unroll
                    for (int i = 0; i < RTU; ++i)
#pragma unroll
                        for (int jj = 0; jj < JW; ++jj)
                            accT[u][i * NPT + JW * jp + jj] =
                                __builtin_amdgcn_mfma_f32_32x32x16_f16(use[RTU + i], bh[jj], accT[u][i * NPT + JW * jp + jj], 0, 0, 0);
                }
            };
            int st = 0;
#pragma unroll 1
            for (; st + 5 <= nsl; st += 3) {
                step4(st, A4[0], A4[2], std::true_type{});
                step4(st + 1, A4[1], A4[0], std::true_type{});
                step4(st + 2, A4[2], A4[1], std::true_type{});
            }
            if (st < nsl) step4(st, A4[0], A4[2], st + 2 < nsl);
            if (st + 1 < nsl) step4(st + 1, A4[1], A4[0], st + 3 < nsl);
            if (st + 2 < nsl) step4(st + 2, A4[2], A4[1], st + 4 < nsl);
            if (st + 3 < nsl) step4(st + 3, A4[0], A4[2], false);
            // first two A sets of the wave's next (unit, load)
            {
                int nu = unit + 8, nl = l;
                if (u + 1 >= UPW || nu >= nunits) { nu = wid; nl = l + 1; }
                if (nl < NL && nu < nunits) {
                    const _Float16* np = a4_base(nu, nl);
                    loadA4(np, nu % g.NU4, 0, A4[0]);
                    loadA4(np, nu % g.NU4, nsl > 1 ? 1 : 0, A4[1]);
                }
            }
        }
        GH_STAMP(13 + 4 * l);
        __syncthreads();     // hbuf free again (next load / T staging)
        GH_STAMP(14 + 4 * l);
    }

    // ---- P4: T -> LDS as fp32 [k part][row m][pixel] (row scale applied; lanes = consecutive pixels: conflict-free stores and
    // tap reads; one slab per k part, summed in a fixed order by the readers), then the 9-tap sums
    float* T = reinterpret_cast<float*>(hbuf);
    const int Cout = g.Cg;                                   // T rows of THIS group: row m = tap * Cg + (channel within the group)
    const int CoutT = a.Cout, c0 = NG > 1 ? grp * g.Cg : 0;               // all output channels (strides of the partial sums), first of the group
    const int ppx = 1 << g.lpp;                              // pixels per staging pass (whole sub-tiles when npass = 2)
    const long msN = (long)blockIdx.y * a.N;
    float* hpart = a.scratch;
    float* hup = a.scratch + (long)MS * a.N * CoutT * HW;
    float* hdn = hup + (long)MS * g.tiles * CoutT * W;
    // row scales of this wave's T rows, fetched before the loop (one LDS round trip instead of one per group of four rows)
    f32x4_t rs4v[UPW][RTU][4];
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = wid + 8 * u;
        const int ru = unit < nunits ? unit % g.NU4 : 0;      // (once per unit: as an expression below it was re-derived per element)
#pragma unroll
        for (int i = 0; i < RTU; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                rs4v[u][i][gq] = *reinterpret_cast<const f32x4_t*>(t_rs4 + min(ru * RTU + i, g.NRT4 - 1) * 32 + 8 * gq + 4 * kl);
    }
    const int slab = g.Mpad4 << g.lpp;                        // floats per k part: every row of the padded image has a slot, so
                                                              // the stores below need no per-row predicate
    // T staging of one pass; lpc: log2(pixels per pass) -- a compile-time constant when the tile is staged in one pass (every product
    // shape), so that the sixteen stores of a tile take immediate offsets from ONE base address instead of sixteen computed ones
    auto stage_T = [&](int pass, auto lpc) {
        const int lp = lpc;
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int unit = wid + 8 * u;
            if (unit >= nunits) continue;
            const int kp = unit / g.NU4, ru = unit - kp * g.NU4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rtile = ru * RTU + j / NPT;
                if (rtile >= g.NRT4) continue;                              // (wave-uniform) a unit's surplus row tile
                if ((((j % NPT) * 32) >> lp) != pass) continue;             // (wave-uniform) pixel tile of the other pass
                const int q = (j % NPT) * 32 + ml;
                float* dst = T + kp * slab + ((rtile * 32 + 4 * kl) << lp) + (q - (pass << lp));
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        dst[(8 * gq + t) << lp] = accT[u][j][4 * gq + t] * rs4v[u][j / NPT][gq][t];
            }
        }
    };
    constexpr int LPXT = PXT == 128 ? 7 : (PXT == 64 ? 6 : 5);
#pragma unroll 1
    for (int pass = 0; pass < g.npass; ++pass) {
        GH_STAMP(28);
        if (g.npass == 1) stage_T(0, std::integral_constant<int, LPXT>{});
        else stage_T(pass, g.lpp);
        GH_STAMP(29);
        __syncthreads();
        GH_STAMP(20);
        // own rows: out[c][r][x] = sum over taps whose source row r + dy - 1 lies inside the sub-tile.  One item = one output
        // channel of one pixel; every LDS read is unconditional (the tile's own pixel when the tap falls outside) and SELECTED,
        // so the nine reads of a k part are in flight together.
        // (512 threads are a whole number of passes' pixels: a thread keeps ITS pixel and walks the channels -- the tap geometry is
        // computed once per thread, not once per item)
        {
            const int ql = tid & (ppx - 1);
            const int q = (pass << g.lpp) + ql;
            const int sub = q >> g.lsub, qq = q & submask;
            const int r = qq >> g.wshift, x = qq & (W - 1);
            const long n = n0 + sub;
            int off[9];
            bool ok[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tap = dy * 3 + dx;              // out(r, x) += T[tap (dy, dx)][source (r + dy - 1, x + dx - 1)]
                    ok[tap] = r + dy - 1 >= 0 && r + dy - 1 < g.R && x + dx - 1 >= 0 && x + dx - 1 < W;
                    off[tap] = ((tap * Cout) << g.lpp) + ql + (ok[tap] ? (dy - 1) * W + (dx - 1) : 0);
                }
            if (n < a.N) {
                float* hp = hpart + ((msN + n) * CoutT + c0) * HW + (long)(y0 + r) * W + x;
                const int cstep = 512 >> g.lpp;
                for (int ce = tid >> g.lpp; ce < Cout; ce += cstep) {
                    const int cb = ce << g.lpp;
                    float sum = 0.f;
                    const float* tp = T + cb;
                    for (int kp = 0; kp < g.KS; ++kp, tp += slab) {   // k parts of the reduction, fixed order
                        float v[9];
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) v[tap] = tp[off[tap]];
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) sum += ok[tap] ? v[tap] : 0.f;
                    }
                    hp[(long)ce * HW] = sum;
                }
            }
        }
        // halo rows (NI = 1 only): what the tile's first row gives to image row y0 - 1, its last row to row y0 + R
        if (g.NI == 1 && g.R < H) {
            const int hitems = 2 * Cout * W;
            for (int e = tid; e < hitems; e += 512) {
                const int dn = e >= Cout * W;
                const int rem = e - dn * (Cout * W);
                const int co = rem >> g.wshift, x = rem & (W - 1);
                if (dn ? (y0 + g.R >= H) : (y0 == 0)) continue;
                const int rsrc = dn ? g.R - 1 : 0;
                const int dyt = dn ? 0 : 2;                 // filter row applied by the outside pixel to this source row
                int off[3];
                bool ok[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    ok[dx] = x + dx - 1 >= 0 && x + dx - 1 < W;
                    off[dx] = (((dyt * 3 + dx) * Cout + co) << g.lpp) + rsrc * W + (ok[dx] ? x + dx - 1 : x);
                }
                float sacc = 0.f;
                const float* tp = T;
                for (int kp = 0; kp < g.KS; ++kp, tp += slab) {
                    const float v0 = tp[off[0]], v1 = tp[off[1]], v2 = tp[off[2]];
                    sacc += (ok[0] ? v0 : 0.f) + (ok[1] ? v1 : 0.f) + (ok[2] ? v2 : 0.f);
                }
                (dn ? hdn : hup)[(((long)blockIdx.y * g.tiles + tb) * CoutT + c0 + co) * W + x] = sacc;
            }
        }
        if (pass + 1 < g.npass) __syncthreads();
    }
    }      // output-channel groups
    if (pre_on && blockIdx.y == 0 && lane < g.NI && (a.pre.mode == TAIL_AFFINE_FWD || a.pre.mode == TAIL_AFFINE_REV)) {
        const long long v = ld_slot[wid * 2 + lane];         // (written by this wave's lane 0 in the window phase)
        if (v != 0 && n0 + lane < a.N) atomicAdd(a.acc + (n0 + lane), (unsigned long long)v);
    }
    GH_STAMP(21);
    GH_WG_END();
}

// ------------------------------------------------------------------------------------------------ finishing kernel
// (CfinArgs, cfinish_chunk: cnet_fin.h -- the same body runs at the end of k_cnet1w for the tiles a workgroup arrives last at)
template <int PXB, int MSV, bool HALO>
__global__ void __launch_bounds__(256) k_cfinish(CfinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];   // [C][PXB] values, then [C*C] matrix
    __shared__ long long red[4];
    const FinSrc f = fin_src(a.p, a.N, a.H, a.W, a.HW, a.wshift);
    constexpr int LPXB = PXB == 256 ? 8 : (PXB == 64 ? 6 : 4);
    const int chunk = a.xcd_affine && f.lpxt >= LPXB ? cfin_chunk(blockIdx.x, gridDim.x, f.lpxt - LPXB) : (int)blockIdx.x;
    cfinish_chunk<PXB, MSV, HALO, false>(a, f, chunk, (int)(blockIdx.x % (1 + ACC_EXTRA)), fsm, red);
}

// ------------------------------------------------------------------------------------------------ host side
static int g_cnet_ms = 0, g_cnet_flags = 0;
bool cnet_chain_enabled() { return (g_cnet_flags & 4) != 0; }
void cnet_force(int ms, int flags) { g_cnet_ms = ms; g_cnet_flags = flags; }

int cnet_g0(int Cin) { return (9 * ((Cin + 7) / 8) + 1) / 2 * 2; }   // 8-wide k groups of f.0, padded to whole k-steps (two groups each)
int cnet_mpad4(int Cout) { return (9 * Cout + 31) / 32 * 32; }

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// LDS: activation buffer + window planes + tables (rs0 | b0 | rs2 | b2 (at most hidden rows) | rs4 | goff)
static size_t cnet_lds_bytes(const CnetGeo& g, int hidden) {
    return (size_t)CN_HBUF + (size_t)2 * g.winplane * sizeof(_Float16) + ((size_t)4 * hidden + g.Mpad4 + g.G + 4 + 36) * sizeof(float);
}

// f.4 output-channel groups: one up to 56 channels, else the fewest (2..4) equal even-sized groups of at most 56 (0: unsupported)
int cnet_groups(int Cout) {
    if (Cout <= 56) return 1;
    for (int ng = 2; ng <= 4; ++ng)
        if (Cout % ng == 0 && Cout / ng <= 56 && (Cout / ng) % 2 == 0) return ng;
    return 0;
}
size_t cnet_w4_bytes(int hidden, int Cout) {
    const int ng = cnet_groups(Cout);
    return ng ? (size_t)ng * sh2_image_bytes(hidden, cnet_mpad4(Cout / ng)) : 0;
}

bool cnet_geo(int Cin, int H, int W, int hidden, int Cout, int N, int pxt, CnetGeo* out) {
    if (!(hidden == 64 || hidden == 128 || hidden == 256 || hidden == 512)) return false;
    if (pxt == 64 && hidden < 128) return false;
    if (!pow2(W) || !pow2(H) || W < 4 || W > pxt) return false;
    const int HW = H * W;
    if (HW < 64) return false;
    if (Cin < 1 || Cout < 1) return false;
    const int ng = cnet_groups(Cout);
    if (ng == 0 || ng > 2 || (ng == 2 && pxt != 64)) return false;      // (instantiated: one group; two groups at 64-pixel tiles)
    CnetGeo g{};
    g.ng = ng; g.Cg = Cout / ng;
    Cout = g.Cg;      // everything below describes ONE group of f.4 output channels
    g.HW = HW; g.lhw = __builtin_ctz(HW);
    g.pxt = pxt; g.lpxt = __builtin_ctz(pxt);
    g.wshift = __builtin_ctz(W);
    if (HW >= pxt) { g.NI = 1; g.R = pxt / W; g.lsub = g.lpxt; }
    else { g.NI = pxt / HW; g.R = H; g.lsub = __builtin_ctz(HW); }
    if (g.NI > 2) return false;
    g.WP = W + 2;
    g.Wpx = (g.R + 2) * g.WP;
    g.nchunk = (Cin + 7) / 8;
    g.G = cnet_g0(Cin);
    g.steps0 = g.G / 2;
    g.Mpad4 = cnet_mpad4(Cout);
    g.NRT4 = g.Mpad4 / 32;
    if (g.NRT4 > 16) return false;
    const int rtu = 4 / (pxt / 32);                     // row tiles of T per unit
    g.NU4 = (g.NRT4 + rtu - 1) / rtu;
    g.KS = g.NU4 <= 2 ? 4 : (g.NU4 <= 4 ? 2 : 1);
    if ((hidden / CN_MAXMS / 16) / g.KS < 1 && g.KS > 1) g.KS = 1;      // (at least one k-step per part at the largest row split)
    g.npass = (size_t)pxt * 9 * Cout * g.KS * sizeof(float) > (size_t)CN_HBUF ? 2 : 1;     // T staging [k part][9 Cout][pixels] fp32
    if (g.npass == 2 && (g.NI != 2 || (size_t)(pxt / 2) * 9 * Cout * g.KS * sizeof(float) > (size_t)CN_HBUF)) return false;
    g.winplane = g.nchunk * g.NI * g.Wpx * 8;
    auto magic = [](unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); };      // exact quotients for numerators < 2^16
    g.m_nwin = magic((unsigned)(g.NI * g.Wpx)); g.m_Wpx = magic((unsigned)g.Wpx); g.m_WP = magic((unsigned)g.WP);
    if ((long)g.nchunk * g.NI * g.Wpx >= 65536) return false;
    g.lpp = g.lpxt - (g.npass - 1);
    if (cnet_lds_bytes(g, hidden) > 160 * 1024) return false;
    g.tiles = N > 0 ? (int)(((long)N * HW + pxt - 1) / pxt) : 0;
    if (out) *out = g;
    return true;
}

bool cnet_supported(int Cin, int H, int W, int hidden, int Cout) {
    return cnet_geo(Cin, H, W, hidden, Cout, 0, 128, nullptr) || cnet_geo(Cin, H, W, hidden, Cout, 0, 64, nullptr);
}

// scratch: MS partial copies of the f.4 output + the halo rows of every tile (bounded with the smaller tile, the larger split)
size_t cnet_scratch_floats(int N, int H, int W, int Cout) {
    const long tiles = ((long)N * H * W + 63) / 64;
    return (size_t)CN_MAXMS * ((size_t)N * Cout * H * W + (size_t)2 * tiles * Cout * W);
}

size_t cnet_scratch_floats_per_sample(int H, int W, int Cout) {
    const long tiles = ((long)H * W + 63) / 64;
    return (size_t)CN_MAXMS * ((size_t)Cout * H * W + (size_t)2 * tiles * Cout * W);
}

template <int HID, int MS, int UPW, int PXT>
static int launch_cnet_inst2(const CnetArgs& a, const CnetGeo& g, hipStream_t s) {     // two groups of f.4 output channels (C = 96)
    const size_t lds = cnet_lds_bytes(g, HID);
    GH_REQUIRE(!a.pre_on, "cnet: no chained launch with output-channel groups");
    (void)hipFuncSetAttribute((const void*)k_cnet<HID, MS, UPW, PXT, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_cnet<HID, MS, UPW, PXT, false, 2>), dim3(g.tiles, MS), dim3(512), lds, s, a, g);
    GH_LAUNCH_CHECK("k_cnet");
    return GLOWHIP_OK;
}

template <int HID, int MS, int UPW, int PXT>
static int launch_cnet_inst(const CnetArgs& a, const CnetGeo& g, hipStream_t s) {
    const size_t lds = cnet_lds_bytes(g, HID);
    if (a.pre_on) {
        (void)hipFuncSetAttribute((const void*)k_cnet<HID, MS, UPW, PXT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cnet<HID, MS, UPW, PXT, true>), dim3(g.tiles, MS), dim3(512), lds, s, a, g);
    } else {
        (void)hipFuncSetAttribute((const void*)k_cnet<HID, MS, UPW, PXT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cnet<HID, MS, UPW, PXT, false>), dim3(g.tiles, MS), dim3(512), lds, s, a, g);
    }
    GH_LAUNCH_CHECK("k_cnet");
    return GLOWHIP_OK;
}

// tile size, row split and T units per wave of a launch
static bool cnet_select(const CnetArgs& a, CnetGeo* gout, int* ms_out, int* upw_out) {
    // Tile size and row split.  128-pixel tiles halve the weight bytes per MFMA and are taken whenever they alone give every CU a
    // workgroup.  Below that, 64-pixel tiles double the workgroup count without recomputing anything; splitting the h2 rows over
    // MS workgroups per tile (each recomputing h1) comes last.
    CnetGeo g128, g64, g;
    const bool ok128 = cnet_geo(a.Cin, a.H, a.W, a.hidden, a.Cout, a.N, 128, &g128);
    const bool ok64 = cnet_geo(a.Cin, a.H, a.W, a.hidden, a.Cout, a.N, 64, &g64);
    if (!ok128 && !ok64) return false;
    bool use64 = !ok128 || (ok64 && g128.tiles < 224);
    if (g_cnet_flags & 1) use64 = !ok128;      // testing: 128-pixel tiles wherever they exist
    if (g_cnet_flags & 2) use64 = ok64;        // testing: 64-pixel tiles wherever they exist
    // taping / backward launches: the 128-pixel instance is at the register limit and spills once the stores and the sign words
    // are in (184 B); 64-pixel tiles measured 2 % faster on the training step
    if (a.tape_h1 && ok64 && !(g_cnet_flags & 1)) use64 = true;
    // ... except where the one-wave-per-SIMD kernel takes the taping forward (cnet1w_sh.hip: its registers hold the 128-pixel tile)
    const bool c1w_ok = ok128 && !a.pre_on && !(g_cnet_flags & (2 | 16));
    if (a.tape_h1 && c1w_ok && g128.tiles >= 224 && !(a.bwd && (g_cnet_flags & 64)) && cnet1w_takes(a, g128, 1)) use64 = false;      // (flag 64: no backward instance, A/B)
    // ... and, behind the debug switch 0x20000 only, that kernel's instance with the h2 rows split over two workgroups (128-pixel tiles
    // from 112 tiles on: the C = 24 levels that otherwise run one 64-pixel k_cnet workgroup per CU).  Measured slower than k_cnet
    // where it applies -- 47.0 against 45.5 us per launch at config B's level 2, 176 against 150 us at config E's (DESIGN.md 3.2:
    // each half computes all of f.0 again, 29 % of its MFMAs) -- so it is not selected by default; the parity tests run it.
    bool c1w_split = false;
    if (c1w_ok && !a.bwd && !g_cnet_ms && (g_cnet_flags & 32) && g128.tiles >= 112 && cnet1w_takes(a, g128, 2)) { use64 = false; c1w_split = true; }
    g = use64 ? g64 : g128;
    int ms = 1;
    const int ms_max = std::min(CN_MAXMS, a.hidden / (use64 ? 128 : 64));
    while (ms < ms_max && g.tiles * ms < 160) ms *= 2;
    if (g_cnet_ms) ms = std::min(g_cnet_ms, ms_max);
    // T units per wave: (unit rows of T) x (k parts) over 8 waves.  Two units per wave next to the 128 accumulator registers of
    // a 512-row x 128-pixel h2 block would spill: that combination runs with the rows split in two
    const int upw = g.NU4 * g.KS > 8 ? 2 : 1;
    if (upw == 2 && a.hidden == 512 && ms == 1 && !use64) ms = 2;
    while ((a.hidden / ms / 16) / g.KS < 1 && ms > 1) ms /= 2;
    if (c1w_split) ms = 2;
    *gout = g; *ms_out = ms; *upw_out = upw;
    return true;
}

// taping instances (TAPE = true): the training shapes -- one group of f.4 output channels, one T unit per wave
static bool cnet_tape_instance(int hidden, int ms, int upw, int pxt, int ng) {
    return ng == 1 && upw == 1 && (hidden == 512 || (hidden == 256 && ms <= 2) || (hidden == 128 && ms == 1));
}

bool cnet_tape_supported(int Cin, int H, int W, int hidden, int Cout, int N) {
    CnetArgs a{};
    a.Cin = Cin; a.H = H; a.W = W; a.hidden = hidden; a.Cout = Cout; a.N = N;
    a.tape_h1 = reinterpret_cast<float*>(16);      // (selection only: taping launches have their own tile preference)
    CnetGeo g; int ms, upw;
    if ((H * W) % 32 != 0 || !cnet_select(a, &g, &ms, &upw)) return false;
    return cnet_tape_instance(hidden, ms, upw, g.pxt, g.ng);
}

template <int HID, int MS, int PXT, int MODE>
static int launch_cnet_tape(const CnetArgs& a, const CnetGeo& g, hipStream_t s) {
    const size_t lds = cnet_lds_bytes(g, HID);
    (void)hipFuncSetAttribute((const void*)k_cnet<HID, MS, 1, PXT, false, 1, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_cnet<HID, MS, 1, PXT, false, 1, MODE>), dim3(g.tiles, MS), dim3(512), lds, s, a, g);
    GH_LAUNCH_CHECK("k_cnet (taping)");
    return GLOWHIP_OK;
}

int launch_cnet_main(const CnetArgs& a, hipStream_t s, CnetPending* out) {
    GH_REQUIRE(a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV || a.mode == TAIL_ADD_FWD || a.mode == TAIL_ADD_REV,
               "cnet: coupling modes only");
    if (a.N == 0) return GLOWHIP_OK;
    CnetGeo g; int ms, upw;
    GH_REQUIRE(cnet_select(a, &g, &ms, &upw), "cnet: unsupported shape");
    GH_REQUIRE(!a.pre_on || a.pre.MS == ms, "cnet: a chained launch needs the previous step's row split");
    const bool tape = a.tape_h1 != nullptr;
    GH_REQUIRE(!tape || (a.tape_h2 && a.mask1 && a.mask2 && !a.pre_on && g.HW % 32 == 0 &&
                         cnet_tape_instance(a.hidden, ms, upw, g.pxt, g.ng)),
               "cnet: no taping / backward instance for this launch");
    int rc = GLOWHIP_EINVAL;
    // one wave per SIMD (cnet1w_sh.hip) wherever an instance exists for the launch: forward / inverse / taping forward, 128-pixel tiles, no row split
    bool one_wave = false;
    if (!a.pre_on && g.pxt == 128 && !(g_cnet_flags & 16) && !(a.bwd && (g_cnet_flags & 64)) && cnet1w_takes(a, g, ms) && !(ms == 2 && (g_cnet_ms || !(g_cnet_flags & 32)))) {
        GH_TRY(launch_cnet1w(a, g, ms, s));
        rc = GLOWHIP_OK;
        one_wave = true;
    }
#define GH_CNT(hid, m, px)                                                                                     \
    if (rc == GLOWHIP_EINVAL && tape && a.hidden == hid && ms == m && g.pxt == px)                             \
        rc = a.bwd ? launch_cnet_tape<hid, m, px, 2>(a, g, s) : launch_cnet_tape<hid, m, px, 1>(a, g, s);
    GH_CNT(512, 1, 128) GH_CNT(512, 2, 128) GH_CNT(512, 4, 128) GH_CNT(512, 1, 64) GH_CNT(512, 2, 64) GH_CNT(512, 4, 64)
    GH_CNT(256, 1, 128) GH_CNT(256, 2, 128) GH_CNT(256, 1, 64) GH_CNT(256, 2, 64) GH_CNT(128, 1, 128) GH_CNT(128, 1, 64)
#undef GH_CNT
#define GH_CN(hid, m, u, px) if (rc == GLOWHIP_EINVAL && !tape && g.ng == 1 && a.hidden == hid && ms == m && upw == u && g.pxt == px) rc = launch_cnet_inst<hid, m, u, px>(a, g, s);
#define GH_CN2(hid, m, u, px) if (!tape && g.ng == 2 && a.hidden == hid && ms == m && upw == u && g.pxt == px) rc = launch_cnet_inst2<hid, m, u, px>(a, g, s);
    GH_CN2(512, 1, 1, 64) GH_CN2(512, 2, 1, 64) GH_CN2(512, 4, 1, 64) GH_CN2(256, 1, 1, 64) GH_CN2(256, 2, 1, 64) GH_CN2(128, 1, 1, 64)
#undef GH_CN2
    GH_CN(512, 1, 1, 128) GH_CN(512, 2, 1, 128) GH_CN(512, 4, 1, 128) GH_CN(256, 1, 1, 128) GH_CN(256, 2, 1, 128) GH_CN(256, 4, 1, 128)
    GH_CN(128, 1, 1, 128) GH_CN(128, 2, 1, 128) GH_CN(64, 1, 1, 128)
    GH_CN(512, 2, 2, 128) GH_CN(512, 4, 2, 128) GH_CN(256, 1, 2, 128) GH_CN(256, 2, 2, 128) GH_CN(256, 4, 2, 128) GH_CN(128, 1, 2, 128)
    GH_CN(128, 2, 2, 128) GH_CN(64, 1, 2, 128)
    GH_CN(512, 1, 1, 64) GH_CN(512, 2, 1, 64) GH_CN(512, 4, 1, 64) GH_CN(256, 1, 1, 64) GH_CN(256, 2, 1, 64) GH_CN(128, 1, 1, 64)
    GH_CN(512, 1, 2, 64) GH_CN(512, 2, 2, 64) GH_CN(512, 4, 2, 64) GH_CN(256, 1, 2, 64) GH_CN(256, 2, 2, 64) GH_CN(128, 1, 2, 64)
#undef GH_CN
    if (rc == GLOWHIP_EINVAL) set_error("cnet: no kernel instance for hidden=%d ms=%d upw=%d tile=%d", a.hidden, ms, upw, g.pxt);
    GH_TRY(rc);
    if (out) {
        out->scratch = a.scratch; out->MS = ms; out->tiles = g.tiles; out->R = g.R; out->NI = g.NI; out->lpxt = g.lpxt;
        out->bias = a.bias; out->scale = a.scale; out->mode = a.mode; out->Cout = a.Cout;
        out->z = a.pre_on ? a.pre_z_new : a.z_in;
        out->z_bs = a.pre_on ? a.pre_z_new_bs : a.z_in_bs;
        out->one_wave = one_wave ? 1 : 0;
        out->finished = one_wave && cnet1w_finishes(a, g, ms) ? 1 : 0;      // (launch_cnet1w took the fused-finishing instance)
    }
    return GLOWHIP_OK;
}

int launch_cnet_finish(const CnetArgs& a, const CnetPending& p, hipStream_t s) {
    if (a.N == 0) return GLOWHIP_OK;
    const int HW = a.H * a.W;
    CfinArgs f{p, a.mix, a.z_out, a.z_out_bs, a.acc, a.N, a.H, a.W, HW, __builtin_ctz(a.W), (g_cnet_flags & 8) ? 0 : 1, a.tape_hout};
    const bool paired = p.mode == TAIL_AFFINE_FWD || p.mode == TAIL_AFFINE_REV;
    const int C = 2 * (paired ? p.Cout / 2 : p.Cout);
    GH_REQUIRE(a.mix.C == 0 || a.mix.C == C, "cnet: mixer channel count %d != %d", a.mix.C, C);
    const long total_px = (long)a.N * HW;
    // small levels: 16 pixels per workgroup, so that the launch still covers the chip; large ones (config D / E level 1): 256
    const int pxb = total_px < 16384 ? 16 : (total_px >= 131072 && total_px % 256 == 0 && HW % 256 == 0 && p.lpxt <= 8 ? 256 : 64);
    const size_t flds = ((size_t)C * pxb + (a.mix.C && a.mix.matrix ? (size_t)C * C : 0)) * sizeof(float);
    GH_REQUIRE(flds <= 64 * 1024, "cnet: finishing kernel LDS");
    const bool halos = p.NI == 1 && p.R < a.H;
    bool launched = false;
#define GH_CF(px, m, h)                                                                                                       \
    if (!launched && pxb == px && p.MS == m && halos == h) {                                                                  \
        if (flds > 32 * 1024)                                                                                                 \
            (void)hipFuncSetAttribute((const void*)k_cfinish<px, m, h>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds); \
        hipLaunchKernelGGL((k_cfinish<px, m, h>), dim3((unsigned)(total_px / px)), dim3(256), flds, s, f);                    \
        launched = true;                                                                                                      \
    }
    GH_CF(16, 1, false) GH_CF(16, 1, true) GH_CF(16, 2, false) GH_CF(16, 2, true) GH_CF(16, 4, false) GH_CF(16, 4, true)
    GH_CF(64, 1, false) GH_CF(64, 1, true) GH_CF(64, 2, false) GH_CF(64, 2, true) GH_CF(64, 4, false) GH_CF(64, 4, true)
    GH_CF(256, 1, false) GH_CF(256, 1, true) GH_CF(256, 2, false) GH_CF(256, 2, true)
#undef GH_CF
    GH_REQUIRE(launched, "cnet: no finishing kernel for row split %d", p.MS);
    GH_LAUNCH_CHECK("k_cfinish");
    return GLOWHIP_OK;
}

int launch_cnet(const CnetArgs& a, hipStream_t s) {
    CnetPending p{};
    GH_TRY(launch_cnet_main(a, s, &p));
    if (a.N == 0) return GLOWHIP_OK;
    return launch_cnet_finish(a, p, s);
}

// window-time finishing stages [C][window] + [C][C] + [Cin][window] floats in the (then idle) activation buffer
bool cnet_pre_supported(int Cin, int H, int W, int hidden, int Cout, int C) {
    if (C > 96 || cnet_groups(Cout) != 1) return false;
    for (int pxt : {128, 64}) {
        CnetGeo g;
        if (!cnet_geo(Cin, H, W, hidden, Cout, 0, pxt, &g)) continue;
        const size_t kp = ((size_t)g.NI * (g.R + 2) * W + 63) / 64 * 64;      // upper bound of the in-image window pixels
        if (((size_t)C * kp + (size_t)Cin * kp + (size_t)C * (C + 3) + 32) * sizeof(float) > (size_t)CN_HBUF) return false;
    }
    return true;
}

}  // namespace glowhip
