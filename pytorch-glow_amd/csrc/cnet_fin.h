// cnet_fin.h -- how the partial sums k_cnet leaves behind (cnet_sh.hip: MS row-split copies of the tile's own rows + the halo
// rows a tile contributes to its neighbours) are gathered for one element.  Used by the finishing kernels of cnet_sh.hip and by
// k_chanmix_bwd (backward.hip), which adds the backward launch's d L / d y1 on the fly.
#pragma once
#include "sh.h"
#include "conv_mfma.h"

namespace glowhip {

// Shared by the finishing kernel and by k_cnet's window-time finishing of the previous step: the two must agree bit for bit.
struct FinSrc {
    const float* hpart; const float* hup; const float* hdn;      // partial sums of h = f(z1): own rows, halo rows up / down
    const float* bias; const float* scale;                       // f.4 bias, exp(3 logs)
    int MS, tiles, R, lpxt, N, Cout, HW, W, H, wshift, mode;
    bool halos, paired;
};

__device__ __forceinline__ FinSrc fin_src(const CnetPending& p, int N, int H, int W, int HW, int wshift) {
    FinSrc f;
    f.hpart = p.scratch;
    f.hup = p.scratch + (long)p.MS * N * p.Cout * HW;
    f.hdn = f.hup + (long)p.MS * p.tiles * p.Cout * W;
    f.bias = p.bias; f.scale = p.scale;
    f.MS = p.MS; f.tiles = p.tiles; f.R = p.R; f.lpxt = p.lpxt; f.N = N; f.Cout = p.Cout; f.HW = HW; f.W = W; f.H = H;
    f.wshift = wshift; f.mode = p.mode;
    f.halos = p.NI == 1 && p.R < H;
    f.paired = p.mode == TAIL_AFFINE_FWD || p.mode == TAIL_AFFINE_REV;
    return f;
}

// Partial sums of h = f(z1) for coupling channel c at pixel p of image n: (se, so) = the shift (and, affine, the scale logit) before
// bias and exp(3 logs).  Every load is unconditional (clamped index, selected value), so a caller that gathers several elements
// before using any has all of their loads in flight together.
// COH: the partial sums come from other workgroups of the RUNNING launch (fused finishing): agent-scope relaxed atomic loads.
template <bool COH>
__device__ __forceinline__ float fin_ld(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <int MSV, bool HALO = true, bool COH = false>      // MSV > 0: f.MS known at compile time (loops unrolled: every load of the element in
                                          // flight at once); HALO = false: the caller knows f.halos is false
__device__ __forceinline__ void fin_gather_t(const FinSrc& f, long n, int c, int p, float& se, float& so) {
    const int ms = MSV > 0 ? MSV : f.MS;
    const int y = p >> f.wshift, x = p & (f.W - 1);
    const int ce = f.paired ? 2 * c : c;
    se = 0.f; so = 0.f;
#pragma unroll
    for (int m = 0; m < ms; ++m) {
        const long base = (((long)m * f.N + n) * f.Cout + ce) * f.HW + p;
        se += fin_ld<COH>(f.hpart + base);
        so += fin_ld<COH>(f.hpart + base + (f.paired ? f.HW : 0));
    }
    if (HALO) {   // halo rows: no branch either (without halos the selects below drop a valid but unused slot of the scratch buffer)
        const int r = y & (f.R - 1);
        const long tile = (n * f.HW + (long)(y - r) * f.W) >> f.lpxt;    // tile holding row y
        const bool wd = f.halos && r == 0 && y > 0;                   // row below the previous tile: its `hdn`
        const bool wu = f.halos && r == f.R - 1 && y < f.H - 1;       // row above the next tile: its `hup`
        const long td = tile > 0 ? tile - 1 : 0, tu = tile + 1 < f.tiles ? tile + 1 : tile;
#pragma unroll
        for (int m = 0; m < ms; ++m) {
            const long hd = (((long)m * f.tiles + td) * f.Cout + ce) * f.W + x;
            const long hu = (((long)m * f.tiles + tu) * f.Cout + ce) * f.W + x;
            // loaded unconditionally, SELECTED (never multiplied by a 0/1 mask: an unused slot of the scratch buffer may hold
            // NaN or inf bit patterns, and 0 * NaN is NaN)
            const float d0 = fin_ld<COH>(f.hdn + hd), u0 = fin_ld<COH>(f.hup + hu);
            const float d1 = fin_ld<COH>(f.hdn + hd + (f.paired ? f.W : 0)), u1 = fin_ld<COH>(f.hup + hu + (f.paired ? f.W : 0));
            se += (wd ? d0 : 0.f) + (wu ? u0 : 0.f);
            so += (wd ? d1 : 0.f) + (wu ? u1 : 0.f);
        }
    }
}

__device__ __forceinline__ void fin_gather(const FinSrc& f, long n, int c, int p, float& se, float& so) { fin_gather_t<0>(f, n, c, p, se, so); }

// Updated z2 value of coupling channel c given its current value zin and the gathered sums; the log-det term of the element is
// added to ldq as Q31.32 fixed point (integer sums are exact: the per-sample total does not depend on how elements are grouped
// into workgroups, so every user produces the same bits).
__device__ __forceinline__ float fin_apply_k(const FinSrc& f, float se, float so, float zin, float bias_e, float scale_e, float bias_o,
                                             float scale_o, long long& ldq, float& bad) {
    const float A_ = (se + bias_e) * scale_e;
    if (!f.paired) return f.mode == TAIL_ADD_FWD ? zin + A_ : zin - A_;
    const float B_ = (so + bias_o) * scale_o;
    const float sc = sigmoidf_(B_ + 2.0f);
    const float lg = logf(sc);
    const float zr = f.mode == TAIL_AFFINE_FWD ? (zin + A_) * sc : zin / sc - A_;
    // a non-finite log-det term (fp16-range overflow upstream, diverged weights, a saturated sigmoid) cannot go into the
    // fixed-point sum: `bad` returns it and the caller raises the sample's sticky flag (common.h).  A non-finite z needs no flag
    // of its own: it reaches a prior's logp, which is then non-finite.
    const float term = f.mode == TAIL_AFFINE_FWD ? lg : -lg;
    if (isfinite(term)) ldq += __double2ll_rn((double)term * FIX_SCALE);
    else bad = term;
    return zr;
}
__device__ __forceinline__ float fin_apply(const FinSrc& f, int c, float se, float so, float zin, long long& ldq, float& bad) {
    const int ce = f.paired ? 2 * c : c;
    return fin_apply_k(f, se, so, zin, f.bias[ce], f.scale[ce], f.bias[ce + (f.paired ? 1 : 0)], f.scale[ce + (f.paired ? 1 : 0)], ldq, bad);
}

__device__ __forceinline__ float fin_couple(const FinSrc& f, long n, int c, int p, float zin, long long& ldq, float& bad) {
    float se, so;
    fin_gather(f, n, c, p, se, so);
    return fin_apply(f, c, se, so, zin, ldq, bad);
}

// sum of a Q31.32 term over a workgroup of NT threads (valid in thread 0); red: NT / 64 slots of LDS
template <int NT>
__device__ __forceinline__ long long block_sum_ll(long long v, long long* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    long long t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < NT / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}


// One workgroup = 64 consecutive pixels of one image x all channels: sums the MS partials and the neighbour tiles' halo rows,
// (h + bias) * exp(3 logs), coupling, per-sample log-det, then the channel mixer on the finished pixels.
struct CfinArgs {
    CnetPending p;            // partial sums + coupling of the step being finished
    CnetMixer mix;
    float* z_out; long z_out_bs;
    unsigned long long* acc;
    int N, H, W, HW, wshift;
    int xcd_affine;
    float* tape_hout;         // training tape (CnetArgs): hout, and the coupled z2 written back into p.z when mixing out of place
};

// XCD affinity (speed only; any block-to-chunk permutation is correct).  The hardware places block b on XCD b % 8, and each XCD
// has its own L2.  k_cnet's tile t -- whose partial sums this kernel reads, and whose successor in the next FlowStep reads the
// state this kernel writes -- runs on XCD t % 8.  So block b = 8 s + x takes a pixel chunk of a tile t with t % 8 == x: partial
// sums and state then travel between kernels through ONE XCD's L2 instead of through memory (halo rows excepted).
// lr = log2(chunks per k_cnet tile).
__device__ __forceinline__ int cfin_chunk(int b, int nblk, int lr) {
    if (nblk & ((8 << lr) - 1)) return b;                 // not whole groups of 8 tiles: identity
    const int x = b & 7, s = b >> 3;
    return ((((s >> lr) << 3) + x) << lr) + (s & ((1 << lr) - 1));
}

// The finishing of ONE chunk of PXB consecutive pixels of one image x all channels by 256 threads -- the body of k_cfinish
// (cnet_sh.hip), and of the fused finishing at the end of k_cnet1w (cnet1w_sh.hip: the workgroup that arrives LAST at a tile's counter
// finishes the tile; COH = true there: the partial sums were written by other workgroups of the SAME launch, possibly on another XCD,
// so they are read as agent-scope relaxed atomic loads -- around the XCD's L2 -- as they were written).  Same code, same operation
// order: the two forms agree bit for bit.  `accrow`: which of the sample's 1 + ACC_EXTRA log-det accumulator rows takes the atomic.
// fsm: (C * PXB + C * C) floats of LDS; red: 4 words.  Ends without a barrier: a caller that reuses fsm / red must place one.
template <int PXB, int MSV, bool HALO, bool COH>
__device__ __forceinline__ void cfinish_chunk(const CfinArgs& a, const FinSrc& f, int chunk, int accrow, float* fsm, long long* red) {
    const int tid = threadIdx.x;
    const int HW = a.HW;
    const long gp0 = (long)chunk * PXB;
    const long n = gp0 / HW;
    const int p0 = (int)(gp0 - n * HW);
    const int Ch = f.paired ? f.Cout / 2 : f.Cout;         // channels of z2 (= C/2)
    const int C = 2 * Ch;
    float* mixv = fsm;
    float* mixm = fsm + C * PXB;
    // The launch is latency-bound (a few KB per workgroup): EVERYTHING it reads from memory -- the mixer matrix, the state, the
    // MS partial sums and halo rows, the per-channel constants -- is requested in one round, branch-free (fin_gather_t), before the
    // first value is used.
    constexpr int MREG = 10;                     // C <= 48: 2304 / 256 = 9 values per thread; wider mixers use the loop below
    float mreg[MREG];
    const bool mfast = a.mix.C && a.mix.matrix && C * C <= MREG * 256;
    if (mfast) {      // unconditional, from clamped addresses (a lane-predicated load is a branch: hipcc then waited for ALL of these
                      // before it issued the first load of the round below -- two trips to memory instead of one)
#pragma unroll
        for (int k = 0; k < MREG; ++k) mreg[k] = a.mix.matrix[min(tid + 256 * k, C * C - 1)];
    }
    const float* zi = a.p.z + n * a.p.z_bs;
    float* zn = a.z_out + n * a.z_out_bs;
    long long ldq = 0;
    // elements per thread and round: with 256-pixel workgroups (launches of 131 072 pixels and more: 64-pixel workgroups came
    // in four rounds per CU, each a full latency chain -- 32 us for 38 MB at config E's 128-wide level) all six at once
    constexpr int U = PXB == 256 ? 6 : 2;
    const int total = Ch * PXB;
    const bool an = a.mix.C && !a.mix.reverse;
    // The first round is peeled: as the first trip of a loop its loads were issued only after everything requested in front of the
    // loop had arrived (the wait-count pass merges the loop's own back edge into the header: `s_waitcnt vmcnt(0)` at its top).
    auto round = [&](int e0) {
        float se[U], so[U], zin[U], z1v[U], kb[U][4], km[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + 256 * u, total - 1);
            const int c = e / PXB, q = e - c * PXB;
            zin[u] = zi[(long)(Ch + c) * HW + p0 + q];
            z1v[u] = zi[(long)c * HW + p0 + q];
            fin_gather_t<MSV, HALO, COH>(f, n, c, p0 + q, se[u], so[u]);
            const int ce = f.paired ? 2 * c : c, co = ce + (f.paired ? 1 : 0);
            kb[u][0] = f.bias[ce]; kb[u][1] = f.scale[ce]; kb[u][2] = f.bias[co]; kb[u][3] = f.scale[co];
            km[u][0] = an ? a.mix.bias[c] : 0.f; km[u][1] = an ? a.mix.scale[c] : 1.f;
            km[u][2] = an ? a.mix.bias[Ch + c] : 0.f; km[u][3] = an ? a.mix.scale[Ch + c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            if (e >= total) continue;
            const int c = e / PXB, q = e - c * PXB;
            const int p = p0 + q;
            float bad = 0.f;
            const float zres = fin_apply_k(f, se[u], so[u], zin[u], kb[u][0], kb[u][1], kb[u][2], kb[u][3], ldq, bad);
            if (bad != 0.f) fix_flag_nonfinite(a.acc, n, a.N, bad);
            if (a.tape_hout) {
                const int ce = f.paired ? 2 * c : c;
                a.tape_hout[(n * f.Cout + ce) * HW + p] = (se[u] + kb[u][0]) * kb[u][1];
                if (f.paired) a.tape_hout[(n * f.Cout + ce + 1) * HW + p] = (so[u] + kb[u][2]) * kb[u][3];
                if (zn != zi) const_cast<float*>(zi)[(long)(Ch + c) * HW + p] = zres;
            }
            if (a.mix.C) {
                if (!a.mix.reverse) {     // ActNorm of the next step on both halves, staged for its matrix / gather
                    mixv[c * PXB + q] = (z1v[u] + km[u][0]) * km[u][1];
                    mixv[(Ch + c) * PXB + q] = (zres + km[u][2]) * km[u][3];
                } else {
                    mixv[c * PXB + q] = z1v[u];
                    mixv[(Ch + c) * PXB + q] = zres;
                }
            } else {
                zn[(long)(Ch + c) * HW + p] = zres;
                if (zn != zi) zn[(long)c * HW + p] = z1v[u];       // out of place: z1 travels along
            }
        }
    };
    round(tid);
    for (int e0 = tid + 256 * U; e0 < total; e0 += 256 * U) round(e0);
    if (mfast) {
#pragma unroll
        for (int k = 0; k < MREG; ++k)
            if (tid + 256 * k < C * C) mixm[tid + 256 * k] = mreg[k];
    } else if (a.mix.C && a.mix.matrix) {      // wider mixers: eight requests in flight per trip (as a plain copy loop every element was a trip)
        for (int e0 = tid; e0 < C * C; e0 += 256 * 8) {
            float mv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mv[k] = a.mix.matrix[min(e0 + 256 * k, C * C - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (e0 + 256 * k < C * C) mixm[e0 + 256 * k] = mv[k];
        }
    }
    if (f.paired) {     // per-sample log-det: ONE atomic per workgroup (64 workgroups and more share a sample's accumulator: per-wave
                        // atomics queued up behind one another and doubled the launch time), issued before the mixer phase,
                        // whose time hides its round trip
        const long long tot = block_sum_ll<256>(ldq, red);
        // (one of the 1 + ACC_EXTRA rows of the sample's accumulators, common.h: k_finalize adds the rows)
        if (tid == 0 && tot != 0) {
            const int row = accrow;
            atomicAdd(a.acc + (row == 0 ? n : (long)(1 + row) * a.N + n), (unsigned long long)tot);
        }
    } else if (a.mix.C) {
        __syncthreads();
    }
    if (a.mix.C) {
        // thread = (output group og, pixel q): outputs o = og, og + OG, ... four at a time (the staged value v[i][q] is read once
        // for four outputs; the matrix rows are wave-uniform LDS broadcasts)
        constexpr int OG = PXB >= 256 ? 1 : 256 / PXB;
        const int q = tid & (PXB - 1), og = PXB >= 256 ? 0 : tid / PXB;
        for (int ob = og; ob < C; ob += 4 * OG) {
            float r[4];
            if (a.mix.matrix) {   // same operation order as k_chanmix: r = fma(m[o][i], v[i], r), i ascending
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = 0.f;
                for (int i = 0; i < C; ++i) {
                    const float vi = mixv[i * PXB + q];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o = ob + j * OG;
                        r[j] = fmaf(mixm[(o < C ? o : 0) * C + i], vi, r[j]);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = ob + j * OG;
                    r[j] = mixv[(o < C ? (a.mix.gather ? a.mix.gather[o] : o) : 0) * PXB + q];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = ob + j * OG;
                if (o >= C) continue;
                float v = r[j];
                if (a.mix.reverse) v = v * a.mix.scale[o] - a.mix.bias[o];
                zn[(long)o * HW + p0 + q] = v;
            }
        }
    }
}


}  // namespace glowhip
