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
template <int MSV, bool HALO = true>      // MSV > 0: f.MS known at compile time (loops unrolled: every load of the element in
                                          // flight at once); HALO = false: the caller knows f.halos is false
__device__ __forceinline__ void fin_gather_t(const FinSrc& f, long n, int c, int p, float& se, float& so) {
    const int ms = MSV > 0 ? MSV : f.MS;
    const int y = p >> f.wshift, x = p & (f.W - 1);
    const int ce = f.paired ? 2 * c : c;
    se = 0.f; so = 0.f;
#pragma unroll
    for (int m = 0; m < ms; ++m) {
        const long base = (((long)m * f.N + n) * f.Cout + ce) * f.HW + p;
        se += f.hpart[base];
        so += f.hpart[base + (f.paired ? f.HW : 0)];
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
            const float d0 = f.hdn[hd], u0 = f.hup[hu], d1 = f.hdn[hd + (f.paired ? f.W : 0)], u1 = f.hup[hu + (f.paired ? f.W : 0)];
            se += (wd ? d0 : 0.f) + (wu ? u0 : 0.f);
            so += (wd ? d1 : 0.f) + (wu ? u1 : 0.f);
        }
    }
}

}  // namespace glowhip
