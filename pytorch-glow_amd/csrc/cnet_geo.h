// cnet_geo.h -- tile geometry of the one-kernel coupling network, shared by its two kernel files: cnet_sh.hip (k_cnet: eight
// waves, two per SIMD) and cnet1w_sh.hip (k_cnet1w: four waves, one per SIMD, 512 registers each).
#pragma once
#include "sh.h"

namespace glowhip {

constexpr int CN_HBUF = 128 * 1024;          // bytes of the h1 / h2 / T region
constexpr int CN_MAXMS = 4;

struct CnetGeo {
    int wshift, lsub, NI, R, WP, Wpx, nchunk, G, steps0, Mpad4, NRT4, NU4, KS, npass, tiles;
    int winplane;     // halfs per window plane
    int HW, lhw;      // pixels per image and its log2
    int lpp;          // log2(pixels per staging pass)
    int pxt, lpxt;    // pixels per workgroup tile (128 or 64) and its log2
    unsigned m_nwin, m_Wpx, m_WP;     // ceil(2^32 / d) of the window's three divisors: n / d = umulhi(n, m) for the slot indices (< 2^16)
    int ng, Cg;       // f.4 in ng groups of Cg output channels (Mpad4, NRT4, NU4, KS, npass describe ONE group): wide steps
                      // (C = 96: Cout = 96 = 2 x 48) run P3 + P4 once per group, h2 handed over again from the registers
};

__host__ __device__ inline int cnet_trow(int M9) {   // T row stride (floats): multiple of 4, an odd multiple (bank spread)
    int r = (M9 + 3) / 4;
    if ((r & 1) == 0) ++r;
    return r * 4;
}

// geometry of a launch with pxt-pixel tiles (false: the shape has none); N = 0: shape check only
bool cnet_geo(int Cin, int H, int W, int hidden, int Cout, int N, int pxt, CnetGeo* out);

// ---- k_cnet1w (cnet1w_sh.hip): may this launch run on it with the h2 rows split over ms workgroups per tile, and the launch itself
// (same partial-sum layout as k_cnet with that row split and 128-pixel tiles: the finishing kernel does not know the difference)
bool cnet1w_takes(const CnetArgs& a, const CnetGeo& g128, int ms);
int launch_cnet1w(const CnetArgs& a, const CnetGeo& g, int ms, hipStream_t s);
bool cnet1w_finishes(const CnetArgs& a, const CnetGeo& g, int ms);      // that launch finishes the step itself (CnetArgs::fin_cnt)

}  // namespace glowhip
