// f02_sh.hip -- f.0 and f.2 of the coupling network as ONE kernel on split-half operands (sh.h):
//   h1 = relu(conv3x3(z1; W0') + b0')     (network/module.py:314-315, ActNorm folded)      -- never leaves the CU
//   h2 = relu(W2' h1 + b2')               (network/module.py:316-317)                      -- written as an SH tensor
// Separately the two kernels write and re-read h1 (2 x 134 MB per level-1 layer at B=64) and are bound by that traffic;
// fused, h1 for a 64-pixel tile x all `hidden` channels is 128 KiB as (hi, lo) halves and fits the 160 KiB LDS.
//
// Workgroup = one 64-pixel SH tile (R = 64 / W image rows), 8 waves.
//   phase 0: the z1 window (R + 2 rows, zero padded, channels in chunks of 8) is split into halves and stored in LDS;
//   phase 1: h1 by MFMA, a wave taking two 32-channel groups x both 32-pixel halves at a time (weights straight from L2 as
//            A fragments, window fragments from LDS at tap-shifted addresses), written as (hi, lo) halves into the LDS
//            image [plane][chunk][64 pixels][8] -- exactly the B-operand layout of phase 2;
//   phase 2: the 1x1 convolution over ALL of K with the B operand resident: no ring, no barrier, no counted waits -- a
//            wave owns 64 output channels x 64 pixels and free-runs, its A fragments (W2', two k-steps ahead) coming from L2.
// ONE barrier separates the phases.  The redundant work of tiling (each f.2 row tile recomputing h1) is avoided by giving the
// workgroup all `hidden` rows, which is what makes the fusion a net win: +16 % MFMA work at level 1, -50 % HBM traffic.
#include "sh.h"
#include "conv_mfma.h"

GH_STAMPS_DEFINE(f02)

namespace glowhip {

__global__ void __launch_bounds__(512)
k_f02_sh(const float* __restrict__ X, long x_bs, const _Float16* __restrict__ W0, const float* __restrict__ b0,
         const _Float16* __restrict__ W2, const float* __restrict__ b2, _Float16* __restrict__ Ysh, int N, int Cin, int H,
         int W, int M, int wshift) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_q[];
    const int R = 64 >> wshift;                         // image rows of one 64-pixel tile
    const int HW = H * W, WP = W + 2, Wpx = (R + 2) * WP;
    const int nchunk = (Cin + 7) >> 3;
    const int G = (9 * nchunk + 1) & ~1;                // 8-wide k groups of f.0, even count
    const int steps0 = G >> 1;
    const int NCK = M >> 3;                             // 8-channel chunks of h1 (K of f.2 = M)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, ml = lane & 31;
    const long tile = blockIdx.x;                       // = global pixel / 64
    const long gp0 = tile * 64;
    const long n = gp0 / HW;
    const int y0 = (int)(gp0 - n * HW) >> wshift;
    const long win_plane = (long)nchunk * Wpx * 8;      // halfs per window plane
    _Float16* h1 = smem_q + 2 * win_plane;              // [plane][NCK][64][8]
    const long h1_plane = (long)NCK * 64 * 8;

    GH_STAMP(32);
    // requested before the window is built so that their L2 round trips overlap it: the first two A fragment sets of
    // this wave's first phase-1 pass
    const long w0_plane = (long)G * M * 8;
    const long sstep0 = (long)2 * M * 8;
    h8 P0[4], P1[4];
    {
        const int g0 = 2 * wid < (M >> 5) ? 2 * wid : 0;
        const _Float16* p = W0 + ((long)kl * M + g0 * 32 + ml) * 8;
        P0[0] = *reinterpret_cast<const h8*>(p);
        P0[1] = *reinterpret_cast<const h8*>(p + 32 * 8);
        P0[2] = *reinterpret_cast<const h8*>(p + w0_plane);
        P0[3] = *reinterpret_cast<const h8*>(p + w0_plane + 32 * 8);
        const _Float16* p1 = p + (steps0 > 1 ? sstep0 : 0);
        P1[0] = *reinterpret_cast<const h8*>(p1);
        P1[1] = *reinterpret_cast<const h8*>(p1 + 32 * 8);
        P1[2] = *reinterpret_cast<const h8*>(p1 + w0_plane);
        P1[3] = *reinterpret_cast<const h8*>(p1 + w0_plane + 32 * 8);
    }

    // ---- phase 0: window -> (hi, lo) halves in LDS, one (chunk, window pixel) slot = 8 channels
    const float* xin = X + n * x_bs;
    for (int e = tid; e < nchunk * Wpx; e += 512) {
        const int ch = e / Wpx, wp = e - ch * Wpx;
        const int r = wp / WP, c = wp - r * WP;
        const int yy = y0 - 1 + r, xx = c - 1;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        h8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ci = ch * 8 + q;
            const float v = (in && ci < Cin) ? xin[(long)ci * HW + yy * W + xx] : 0.f;
            _Float16 a, b;
            sh_split(v, a, b);
            hi[q] = a; lo[q] = b;
        }
        *reinterpret_cast<h8*>(smem_q + (long)e * 8) = hi;
        *reinterpret_cast<h8*>(smem_q + win_plane + (long)e * 8) = lo;
    }
    __syncthreads();

    GH_STAMP(33);
    // ---- phase 1: h1 = relu(conv3x3 + b0'); a wave takes two 32-channel groups x both 32-pixel halves at once (four
    // independent accumulator pairs keep the matrix pipe busy while the next A fragments travel from L2, two steps ahead)
    {
        const long sstep = sstep0;
        int pbase[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = u * 32 + ml;
            pbase[u] = ((q >> wshift) * WP + (q & (W - 1))) * 8;
        }
        for (int g0 = 2 * wid; g0 < (M >> 5); g0 += 16) {
            f32x16_t am[2][2], ax[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(b0 + (g0 + i) * 32 + 8 * gq + 4 * kl);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t) { am[i][u][4 * gq + t] = b4[t]; ax[i][u][4 * gq + t] = 0.f; }
                }
            const _Float16* ap = W0 + ((long)kl * M + g0 * 32 + ml) * 8;
            auto loadA0 = [&](int st, h8 (&a)[4]) {
                const _Float16* p = ap + (long)st * sstep;
                a[0] = *reinterpret_cast<const h8*>(p);
                a[1] = *reinterpret_cast<const h8*>(p + 32 * 8);
                a[2] = *reinterpret_cast<const h8*>(p + w0_plane);
                a[3] = *reinterpret_cast<const h8*>(p + w0_plane + 32 * 8);
            };
            h8 A0[4], A1[4], A2[4];
            if (g0 == 2 * wid) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { A0[t] = P0[t]; A1[t] = P1[t]; }
            } else {
                loadA0(0, A0);
                if (steps0 > 1) loadA0(1, A1);
            }
#pragma unroll 1
            for (int st = 0; st < steps0; ++st) {
                if (st + 2 < steps0) loadA0(st + 2, A2);
                int g = 2 * st + kl;
                g = g < 9 * nchunk ? g : 0;                  // padded groups carry zero weights
                const int tap = g / nchunk, ch = g - tap * nchunk;
                const int dy = tap / 3, dx = tap - dy * 3;
                const int goff = (ch * Wpx + dy * WP + dx) * 8;
                h8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bh[u] = *reinterpret_cast<const h8*>(smem_q + goff + pbase[u]);
                    bl[u] = *reinterpret_cast<const h8*>(smem_q + win_plane + goff + pbase[u]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        am[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[i], bh[u], am[i][u], 0, 0, 0);
                        ax[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[i], bl[u], ax[i][u], 0, 0, 0);
                        ax[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[2 + i], bh[u], ax[i][u], 0, 0, 0);
                    }
#pragma unroll
                for (int t = 0; t < 4; ++t) { A0[t] = A1[t]; A1[t] = A2[t]; }
            }
            // relu, split, store into the B-operand image: a lane's 4 consecutive channels = 8 bytes per plane
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        h4 hi, lo;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float v = relu_(am[i][u][4 * gq + t] + ax[i][u][4 * gq + t] * SH_LO_INV);
                            _Float16 a, b;
                            sh_split(v, a, b);
                            hi[t] = a; lo[t] = b;
                        }
                        const int chunk = (g0 + i) * 4 + gq;
                        _Float16* dst = h1 + ((long)chunk * 64 + u * 32 + ml) * 8 + 4 * kl;
                        *reinterpret_cast<h4*>(dst) = hi;
                        *reinterpret_cast<h4*>(dst + h1_plane) = lo;
                    }
        }
    }
    GH_STAMP(34);
    __syncthreads();
    GH_STAMP(35);

    // ---- phase 2: h2 rows [64 w, 64 w + 64) x 64 pixels per wave; B resident in LDS, A two k-steps ahead from L2
    for (int o_base = wid * 64; o_base < M; o_base += 512) {
        f32x16_t accm[2][2], accx[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(b2 + o_base + i * 32 + 8 * gq + 4 * kl);
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int t = 0; t < 4; ++t) { accm[i][jn][4 * gq + t] = b4[t]; accx[i][jn][4 * gq + t] = 0.f; }
            }
        const long w2_plane = (long)M * M;               // K = M
        const _Float16* ap = W2 + ((long)kl * M + o_base + ml) * 8;
        const long sstep = (long)2 * M * 8;
        auto loadA = [&](int s, h8 (&a)[4]) {
            const _Float16* p = ap + (long)s * sstep;
            a[0] = *reinterpret_cast<const h8*>(p);
            a[1] = *reinterpret_cast<const h8*>(p + 32 * 8);
            a[2] = *reinterpret_cast<const h8*>(p + w2_plane);
            a[3] = *reinterpret_cast<const h8*>(p + w2_plane + 32 * 8);
        };
        const int nst = M >> 4;
        h8 A0[4], A1[4], A2[4];   // (four steps in flight measured slower than two)
        loadA(0, A0);
        if (nst > 1) loadA(1, A1);
        const _Float16* bp = h1 + ((long)kl * 64 + ml) * 8;
#pragma unroll 1
        for (int s = 0; s < nst; ++s) {
            if (s + 2 < nst) loadA(s + 2, A2);
            const _Float16* bs = bp + (long)s * 2 * 64 * 8;
            h8 bh[2], bl[2];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                bh[jn] = *reinterpret_cast<const h8*>(bs + jn * 256);
                bl[jn] = *reinterpret_cast<const h8*>(bs + jn * 256 + h1_plane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    accm[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[i], bh[jn], accm[i][jn], 0, 0, 0);
                    accx[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[i], bl[jn], accx[i][jn], 0, 0, 0);
                    accx[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[2 + i], bh[jn], accx[i][jn], 0, 0, 0);
                }
#pragma unroll
            for (int t = 0; t < 4; ++t) { A0[t] = A1[t]; A1[t] = A2[t]; }
        }
        GH_STAMP(36);
        // epilogue: relu, split, 8-byte stores into the tile-major SH tensor
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const long px = gp0 + jn * 32 + ml;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int o0 = o_base + i * 32 + 8 * gq + 4 * kl;
                    h4 hi, lo;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float v = relu_(accm[i][jn][4 * gq + t] + accx[i][jn][4 * gq + t] * SH_LO_INV);
                        _Float16 a, b;
                        sh_split(v, a, b);
                        hi[t] = a; lo[t] = b;
                    }
                    _Float16* dst = Ysh + sh_off(NCK, 0, o0 >> 3, px) + (o0 & 7);
                    *reinterpret_cast<h4*>(dst) = hi;
                    *reinterpret_cast<h4*>(dst + (long)NCK * SH_CHUNK_STEP) = lo;
                }
        }
        GH_STAMP(37);
    }
}

bool f02_sh_supported(int Cin, int H, int W, int hidden) {
    if (!first_sh_supported(Cin, H, W, hidden) || !gemm_sh_supported(hidden, hidden, H, W)) return false;
    if (hidden % 64 != 0 || hidden > 512) return false;      // h1 tile = hidden * 256 bytes of LDS; waves take channel-group pairs
    const int R = 64 / W;
    if (R < 1 || H % R != 0) return false;
    const size_t lds = ((size_t)2 * ((Cin + 7) / 8) * (R + 2) * (W + 2) * 8 + (size_t)2 * hidden * 64) * sizeof(_Float16);
    return lds <= 160 * 1024;      // window planes + the h1 tile
}

// w0: first_sh image (REPACK_SH_FIRST), w2: gemm_sh image (REPACK_SH_GEMM)
int launch_f02_sh(const float* x, long x_bs, const void* w0, const void* w2, _Float16* y_sh, int N, int Cin, int H, int W,
                  int hidden, hipStream_t s) {
    GH_REQUIRE(f02_sh_supported(Cin, H, W, hidden), "f02_sh: unsupported shape");
    if (N == 0) return GLOWHIP_OK;
    const int nchunk = (Cin + 7) / 8, G = (9 * nchunk + 1) & ~1;
    const int wshift = W == 64 ? 6 : (W == 32 ? 5 : (W == 16 ? 4 : 3));
    const int R = 64 / W;
    const float* b0 = (const float*)((const char*)w0 + align_up((size_t)2 * G * hidden * 8 * sizeof(_Float16), 16));
    const float* b2 = (const float*)((const char*)w2 + align_up((size_t)2 * hidden * hidden * sizeof(_Float16), 16));
    const size_t lds = ((size_t)2 * nchunk * (R + 2) * (W + 2) * 8 + (size_t)2 * hidden * 64) * sizeof(_Float16);
    const unsigned grid = (unsigned)((long)N * H * W / 64);
    (void)hipFuncSetAttribute((const void*)k_f02_sh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_f02_sh, dim3(grid), dim3(512), lds, s, x, x_bs, (const _Float16*)w0, b0, (const _Float16*)w2, b2,
                       y_sh, N, Cin, H, W, hidden, wshift);
    GH_LAUNCH_CHECK("k_f02_sh");
    return GLOWHIP_OK;
}

}  // namespace glowhip
