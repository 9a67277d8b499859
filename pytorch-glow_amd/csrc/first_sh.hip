// first_sh.hip -- f.0 (3x3 convolution C/2 -> hidden + ActNorm + ReLU, network/module.py:252-259,314-315) on split-half
// operands (sh.h): fp32 z in, h1 out as an SH tensor.  The kernel is bound by writing h1 (134 MB per level-1 layer), so
// the design goal is simply: few instructions per output, wide coalesced stores, several workgroups per CU.
//
// Workgroup = R image rows (32*NT pixels) x all `hidden` output channels.  The input window (rows y0-1..y0+R, one zero
// column left and right, channels padded to 8) is split into (hi, lo) halves ONCE and kept in LDS as
// [plane][8-channel chunk][window pixel][8]; k runs over (tap, chunk) groups of 8, so a B fragment (8 consecutive k of one
// pixel) is one 16-byte LDS read at a tap-shifted window address -- im2col never exists.  The weights (ActNorm scale
// folded in, [plane][group][hidden][8]) are read straight from L2 as A fragments: consecutive lanes = consecutive output
// channels = contiguous 16-byte groups.  A wave owns one 32-channel M-tile x all NT pixel tiles; blockIdx.y walks the
// hidden/128 channel passes (the window is rebuilt per pass: 200 slots, cheaper than the lost parallelism).
#include "sh.h"
#include "conv_mfma.h"

namespace glowhip {

template <int NT, int R>
__global__ void __launch_bounds__(256)
k_first_sh(const float* __restrict__ X, long x_bs, const _Float16* __restrict__ Wsh, const float* __restrict__ bias,
           _Float16* __restrict__ Ysh, int N, int Cin, int H, int W, int M, int wshift, int relu) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_f[];
    const int HW = H * W, WP = W + 2, Wpx = (R + 2) * WP;
    const int nchunk = (Cin + 7) >> 3;
    const int G = (9 * nchunk + 1) & ~1;              // 8-wide k groups, padded to an even count (k16 steps)
    const int steps = G >> 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, ml = lane & 31;
    const int bpi = H / R;
    const long n = blockIdx.x / bpi;
    const int y0 = (int)(blockIdx.x - n * bpi) * R;
    const long plane_h = (long)nchunk * Wpx * 8;      // halfs per LDS plane

    // ---- window: fp32 -> (hi, lo), one (chunk, window pixel) slot = 8 channels = 16 B per plane
    const float* xin = X + n * x_bs;
    for (int e = tid; e < nchunk * Wpx; e += 256) {
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
        *reinterpret_cast<h8*>(smem_f + (long)e * 8) = hi;
        *reinterpret_cast<h8*>(smem_f + plane_h + (long)e * 8) = lo;
    }
    __syncthreads();

    // window offset (halfs) of this lane's pixel in each of its NT pixel tiles
    int pbase[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int q = u * 32 + ml;
        pbase[u] = ((q >> wshift) * WP + (q & (W - 1))) * 8;
    }
    const long w_plane = (long)G * M * 8;

    {
        const int o_tile = blockIdx.y * 128 + wid * 32;      // this wave's 32 output channels (blockIdx.y = 128-channel pass)
        f32x16_t accm[NT], accx[NT];   // main accumulators start at the folded ActNorm bias
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bias + o_tile + 8 * gq + 4 * kl);
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) { accm[u][4 * gq + q] = b4[q]; accx[u][4 * gq + q] = 0.f; }
        }
        const _Float16* ap = Wsh + ((long)kl * M + o_tile + ml) * 8;
        const long sstep = (long)2 * M * 8;
        // A fragments come straight from L2: keep two k-steps in flight
        h8 a0h = *reinterpret_cast<const h8*>(ap), a0l = *reinterpret_cast<const h8*>(ap + w_plane);
        h8 a1h = a0h, a1l = a0l;
        if (steps > 1) {
            a1h = *reinterpret_cast<const h8*>(ap + sstep);
            a1l = *reinterpret_cast<const h8*>(ap + sstep + w_plane);
        }
#pragma unroll 1
        for (int s = 0; s < steps; ++s) {
            h8 a2h = a1h, a2l = a1l;
            if (s + 2 < steps) {
                a2h = *reinterpret_cast<const h8*>(ap + (long)(s + 2) * sstep);
                a2l = *reinterpret_cast<const h8*>(ap + (long)(s + 2) * sstep + w_plane);
            }
            int g = 2 * s + kl;
            g = g < 9 * nchunk ? g : 0;                      // padded groups carry zero weights: any finite B will do
            const int tap = g / nchunk, ch = g - tap * nchunk;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int goff = (ch * Wpx + dy * WP + dx) * 8;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const h8 bh = *reinterpret_cast<const h8*>(smem_f + goff + pbase[u]);
                const h8 bl = *reinterpret_cast<const h8*>(smem_f + plane_h + goff + pbase[u]);
                accm[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, accm[u], 0, 0, 0);
                accx[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, accx[u], 0, 0, 0);
                accx[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, accx[u], 0, 0, 0);
            }
            a0h = a1h; a0l = a1l; a1h = a2h; a1l = a2l;
        }
        // epilogue: + folded bias, ReLU, split, 8-byte stores (a half-wave's 32 lanes x 2 = 512 contiguous bytes)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const long px = n * HW + (long)y0 * W + u * 32 + ml;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int o0 = o_tile + 8 * gq + 4 * kl;
                h4 hi, lo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = accm[u][4 * gq + q] + accx[u][4 * gq + q] * SH_LO_INV;
                    const float v = relu ? relu_(t) : t;
                    _Float16 a, b;
                    sh_split(v, a, b);
                    hi[q] = a; lo[q] = b;
                }
                _Float16* dst = Ysh + sh_off(M >> 3, 0, o0 >> 3, px) + (o0 & 7);
                *reinterpret_cast<h4*>(dst) = hi;
                *reinterpret_cast<h4*>(dst + (long)(M >> 3) * SH_CHUNK_STEP) = lo;
            }
        }
    }
}

static int first_sh_rows(int W) { return W == 64 ? 1 : (W == 32 ? 2 : 4); }   // 64 / 64 / 64 / 32 pixels per workgroup
static int first_sh_groups(int Cin) { return (9 * ((Cin + 7) / 8) + 1) & ~1; }

bool first_sh_supported(int Cin, int H, int W, int Cout) {
    if (Cout % 128 != 0 || Cin < 1 || Cin > 128) return false;      // (Cin = 96: config E's 8x8 level, 47 us on the fp32 kernel)
    if (W != 8 && W != 16 && W != 32 && W != 64) return false;
    if ((size_t)2 * ((Cin + 7) / 8) * (first_sh_rows(W) + 2) * (W + 2) * 8 * sizeof(_Float16) > 64 * 1024) return false;   // window planes in LDS
    return H % first_sh_rows(W) == 0 && (H * W) % 64 == 0;
}

size_t first_sh_packed_bytes(int Cin, int Cout) {
    return align_up((size_t)2 * first_sh_groups(Cin) * Cout * 8 * sizeof(_Float16), 16) + (size_t)Cout * sizeof(float);
}

int launch_first_sh(const float* x, long x_bs, const void* wsh, _Float16* y_sh, int N, int Cin, int H, int W, int Cout,
                    int relu, hipStream_t s) {
    GH_REQUIRE(first_sh_supported(Cin, H, W, Cout), "first_sh: unsupported shape");
    if (N == 0) return GLOWHIP_OK;
    const int G = first_sh_groups(Cin), nchunk = (Cin + 7) / 8;
    const _Float16* w = (const _Float16*)wsh;
    const float* bias = (const float*)((const char*)wsh + align_up((size_t)2 * G * Cout * 8 * sizeof(_Float16), 16));
    const int R = first_sh_rows(W);
    const size_t lds = (size_t)2 * nchunk * (R + 2) * (W + 2) * 8 * sizeof(_Float16);
    const unsigned grid = (unsigned)(N * (H / R));
    const int wshift = W == 64 ? 6 : (W == 32 ? 5 : (W == 16 ? 4 : 3));
#define GH_FSH_CASE(nt, r)                                                                                            \
    if (R == r && r * W == 32 * nt) {                                                                                           \
        if (lds > 32 * 1024) (void)hipFuncSetAttribute((const void*)k_first_sh<nt, r>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_first_sh<nt, r>), dim3(grid, Cout / 128), dim3(256), lds, s, x, x_bs, w, bias, y_sh, N, Cin, H, W, \
                           Cout, wshift, relu);                                                                       \
        GH_LAUNCH_CHECK("k_first_sh");                                                                                \
        return GLOWHIP_OK;                                                                                            \
    }
    GH_FSH_CASE(2, 1) GH_FSH_CASE(2, 2) GH_FSH_CASE(2, 4) GH_FSH_CASE(1, 4)
#undef GH_FSH_CASE
    set_error("first_sh: no kernel instance");
    return GLOWHIP_EINVAL;
}

}  // namespace glowhip
