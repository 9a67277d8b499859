// lu.hip -- in-kernel LU of the invertible 1x1 convolution weight (replaces torch.det and
// Tensor.inverse at network/module.py:357,365).  One workgroup per matrix runs Gauss-Jordan
// elimination with partial pivoting in fp64 on the augmented matrix [W | I] held in LDS (C <= 64)
// or in an L2-resident scratch buffer (larger C): log|det W| = sum log|pivot|, W^-1 = right half.
#include "kernels.h"

namespace glowhip {

size_t invconv_scratch_bytes(int C) { return (size_t)C * 2 * C * sizeof(double); }

constexpr int LU_LDS_MAX_C = 64;  // 64 * 128 * 8 B = 64 KiB

template <bool USE_LDS>
__global__ void __launch_bounds__(256) k_invconv_prepare(const float* __restrict__ w, int C, float* __restrict__ winv,
                                                         float* __restrict__ logabsdet, double* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) double lds_aug[];
    __shared__ int s_piv;
    __shared__ double s_logdet;
    double* A = USE_LDS ? lds_aug : scratch;
    const int tid = threadIdx.x, W2 = 2 * C;
    for (int e = tid; e < C * W2; e += 256) {
        int r = e / W2, c = e - r * W2;
        A[e] = (c < C) ? (double)w[r * C + c] : ((c - C == r) ? 1.0 : 0.0);
    }
    if (tid == 0) s_logdet = 0.0;
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        // partial pivoting: first wave finds argmax |A[r][k]|, r >= k (ties -> lowest row: deterministic)
        if (tid < 64) {
            double best = -1.0;
            int bi = k;
            for (int r = k + tid; r < C; r += 64) {
                double v = fabs(A[r * W2 + k]);
                if (v > best) { best = v; bi = r; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                double ov = __shfl_down(best, o, 64);
                int oi = __shfl_down(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (tid == 0) s_piv = bi;
        }
        __syncthreads();
        const int pr = s_piv;
        if (pr != k) {
            for (int c = tid; c < W2; c += 256) {
                double t = A[k * W2 + c];
                A[k * W2 + c] = A[pr * W2 + c];
                A[pr * W2 + c] = t;
            }
        }
        __syncthreads();
        const double piv = A[k * W2 + k];
        if (tid == 0) s_logdet += log(fabs(piv));
        __syncthreads();  // everyone has read piv before the row is scaled
        const double inv = 1.0 / piv;
        for (int c = tid; c < W2; c += 256) A[k * W2 + c] *= inv;
        __syncthreads();
        // eliminate column k from every other row; the column's multipliers are read before any write
        // to column k because each (r, c) element is owned by exactly one thread and c==k is written last
        for (int e = tid; e < C * W2; e += 256) {
            int r = e / W2, c = e - r * W2;
            if (r == k || c == k) continue;
            A[e] = fma(-A[r * W2 + k], A[k * W2 + c], A[e]);
        }
        __syncthreads();
        for (int r = tid; r < C; r += 256)
            if (r != k) A[r * W2 + k] = 0.0;
        __syncthreads();
    }
    if (winv)
        for (int e = tid; e < C * C; e += 256) {
            int r = e / C, c = e - r * C;
            winv[e] = (float)A[r * W2 + C + c];
        }
    if (tid == 0 && logabsdet) logabsdet[0] = (float)s_logdet;
}

int launch_invconv_prepare(const float* w, int C, float* winv, float* logabsdet, void* scratch, hipStream_t s) {
    GH_REQUIRE(C > 0 && C <= 1024, "invconv_prepare: C=%d unsupported", C);
    if (C <= LU_LDS_MAX_C) {
        size_t lds = invconv_scratch_bytes(C);
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)k_invconv_prepare<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        hipLaunchKernelGGL(k_invconv_prepare<true>, dim3(1), dim3(256), lds, s, w, C, winv, logabsdet, (double*)nullptr);
    } else {
        GH_REQUIRE(scratch != nullptr, "invconv_prepare: scratch buffer required for C=%d", C);
        hipLaunchKernelGGL(k_invconv_prepare<false>, dim3(1), dim3(256), 0, s, w, C, winv, logabsdet, (double*)scratch);
    }
    GH_LAUNCH_CHECK("k_invconv_prepare");
    return GLOWHIP_OK;
}

}  // namespace glowhip
