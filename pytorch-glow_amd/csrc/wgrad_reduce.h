// wgrad_reduce.h -- the split-K reduction of a weight-gradient GEMM as a device function (wgrad_mfma.hip's reduction kernels and
// backward.hip's k_chanmix_bwd_reduce, which runs a FlowStep's three reductions beside its mixer backward).
#pragma once
#include "backward.h"

namespace glowhip {

// dW[f(m, n)] = sum_split partial[split][m][n], splits added in order.  `block`: the job's block index (256 elements of 4 columns).
//   mode 0: dW[m*Nreal + n]                      (f.2: [512][512];  f.0: [512][Ch*9] = dW0[o][i][tap] flat)
//   mode 1: m = o*9 + tap, n = i: dW[(o*Nreal + i)*9 + tap]      (f.4: dW4[o][i][tap])
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, float* __restrict__ dw, int splits,
                                                  int Mpad, int Npad, int Mreal, int Nreal, int mode, long block) {
    typedef float f32x4r __attribute__((ext_vector_type(4)));
    // four consecutive columns per thread (Npad % 64 == 0: 16-byte loads), eight splits' loads in flight, added in split order
    const int n4 = (Nreal + 3) >> 2;
    const long e = block * 256 + threadIdx.x;
    if (e >= (long)Mreal * n4) return;
    const int m = (int)(e / n4), n0 = (int)(e - (long)m * n4) * 4;
    f32x4r s = {0.f, 0.f, 0.f, 0.f};
    const float* src = partial + (long)m * Npad + n0;
    const long stride = (long)Mpad * Npad;
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
        f32x4r v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4r*>(src + (k + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += *reinterpret_cast<const f32x4r*>(src + k * stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        if (n >= Nreal) break;
        if (mode == 0) dw[(long)m * Nreal + n] = s[j];
        else {
            const int o = m / 9, tap = m - o * 9;
            dw[((long)o * Nreal + n) * 9 + tap] = s[j];
        }
    }
}

}  // namespace glowhip
