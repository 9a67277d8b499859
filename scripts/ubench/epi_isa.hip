// epi_isa.hip -- what hipcc emits for the relu + split-half epilogue variants (compile with -S; not a benchmark)
#include <hip/hip_runtime.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

extern "C" __global__ void epi_old(const f32x4_t* acc, const f32x4_t* rs, const f32x4_t* bb, h4* ohi, h4* olo) {
    const int t = threadIdx.x;
    f32x4_t a = acc[t], r = rs[t], b = bb[t];
    h4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = fmaf(a[k], r[k], b[k]);
        v = v < 0.f ? 0.f : v;
        _Float16 x0 = (_Float16)v;
        _Float16 x1 = (_Float16)(v - (float)x0);
        hi[k] = x0; lo[k] = x1;
    }
    ohi[t] = hi; olo[t] = lo;
}

__device__ __forceinline__ float relu_i(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

extern "C" __global__ void epi_new(const f32x4_t* acc, const f32x4_t* rs, const f32x4_t* bb, h4* ohi, h4* olo) {
    const int t = threadIdx.x;
    f32x4_t a = acc[t], r = rs[t], b = bb[t];
    h4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
        float v0 = relu_i(fmaf(a[k], r[k], b[k]));
        float v1 = relu_i(fmaf(a[k + 1], r[k + 1], b[k + 1]));
        f32x2_t vv = {v0, v1};
        h2 x = __builtin_convertvector(vv, h2);
        float r0 = fmaf((float)x[0], -1.0f, v0);
        float r1 = fmaf((float)x[1], -1.0f, v1);
        f32x2_t rr = {r0, r1};
        h2 y = __builtin_convertvector(rr, h2);
        hi[k] = x[0]; hi[k + 1] = x[1]; lo[k] = y[0]; lo[k + 1] = y[1];
    }
    ohi[t] = hi; olo[t] = lo;
}
