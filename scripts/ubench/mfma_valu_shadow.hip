// Microbenchmark: how many independent VALU instructions issue in the shadow of a v_mfma_f32_32x32x16_f16 (32 clocks on the
// matrix pipe), (a) from the SAME wave, interleaved NV per MFMA, (b) from the OTHER wave of the SIMD (one wave only MFMAs, its
// partner only VALU).  hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_shadow.hip -o /tmp/mvs && /tmp/mvs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// KIND 0: v_fma_f32   1: v_pk_mul_f32   2: v_cvt_pk_f16_f32
template <int KIND>
__device__ __forceinline__ void valu(float& x, f2& p, float c) {
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(x) : "v"(c));
}

// MODE 0: every wave interleaves NV VALU per MFMA.  MODE 1: waves 0-3 of the workgroup MFMAs only, waves 4-7 VALU only (8 per "slot").
template <int NV, int KIND, int MODE>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters) {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 0.001f); b[q] = (_Float16)(q * 0.5f); }
    float x[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.5f + i; p[i] = f2{x[i], 1.0f}; }
    const float c = 1.0001f;
    const bool mf = MODE == 0 || threadIdx.x < 256, va = MODE == 0 || threadIdx.x >= 256;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                MFMA(acc[i]);
#pragma unroll
                for (int v = 0; v < NV; ++v) valu<KIND>(x[(i * NV + v) & 7], p[(i * NV + v) & 7], c);
            }
        }
    } else if (mf) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) MFMA(acc[i]);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 32; ++v) valu<KIND>(x[v & 7], p[v & 7], c);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    (void)va;
}

template <int NV, int KIND, int MODE>
void run(int threads, int iters) {
    const int blocks = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, 8 * blocks * (threads / 64));
    k<NV, KIND, MODE><<<blocks, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    k<NV, KIND, MODE><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(unsigned long long) * (threads / 64), hipMemcpyDeviceToHost);
    const char* kn = KIND == 0 ? "v_fma_f32" : KIND == 1 ? "v_pk_mul_f32" : "v_cvt_pk_f16_f32";
    if (MODE == 0)
        printf("same wave, %d waves/SIMD, %2d x %-16s per MFMA: %.1f ticks per MFMA (+VALU group)\n", threads / 256, NV, kn, (double)h[0] / (4.0 * iters));
    else
        printf("split waves (%s): MFMA wave %.1f ticks per MFMA; VALU wave %.2f ticks per VALU instruction\n", kn, (double)h[0] / (4.0 * iters),
               (double)h[4] / (32.0 * iters));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0, 0>(256, 4000);
    run<2, 0, 0>(256, 4000); run<4, 0, 0>(256, 4000); run<6, 0, 0>(256, 4000); run<8, 0, 0>(256, 4000); run<12, 0, 0>(256, 4000);
    run<4, 1, 0>(256, 4000); run<8, 1, 0>(256, 4000); run<4, 2, 0>(256, 4000); run<8, 2, 0>(256, 4000);
    run<0, 0, 0>(512, 4000); run<4, 0, 0>(512, 4000); run<8, 0, 0>(512, 4000); run<4, 1, 0>(512, 4000); run<8, 1, 0>(512, 4000);
    run<0, 0, 1>(512, 4000); run<0, 1, 1>(512, 4000); run<0, 2, 1>(512, 4000);
    return 0;
}
