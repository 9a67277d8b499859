// Microbenchmark: issue rate of v_mfma_f32_32x32x16_f16 with 1 or 2 waves per SIMD, 8 or 4 independent accumulators.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters) {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 0.001f); b[q] = (_Float16)(q * 0.5f); }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run(int threads, int blocks, int iters) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, 8 * blocks * (threads / 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = 3.0 * NACC * iters;                       // MFMAs per wave
    const double flops = nm * 32768.0 * blocks * (threads / 64);
    printf("NACC=%d threads=%d blocks=%d: %.3f ms, %.1f TFLOP/s; wave0 %.1f counter ticks per MFMA (x waves/SIMD = %d); ticks/us %.0f\n", NACC, threads, blocks, ms,
           flops / ms / 1e9, h[0] / nm, threads / 256, h[0] / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<8>(256, 256, 2000);
    run<8>(512, 256, 2000);
    run<4>(512, 256, 2000);
    run<8>(256, 512, 2000);
    run<8>(512, 512, 1000);
    return 0;
}
