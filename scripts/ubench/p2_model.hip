// Microbenchmark: the inner loop of k_cnet's P2 in isolation -- per k-step 24 MFMAs (8 accumulators x 3), optionally 4 global
// 16-byte loads (A operand, L2-resident image shared by every CU, requested two k-steps ahead) and 8 LDS 16-byte reads (B operand).
// Which of the operand streams costs MFMA issue slots?
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/p2_model.hip -o scripts/ubench/p2_model.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <bool GL, bool DS, int WAVES, int AHEAD = 2, bool DUMMY = false, bool SPREAD = false, bool DMA = false>
__global__ void __launch_bounds__(WAVES * 64) k(const char* img, float* out, int ksteps, int reps, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dma = lds + 32768 + (threadIdx.x >> 6) * 12288;   // per-wave ring of three 4 KB slots (overlaps the B area: timing only)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kl = lane >> 5, ml = lane & 31;
    for (int e = threadIdx.x; e < 128 * 1024 / 4; e += WAVES * 64) reinterpret_cast<float*>(lds)[e] = 0.001f * e;
    __syncthreads();
    f16v acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long plane = (long)ksteps * 2 * 512 * 16;
    const char* ap = img + ((long)kl * 512 + (wid % 8) * 64 + ml) * 16;
    const char* bp = lds + (kl * 128 + ml) * 16;
    h8 A[5][4];
    auto loadA = [&](int s, h8 (&d)[4]) {
        const char* p = ap + (long)s * (2 * 512 * 16);
        d[0] = *reinterpret_cast<const h8*>(p); d[1] = *reinterpret_cast<const h8*>(p + 512);
        d[2] = *reinterpret_cast<const h8*>(p + plane); d[3] = *reinterpret_cast<const h8*>(p + 512 + plane);
    };
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 8; ++q) A[i][j][q] = (_Float16)(0.01f * (lane + q + i + j));
    h8 bh[4], bl[4], K0[4];
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    h8 sink = A[0][0];
    for (int j = 0; j < 4; ++j) K0[j] = A[2][j];
    for (int j = 0; j < 4; ++j) { bh[j] = A[0][j]; bl[j] = A[1][j]; }
    auto kstep = [&](int s, const h8 (&use)[4], h8 (&fill)[4]) {
        const char* lp = ap + (long)min(s + AHEAD, ksteps - 1) * (2 * 512 * 16);
        if (GL && DMA) {
            // request k-step s + 2 into ring slot (s + 2) % 3 of this wave's private LDS area, then wait for k-step s and read it
            const char* lp2 = ap - ((long)kl * 512 + ml) * 16 + (long)min(s + 2, ksteps - 1) * (2 * 512 * 16);   // wave base (lane offset added by the DMA)
            char* slot = dma + ((s + 2) % 3) * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const char* src = lp2 + (q & 1) * 512 + (q >> 1) * plane + ((long)kl * 512 + ml) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70 | 8);      // vmcnt(8): the four loads of k-step s have landed (two younger groups in flight)
            const char* rs = dma + (s % 3) * 4096 + lane * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) fill[q] = *reinterpret_cast<const h8*>(rs + q * 1024);
        }
        if (GL && !SPREAD && !DMA) { loadA(min(s + AHEAD, ksteps - 1), fill); __builtin_amdgcn_sched_barrier(0); }
        if (DS) {
            const char* bs = bp + (long)(s & 15) * (2 * 128 * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = *reinterpret_cast<const h8*>(bs + j * 512);
                bl[j] = *reinterpret_cast<const h8*>(bs + j * 512 + 65536);
            }
        }
        if (GL && SPREAD) { fill[0] = *reinterpret_cast<const h8*>(lp); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(DMA ? fill[0] : use[0], bh[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(DMA ? fill[1] : use[1], bh[j], acc[4 + j], 0, 0, 0);
        if (GL && SPREAD) { __builtin_amdgcn_sched_barrier(0); fill[1] = *reinterpret_cast<const h8*>(lp + 512); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(DMA ? fill[0] : use[0], bl[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(DMA ? fill[1] : use[1], bl[j], acc[4 + j], 0, 0, 0);
        if (GL && SPREAD) { __builtin_amdgcn_sched_barrier(0); fill[2] = *reinterpret_cast<const h8*>(lp + plane); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(DMA ? fill[2] : use[2], bh[j], acc[j], 0, 0, 0);
        if (GL && SPREAD) { __builtin_amdgcn_sched_barrier(0); fill[3] = *reinterpret_cast<const h8*>(lp + 512 + plane); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(DMA ? fill[3] : use[3], bh[j], acc[4 + j], 0, 0, 0);
    };
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (AHEAD == 2) {
            if (GL && !DMA) { loadA(0, A[0]); loadA(1, A[1]); }
            if (GL && DMA) {
                for (int g2 = 0; g2 < 2; ++g2)
                    for (int q = 0; q < 4; ++q) {
                        const char* src = ap + (long)g2 * (2 * 512 * 16) + (q & 1) * 512 + (q >> 1) * plane;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(dma + g2 * 4096 + q * 1024), 16, 0, 0);
                    }
            }
#pragma unroll 1
            for (int s = 0; s + 3 <= ksteps; s += 3) {
                kstep(s, A[0], A[2]);
                kstep(s + 1, A[1], A[0]);
                kstep(s + 2, A[2], A[1]);
            }
        } else if (AHEAD == 4) {
            if (GL) { loadA(0, A[0]); loadA(1, A[1]); loadA(2, A[2]); loadA(3, A[3]); }
#pragma unroll 1
            for (int s = 0; s + 5 <= ksteps; s += 5) {
                kstep(s, A[0], A[4]);
                kstep(s + 1, A[1], A[0]);
                kstep(s + 2, A[2], A[1]);
                kstep(s + 3, A[3], A[2]);
                kstep(s + 4, A[4], A[3]);
            }
        } else {
            if (GL) { loadA(0, A[0]); }
#pragma unroll 1
            for (int s = 0; s + 2 <= ksteps; s += 2) {
                kstep(s, A[0], A[1]);
                kstep(s + 1, A[1], A[0]);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = (float)sink[0] + (float)sink[3];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wid] = t1 - t0;
}

template <bool GL, bool DS, int WAVES, int AHEAD = 2, bool DUMMY = false, bool SPREAD = false, bool DMA = false>
void run(const char* img) {
    float* out; unsigned long long* cyc;
    const int blocks = 256;
    hipMalloc(&out, 4 * blocks * WAVES * 64); hipMalloc(&cyc, 64);
    const int ksteps = 30, reps = 20;
    hipFuncSetAttribute((const void*)k<GL, DS, WAVES, AHEAD, DUMMY, SPREAD, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<GL, DS, WAVES, AHEAD, DUMMY, SPREAD, DMA><<<blocks, WAVES * 64, 128 * 1024>>>(img, out, ksteps, 2, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<GL, DS, WAVES, AHEAD, DUMMY, SPREAD, DMA><<<blocks, WAVES * 64, 128 * 1024>>>(img, out, ksteps, reps, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    const double nm = 24.0 * ksteps * reps;       // MFMAs per wave
    const double ideal = nm * 32 * (WAVES / 4);   // cycles if the pipe never idles
    unsigned long long cmax = 0; for (int i = 0; i < WAVES; ++i) cmax = c[i] > cmax ? c[i] : cmax;
    printf("dma %d spread %d dummy %d ahead %d global %d lds %d waves %d: %.3f ms, %.0f TFLOP/s, MFMA pipe busy %.2f (slowest wave), first wave %.1f / last %.1f ticks per own MFMA\n", (int)DMA, (int)SPREAD, (int)DUMMY, AHEAD, GL, DS, WAVES, ms,
           nm * 32768.0 * blocks * WAVES / ms / 1e9, ideal / cmax, c[0] / nm, c[WAVES - 1] / nm);
    hipFree(out); hipFree(cyc);
}

int main() {
    char* img; hipMalloc(&img, 4 << 20); hipMemset(img, 0, 4 << 20);
    run<false, false, 8>(img);
    run<true, false, 8>(img);
    run<false, true, 8>(img);
    run<true, true, 8>(img);
    run<false, false, 4>(img);
    run<true, false, 4>(img);
    run<false, true, 4>(img);
    run<true, true, 4>(img);
    run<true, false, 8, 2, false, false, true>(img);
    run<true, true, 8, 2, false, false, true>(img);
    run<true, true, 4, 2, false, false, true>(img);
    run<true, false, 8, 1>(img);
    run<true, true, 8, 1>(img);
    run<true, true, 4, 1>(img);
    return 0;
}
