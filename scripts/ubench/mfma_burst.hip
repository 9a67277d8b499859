// Microbenchmark: one wave per SIMD streams v_mfma_f32_32x32x16_f16 (accumulators in AGPRs); after every G-th MFMA a burst of ND
// ds_read_b128 + NV v_fma_f32 (+ NG global_load_lds_dwordx4).  Cycles per MFMA (32 = the matrix pipe's rate).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_burst.hip -o scripts/ubench/mfma_burst.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define MF(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int G, int ND, int NV, int NG>
__global__ void __launch_bounds__(256) k(const char* src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 64 * 1024 / 16; e += 256) reinterpret_cast<f4*>(lds)[e] = f4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    f16v acc[12];
    for (int i = 0; i < 12; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(lane * 0.001f); b[q] = (_Float16)(q * 0.5f); }
    const unsigned laddr = (unsigned)(size_t)lds + lane * 16;
    f4 d[8];
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane + i;
    const float cc = 1.0001f;
    const char* gp = src + lane * 16 + wid * 4096;
    const unsigned voff = lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            MF(acc[i % 12]);
            if (i % G == G - 1) {
#pragma unroll
                for (int j = 0; j < ND; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[j & 7]) : "v"(laddr), "n"(512 * (j & 7)));
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j & 7]) : "v"(cc));
#pragma unroll
                for (int j = 0; j < NG % 10; ++j) {
                    const int go = ((it * 24 + i + j) & 63) * 16384;
                    __attribute__((address_space(3))) void* ld = (__attribute__((address_space(3))) void*)(lds + 32768 + wid * 2048 + j * 1024);
                    if (NG / 10 == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + go), ld, 16, 0, 0);
                    else if (NG / 10 == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + go + wid * 4096 + voff), ld, 16, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, ld, 16, voff, go + wid * 4096, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + (ND > i ? d[i][0] : 0.f);
    for (int i = 0; i < 12; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int G, int ND, int NV, int NG>
void run(const char* src, float* out, unsigned long long* cyc) {
    const int iters = 1000;
    k<G, ND, NV, NG><<<256, 256, 64 * 1024>>>(src, out, cyc, 10);
    (void)hipDeviceSynchronize();
    k<G, ND, NV, NG><<<256, 256, 64 * 1024>>>(src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[1024]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 1024; ++i) sum += (double)h[i];
    printf("G=%2d: burst of %d ds_read_b128 + %d v_fma + %d lds-dma  -> %.2f cycles per MFMA   (%.2f fillers per MFMA)\n", G, ND, NV, NG, sum / 1024 / iters / 24,
           (double)(ND + NV + NG) / G);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    char* src; float* out; unsigned long long* cyc;
    (void)hipMalloc(&src, 4 << 20); (void)hipMemset(src, 0, 4 << 20);
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
    run<4, 0, 0, 0>(src, out, cyc);
    run<4, 0, 0, 1>(src, out, cyc); run<4, 0, 0, 11>(src, out, cyc); run<4, 0, 0, 21>(src, out, cyc);
    run<2, 0, 0, 1>(src, out, cyc); run<2, 0, 0, 11>(src, out, cyc); run<2, 0, 0, 21>(src, out, cyc);
    run<12, 0, 0, 2>(src, out, cyc); run<12, 0, 0, 12>(src, out, cyc); run<12, 0, 0, 22>(src, out, cyc);
    run<12, 0, 0, 4>(src, out, cyc); run<12, 0, 0, 14>(src, out, cyc); run<12, 0, 0, 24>(src, out, cyc);
    run<1, 1, 2, 0>(src, out, cyc); run<1, 1, 1, 0>(src, out, cyc); run<1, 1, 4, 0>(src, out, cyc); run<2, 2, 4, 0>(src, out, cyc); run<2, 1, 2, 0>(src, out, cyc);
    return 0;
}
