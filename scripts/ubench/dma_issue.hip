// Microbenchmark: what does it cost a wave that streams v_mfma_f32_32x32x16_f16 (one wave per SIMD, 4 waves per CU) to ALSO feed
// an LDS ring -- per "quad" of 12 MFMAs + 8 ds_read_b128:
//   V0 nothing | V1 2 x global_load_lds_dwordx4 | V2 2 x raw_buffer_load_lds (16 B) | V3 2 x global_load_dwordx4 -> VGPR + 2 x ds_write_b128
//   V4 4 x global_load_lds_dwordx4 | V5 2 x global_load_lds_dword (4 B per lane) | V6 2 x global_load_dwordx4 -> VGPR only (consumed by a v_or)
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

template <int V>
__global__ void __launch_bounds__(256) k(const char* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int e = tid; e < 96 * 1024 / 16; e += 256) reinterpret_cast<f4*>(lds)[e] = f4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    h8 bh, bl;
    for (int q = 0; q < 8; ++q) { bh[q] = (_Float16)(lane * 0.001f); bl[q] = (_Float16)(q * 0.5f); }
    const char* gp = src + (size_t)(blockIdx.x & 7) * (1 << 20) + lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    f4 stage[2] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    float sink = 0.f;
    int slot = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    h8 A[2][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { A[0][i][0] = *reinterpret_cast<const h8*>(lds + lane * 16 + i * 512); A[0][i][1] = *reinterpret_cast<const h8*>(lds + lane * 16 + i * 512 + 16384); }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            slot = slot + 2048 >= 16384 ? 0 : slot + 2048;
            const char* ap = lds + slot + lane * 16;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { A[u ^ 1][i][0] = *reinterpret_cast<const h8*>(ap + i * 512); A[u ^ 1][i][1] = *reinterpret_cast<const h8*>(ap + i * 512 + 16384); }
            const int off = ((it + u) & 255) * 4096 + wid * 2048;
            char* dst = lds + 32768 + slot * 2 + wid * 2048;
            if (V == 1 || V == 4) { dma16(gp + off, dst); dma16(gp + off + 1024, dst + 1024); }
            if (V == 4) { dma16(gp + off + 65536, dst + 8192); dma16(gp + off + 65536 + 1024, dst + 8192 + 1024); }
            if (V == 5) { dma4(gp + off, dst); dma4(gp + off + 1024, dst + 1024); }
            if (V == 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, off, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 1024), 16, lane * 16, off + 1024, 0, 0);
            }
            if (V == 3) {
                *reinterpret_cast<f4*>(dst + lane * 16) = stage[0];
                *reinterpret_cast<f4*>(dst + 1024 + lane * 16) = stage[1];
                stage[0] = *reinterpret_cast<const f4*>(gp + off);
                stage[1] = *reinterpret_cast<const f4*>(gp + off + 1024);
            }
            if (V == 6) {
                sink += stage[0][0] + stage[1][0];
                stage[0] = *reinterpret_cast<const f4*>(gp + off);
                stage[1] = *reinterpret_cast<const f4*>(gp + off + 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u][i][0], bh, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u][i][0], bl, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u][i][1], bh, acc[i], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = sink;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s + stage[0][1] + stage[1][1];
    if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
}

template <int V>
void run(const char* name, const char* src, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    k<V><<<256, 256, 96 * 1024>>>(src, out, cyc, 10);
    hipDeviceSynchronize();
    k<V><<<256, 256, 96 * 1024>>>(src, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 1024; ++i) sum += (double)h[i];
    printf("%-70s %.1f cycles per quad of 12 MFMAs (384 at the full rate)\n", name, sum / 1024 / iters);
}

int main() {
    char* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 16 << 20); hipMemset(src, 0, 16 << 20);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<0>("V0 MFMAs + 8 ds_read_b128", src, out, cyc);
    run<1>("V1 + 2 x global_load_lds_dwordx4", src, out, cyc);
    run<2>("V2 + 2 x buffer_load_dwordx4 ... lds", src, out, cyc);
    run<3>("V3 + 2 x global_load_dwordx4 -> VGPR, 2 x ds_write_b128", src, out, cyc);
    run<4>("V4 + 4 x global_load_lds_dwordx4", src, out, cyc);
    run<5>("V5 + 2 x global_load_lds_dword", src, out, cyc);
    run<6>("V6 + 2 x global_load_dwordx4 -> VGPR only", src, out, cyc);
    return 0;
}
