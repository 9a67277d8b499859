// Microbenchmark: cycles per v_mfma_f32_32x32x16_f16 of ONE wave per SIMD (4 waves per CU, 256 CUs) when every MFMA is followed by a
// fixed set of filler instructions (hand-placed with inline asm; the accumulators live in AGPRs as in k_cnet1w).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_fillers.hip -o scripts/ubench/mfma_fillers.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MF(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define DSR(dst, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(laddr))
#define DSRA(dst, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=a"(dst) : "v"(laddr))
#define DSR64(dst, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(dst) : "v"(laddr))
#define VF(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(cc))
#define SA(x) asm volatile("s_add_i32 %0, %0, 1" : "+s"(x))
#define NOP() asm volatile("s_nop 0")

template <int V>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 64 * 1024 / 16; e += 256) reinterpret_cast<f4*>(lds)[e] = f4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    f16v acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(lane * 0.001f); b[q] = (_Float16)(q * 0.5f); }
    const unsigned laddr = (unsigned)(size_t)lds + lane * 16;      // (LDS byte address: low 32 bits of the flat pointer)
    f4 d0, d1, d2, d3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 e0, e1;
    float x0 = lane, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    const float cc = 1.0001f;
    int s0 = 0, s1 = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            MF(acc[i]);
            if (V == 1) { DSR(d0, 0); }
            if (V == 2) { DSR(d0, 0); DSR(d1, 512); }
            if (V == 3) { DSR(d0, 0); SA(s0); SA(s1); }
            if (V == 4) { VF(x0); }
            if (V == 5) { VF(x0); VF(x1); VF(x2); VF(x3); }
            if (V == 6) { NOP(); }
            if (V == 7) { SA(s0); SA(s1); }
            if (V == 8) { DSR(d0, 0); VF(x0); VF(x1); }
            if (V == 9) { DSR(d0, 0); DSR(d1, 512); VF(x0); VF(x1); VF(x2); VF(x3); }
            if (V == 10) { if ((i & 1) == 0) { DSR(d0, 0); DSR(d1, 512); } }
            if (V == 11) { if ((i & 3) == 0) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); } }
            if (V == 12) { if (i == 0) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); DSR(d0, 2048); DSR(d1, 2560); DSR(d2, 3072); DSR(d3, 3584); } }
            if (V == 13) { DSRA(d0, 0); }
            if (V == 20) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); VF(x0); VF(x1); VF(x2); VF(x3); } }
            if (V == 21) { if (i == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); } if (i == 7) { VF(x0); VF(x1); VF(x2); VF(x3); VF(x0); VF(x1); VF(x2); VF(x3); } }
            if (V == 22) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); VF(x0); VF(x1); asm volatile("s_add_u32 s20, s20, 1\n s_addc_u32 s21, s21, 0" ::: "s20", "s21"); } }
            if (V == 23) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); } else { VF(x0); } }
            if (V == 24) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); } else { VF(x0); VF(x1); } }
            if (V == 25) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); } else { VF(x0); VF(x1); VF(x2); VF(x3); } }
            if (V == 26) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); } else if ((i & 3) == 1) { DSR(d3, 1536); VF(x0); } }
            if (V == 27) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); DSR(d0, 2048); } }
            if (V == 28) { if ((i & 3) == 3) { DSR(d0, 0); DSR(d1, 512); DSR(d2, 1024); DSR(d3, 1536); DSR(d0, 2048); DSR(d1, 2560); } }
            if (V == 29) { if ((i & 1) == 1) { DSR(d0, 0); DSR(d1, 512); VF(x0); VF(x1); } }
            if (V == 14) { DSR64(e0, 0); DSR64(e1, 512); }
            if (V == 15) { DSR(d0, 0); asm volatile("s_nop 7"); }
            if (V == 16) { asm volatile("s_nop 3"); DSR(d0, 0); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = x0 + x1 + x2 + x3 + s0 + s1;
    if (V == 1 || V == 2 || V == 3 || V == 8 || V == 9 || V >= 10) s += d0[0];
    if (V >= 10 && V != 13 && V != 14) s += d1[0] + d2[0] + d3[0];
    if (V == 14) s += e0[0] + e1[0];
    if (V == 2 || V == 9) s += d1[0];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    (void)d2; (void)d3; (void)e0; (void)e1;
}

template <int V>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    k<V><<<256, 256, 64 * 1024>>>(out, cyc, 10);
    (void)hipDeviceSynchronize();
    k<V><<<256, 256, 64 * 1024>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[1024]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 1024; ++i) sum += (double)h[i];
    printf("%-60s %.2f cycles per MFMA\n", name, sum / 1024 / iters / 8);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
    run<0>("MFMA only", out, cyc);
    run<1>("+ 1 ds_read_b128", out, cyc);
    run<2>("+ 2 ds_read_b128", out, cyc);
    run<3>("+ 1 ds_read_b128 + 2 s_add", out, cyc);
    run<4>("+ 1 v_fma", out, cyc);
    run<5>("+ 4 v_fma", out, cyc);
    run<6>("+ 1 s_nop 0", out, cyc);
    run<7>("+ 2 s_add", out, cyc);
    run<8>("+ 1 ds_read_b128 + 2 v_fma", out, cyc);
    run<9>("+ 2 ds_read_b128 + 4 v_fma", out, cyc);
    run<10>("2 ds_read_b128 after every 2nd MFMA", out, cyc);
    run<11>("4 ds_read_b128 after every 4th MFMA", out, cyc);
    run<12>("8 ds_read_b128 after every 8th MFMA", out, cyc);
    run<13>("+ 1 ds_read_b128 into AGPRs", out, cyc);
    run<20>("MMMM [4 ds + 4 vfma]", out, cyc);
    run<21>("MMMM [4 ds] MMMM [8 vfma]", out, cyc);
    run<22>("MMMM [4 ds + 2 vfma + 2 salu]", out, cyc);
    run<23>("M[v] M[v] M[v] M[4 ds]", out, cyc);
    run<24>("M[2v] M[2v] M[2v] M[4 ds]", out, cyc);
    run<25>("M[4v] M[4v] M[4v] M[4 ds]", out, cyc);
    run<26>("M M[ds+v] M M[3 ds]", out, cyc);
    run<27>("MMMM [5 ds]", out, cyc);
    run<28>("MMMM [6 ds]", out, cyc);
    run<29>("MM [2 ds + 2 v]", out, cyc);
    run<14>("+ 2 ds_read_b64", out, cyc);
    run<15>("+ 1 ds_read_b128 + s_nop 7", out, cyc);
    run<16>("+ s_nop 3 + 1 ds_read_b128", out, cyc);
    return 0;
}
