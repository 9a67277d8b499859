// Microbenchmark: the relu + split-half + LDS-store epilogue of k_cnet's P1 / h2 hand-over, per value, for three instruction
// mixes, at one and two waves per SIMD; plus a probe of the NaN bit patterns the matrix pipe and the VALU produce (the
// integer-max ReLU keeps NaNs whose sign bit is clear).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/epi_split.hip -o scripts/ubench/bin/epi_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float relu_cmp(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float relu_bits(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

template <int VAR0>
__device__ __forceinline__ void split4(f32x4_t acc, f32x4_t rs, f32x4_t bb, h4& hi, h4& lo) {
    constexpr int VAR = VAR0;
    if (VAR == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float v = relu_cmp(fmaf(acc[t], rs[t], bb[t]));
            const _Float16 x0 = (_Float16)v;
            hi[t] = x0; lo[t] = (_Float16)(v - (float)x0);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            const float v0 = relu_bits(fmaf(acc[t], rs[t], bb[t])), v1 = relu_bits(fmaf(acc[t + 1], rs[t + 1], bb[t + 1]));
            const f32x2_t vv = {v0, v1};
            const h2 x = __builtin_convertvector(vv, h2);
            float r0, r1;
            if (VAR == 1) { r0 = v0 - (float)x[0]; r1 = v1 - (float)x[1]; }
            else {
                const unsigned xb = __builtin_bit_cast(unsigned, x);
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(xb), "v"(v0));
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(xb), "v"(v1));
            }
            const f32x2_t rr = {r0, r1};
            const h2 y = __builtin_convertvector(rr, h2);
            hi[t] = x[0]; hi[t + 1] = x[1]; lo[t] = y[0]; lo[t + 1] = y[1];
        }
    }
}

// a wave owns RT x PT tiles of 32 x 32 (16 accumulator registers each); store layout [plane][chunk][pixel][8] as in k_cnet
template <int VAR, int RT, int PT, int NT>
__global__ void __launch_bounds__(NT) k_epi(const float* src, const float* tab, float* sink, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    constexpr int PXT = 128, NCH = 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, kl = lane >> 5, ml = lane & 31;
    float* t_rs = reinterpret_cast<float*>(lds + 2 * NCH * PXT * 8);
    float* t_b = t_rs + 512;
    for (int e = tid; e < 512; e += blockDim.x) { t_rs[e] = tab[e]; t_b[e] = tab[512 + e]; }
    f32x16_t acc[RT][PT];
    for (int i = 0; i < RT; ++i)
        for (int j = 0; j < PT; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = src[(tid * 16 + r + 7 * (i * PT + j)) & 4095];
    __syncthreads();
    const int nw = blockDim.x >> 6;
    const int tiles_w = RT * PT;
    const int rt0 = (wid * tiles_w) / 4 % 8, pt0 = (wid * tiles_w) % 4;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 3) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int oa = ((rt0 + i) & 7) * 32 + 16 * gp + 4 * kl, ob = oa + 8;
                    const f32x4_t rsa = *reinterpret_cast<const f32x4_t*>(t_rs + oa), bba = *reinterpret_cast<const f32x4_t*>(t_b + oa);
                    const f32x4_t rsb = *reinterpret_cast<const f32x4_t*>(t_rs + ob), bbb = *reinterpret_cast<const f32x4_t*>(t_b + ob);
                    const int chunk = ((rt0 + i) & 7) * 4 + 2 * gp + kl;       // lower half-wave stores chunk a, upper chunk b
#pragma unroll
                    for (int j = 0; j < PT; ++j) {
                        h4 hia, loa, hib, lob;
                        f32x4_t a4 = {acc[i][j][8 * gp], acc[i][j][8 * gp + 1], acc[i][j][8 * gp + 2], acc[i][j][8 * gp + 3]};
                        f32x4_t b4 = {acc[i][j][8 * gp + 4], acc[i][j][8 * gp + 5], acc[i][j][8 * gp + 6], acc[i][j][8 * gp + 7]};
                        split4<2>(a4, rsa, bba, hia, loa);
                        split4<2>(b4, rsb, bbb, hib, lob);
                        typedef unsigned u2v __attribute__((ext_vector_type(2)));
                        typedef unsigned u4v __attribute__((ext_vector_type(4)));
                        u2v ha = __builtin_bit_cast(u2v, hia), hb = __builtin_bit_cast(u2v, hib), la = __builtin_bit_cast(u2v, loa), lb = __builtin_bit_cast(u2v, lob);
                        u4v oh, ol;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            auto r = __builtin_amdgcn_permlane32_swap(ha[q], hb[q], false, false);
                            oh[q] = r[0]; oh[2 + q] = r[1];
                            auto r2 = __builtin_amdgcn_permlane32_swap(la[q], lb[q], false, false);
                            ol[q] = r2[0]; ol[2 + q] = r2[1];
                        }
                        _Float16* dst = lds + ((long)chunk * PXT + ((pt0 + j) & 3) * 32 + ml) * 8;
                        *reinterpret_cast<u4v*>(dst) = oh;
                        *reinterpret_cast<u4v*>(dst + (long)NCH * PXT * 8) = ol;
                    }
                }
        } else
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int o = ((rt0 + i) & 7) * 32 + 8 * gq + 4 * kl;
                const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(t_rs + o);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(t_b + o);
                const int chunk = ((rt0 + i) & 7) * 4 + gq;
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    h4 hi, lo;
                    f32x4_t a4 = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                    split4<VAR>(a4, rs, bb, hi, lo);
                    _Float16* dst = lds + ((long)chunk * PXT + ((pt0 + j) & 3) * 32 + ml) * 8 + 4 * kl;
                    *reinterpret_cast<h4*>(dst) = hi;
                    *reinterpret_cast<h4*>(dst + (long)NCH * PXT * 8) = lo;
                }
            }
        __syncthreads();
        // keep the accumulators live and changing so nothing is hoisted
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j][it & 15] += 1.0f;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int e = tid; e < 2 * NCH * PXT * 8; e += blockDim.x) s += (float)lds[e];
    sink[blockIdx.x * blockDim.x + tid] = s + acc[0][0][0];
    if (lane == 0) cyc[blockIdx.x * nw + wid] = t1 - t0;
}

template <int VAR, int RT, int PT, int NT>
void run(const float* src, const float* tab, float* sink, unsigned long long* cyc) {
    const int iters = 200, threads = NT;
    const size_t lds = 2 * 32 * 128 * 8 * 2 + 1024 * 4;
    hipFuncSetAttribute((const void*)k_epi<VAR, RT, PT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_epi<VAR, RT, PT, NT><<<256, threads, lds>>>(src, tab, sink, cyc, 10);
    hipDeviceSynchronize();
    k_epi<VAR, RT, PT, NT><<<256, threads, lds>>>(src, tab, sink, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
    const double vals_simd = (double)RT * PT * 1024 * (threads / 64) / 4;     // values per SIMD and iteration
    printf("variant %d, %d waves, %dx%d tiles per wave: %.0f cycles per pass (%d KB of halves), %.2f cycles per 64 values per SIMD\n", VAR, threads / 64, RT,
           PT, (double)mx / iters, RT * PT * (threads / 64) * 4, (double)mx / iters / (vals_simd / 64));
}

__global__ void k_nan(unsigned* out) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)0.f; b[q] = (_Float16)1.f; }
    f32x16_t acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // case 0: (+inf) * 1 + (-inf) * 1
    const _Float16 pinf = (_Float16)__builtin_inff(), ninf = -pinf;
    a[0] = pinf; a[1] = ninf;
    f32x16_t c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    // case 1: inf * 0
    a[1] = (_Float16)0.f; b[0] = (_Float16)0.f;
    f32x16_t c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    // case 2: a NaN input with the sign bit set
    unsigned short nb = 0xfe00; _Float16 nn; memcpy(&nn, &nb, 2);
    a[0] = nn; b[0] = (_Float16)1.f;
    f32x16_t c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    // case 3: accumulator +inf, product -inf
    for (int r = 0; r < 16; ++r) acc[r] = __builtin_inff();
    a[0] = ninf;
    f32x16_t c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    // case 4: a NaN input with a CLEAR sign bit
    unsigned short pb = 0x7e00; _Float16 pn; memcpy(&pn, &pb, 2);
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    a[0] = pn;
    f32x16_t c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    // case 5: accumulator = NaN with a clear sign bit
    for (int r = 0; r < 16; ++r) acc[r] = __uint_as_float(0x7fc00000u);
    a[0] = (_Float16)1.f;
    f32x16_t c5 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) {
        volatile float one = 1.f, twof = 2.f;
        out[9] = __float_as_uint(c4[0]); out[10] = __float_as_uint(c5[0]);
        out[11] = __float_as_uint(fmaf(__uint_as_float(0x7fc00000u), twof, one));     // fma with a positive NaN
        out[12] = __float_as_uint(fmaf(__uint_as_float(0xffc00000u), -twof, -one));   // negative NaN times a negative scale
        out[13] = __float_as_uint(fmaf(one, twof, __uint_as_float(0x7fc00000u)));     // positive NaN as the addend
        out[0] = __float_as_uint(c0[0]); out[1] = __float_as_uint(c1[0]); out[2] = __float_as_uint(c2[0]); out[3] = __float_as_uint(c3[0]);
        const float vi = __builtin_inff();
        volatile float z = 0.f;
        out[4] = __float_as_uint(vi * z);                       // VALU inf * 0
        out[5] = __float_as_uint(vi - (vi + z));                // VALU inf - inf
        out[6] = __float_as_uint(fmaf(__uint_as_float(0xffc00000u), 2.0f, 1.0f + z));   // fma with a negative NaN
        const f32x2_t two = {c0[0], c2[0]};
        const h2 hh = __builtin_convertvector(two, h2);
        out[7] = __builtin_bit_cast(unsigned, hh);              // packed f16 conversion of those NaNs
        out[8] = __float_as_uint(__int_as_float(max(__float_as_int(c0[0]), 0)));
    }
}

int main() {
    float *src, *tab, *sink; unsigned long long* cyc; unsigned* nb;
    hipMalloc(&src, 4096 * 4); hipMalloc(&tab, 1024 * 4); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 8 * 256 * 8); hipMalloc(&nb, 128);
    float hs[4096], ht[1024];
    for (int i = 0; i < 4096; ++i) hs[i] = ((i * 2654435761u) >> 8 & 0xffff) / 65536.f * 8.f - 3.f;
    for (int i = 0; i < 512; ++i) { ht[i] = 1.0f / 4096.f * (1 + (i & 3)); ht[512 + i] = 0.01f * (i & 7); }
    hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice); hipMemcpy(tab, ht, sizeof(ht), hipMemcpyHostToDevice);
    // P1 at level 1 (8 waves, 2 x 2 tiles per wave and sub-pass) and the h2 hand-over (8 waves, 2 x 4 tiles), 4 waves: 4 x 4
    run<0, 2, 2, 512>(src, tab, sink, cyc); run<1, 2, 2, 512>(src, tab, sink, cyc); run<2, 2, 2, 512>(src, tab, sink, cyc); run<3, 2, 2, 512>(src, tab, sink, cyc);
    run<0, 2, 4, 512>(src, tab, sink, cyc); run<1, 2, 4, 512>(src, tab, sink, cyc); run<2, 2, 4, 512>(src, tab, sink, cyc); run<3, 2, 4, 512>(src, tab, sink, cyc);
    run<0, 4, 4, 256>(src, tab, sink, cyc); run<1, 4, 4, 256>(src, tab, sink, cyc); run<2, 4, 4, 256>(src, tab, sink, cyc); run<3, 4, 4, 256>(src, tab, sink, cyc);
    k_nan<<<1, 64>>>(nb);
    unsigned h[32]; hipMemcpy(h, nb, 128, hipMemcpyDeviceToHost);
    printf("NaN bits 2: mfma(+NaN in) %08x, mfma(acc +NaN) %08x, fma(+NaN * 2 + 1) %08x, fma(-NaN * -2 - 1) %08x, fma(1 * 2 + +NaN) %08x\n", h[9], h[10], h[11], h[12], h[13]);
    printf("NaN bits: mfma inf-inf %08x, mfma inf*0 %08x, mfma(-NaN in) %08x, mfma acc inf + (-inf) %08x, valu inf*0 %08x, valu inf-inf %08x, fma(-NaN) %08x, cvt_pk %08x, relu_bits(mfma NaN) %08x\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    return 0;
}
