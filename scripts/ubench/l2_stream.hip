// Microbenchmark: rate at which every CU can stream the SAME L2-resident weight image with 16-byte-per-lane loads (the A-operand
// stream of k_cnet's P2), by access pattern and loads in flight.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/l2_stream.hip -o scripts/ubench/l2_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// image: 2 planes x (K/8 groups) x 512 rows x 16 bytes.  PATTERN 0: k_cnet's (wave = 2 row tiles x 2 k groups x 2 planes per k-step:
// per load instruction two 512-byte runs 8 KB apart).  PATTERN 1: one contiguous 1 KB per load instruction.
template <int PATTERN, int WAVES, int DEPTH>
__global__ void __launch_bounds__(WAVES * 64) k(const char* img, unsigned* out, int ksteps, int reps, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kl = lane >> 5, ml = lane & 31;
    u4 acc = {0, 0, 0, 0};
    const long plane = (long)ksteps * 2 * 512 * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int s = 0; s < ksteps; s += DEPTH) {
            u4 v[DEPTH][4];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const char* p;
                if (PATTERN == 0) p = img + ((long)((s + d) * 2 + kl) * 512 + (wid * (512 / WAVES)) + ml) * 16;
                else p = img + ((long)(s + d) * 2 * 512 * 16) + (long)wid * (2 * 512 * 16 / WAVES) + lane * 16;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long off = PATTERN == 0 ? i * 512 : i * 1024;
                    v[d][i] = *reinterpret_cast<const u4*>(p + off);
                    v[d][2 + i] = *reinterpret_cast<const u4*>(p + off + plane);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc ^= v[d][i];
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int PATTERN, int WAVES, int DEPTH>
void run(const char* img, int blocks) {
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, 4 * blocks * WAVES * 64); hipMalloc(&cyc, 8);
    const int ksteps = 30, reps = 40;        // 30 k-steps x 32 KB = 960 KB image (rows split over the waves)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<PATTERN, WAVES, DEPTH><<<blocks, WAVES * 64>>>(img, out, ksteps, 2, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<PATTERN, WAVES, DEPTH><<<blocks, WAVES * 64>>>(img, out, ksteps, reps, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // bytes per workgroup: with WAVES < 8 only part of the rows is read
    const double bytes_wg = (double)reps * ksteps * WAVES * 4 * 1024;
    printf("pattern %d waves %d depth %d blocks %d: %.3f ms, %.2f TB/s aggregate, %.1f B/clk/CU (wave 0 clock), %.0f ticks/us\n", PATTERN, WAVES, DEPTH, blocks, ms,
           bytes_wg * blocks / ms / 1e9, bytes_wg / (double)c * (blocks > 256 ? blocks / 256.0 : 1.0), c / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main() {
    char* img; hipMalloc(&img, 2 << 20); hipMemset(img, 1, 2 << 20);
    run<0, 8, 3>(img, 256);
    run<1, 8, 3>(img, 256);
    run<0, 8, 6>(img, 256);
    run<1, 8, 6>(img, 256);
    run<0, 4, 3>(img, 256);
    run<0, 4, 6>(img, 256);
    run<0, 8, 3>(img, 32);
    run<0, 8, 3>(img, 64);
    run<0, 8, 3>(img, 512);
    return 0;
}
