// Microbenchmark: what a device-scope release / acquire costs inside a kernel on MI355X (8 XCDs, private L2s) -- the price of any
// "last workgroup to arrive finishes the tile" scheme.  Each workgroup writes 24 KB, optionally fences, bumps a counter; the last
// of each group of 4 reads the group's 96 KB back and checks it.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/fence_cost.hip -o scripts/ubench/fence_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0: no communication (baseline)  1: __threadfence (agent scope) + atomic, last reads  2: workgroup-scope fence only
                      // 3: no fence at all: the data itself travels as agent-scope relaxed atomic stores / loads (sc1: write-through /
                      //    read-around the XCD's L2), s_waitcnt vmcnt(0) in front of the counter's atomic
__global__ void __launch_bounds__(512) k(float* buf, unsigned* cnt, unsigned* bad, int iters, unsigned long long* cyc) {
    const int tid = threadIdx.x, g = blockIdx.x >> 2;
    __shared__ unsigned s_last;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float* mine = buf + ((long)it * gridDim.x + blockIdx.x) * 6144;
        if (MODE == 3) {
            for (int e = tid; e < 6144; e += 512) __hip_atomic_store(mine + e, (float)(blockIdx.x + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int e = tid; e < 6144; e += 512) mine[e] = (float)(blockIdx.x + it);
        }
        if (MODE == 0) continue;
        if (MODE == 1) __threadfence();
        else if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) s_last = (MODE == 3 ? __hip_atomic_fetch_add(cnt + it * (gridDim.x >> 2) + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : atomicAdd(cnt + it * (gridDim.x >> 2) + g, 1u)) == 3u;
        __syncthreads();
        if (s_last) {
            if (MODE == 1) __threadfence();
            unsigned wrong = 0;
            for (int b = 0; b < 4; ++b) {
                const float* p = buf + ((long)it * gridDim.x + (g * 4 + b)) * 6144;
                for (int e = tid; e < 6144; e += 512)
                    wrong += (MODE == 3 ? __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[e]) != (float)(g * 4 + b + it);
            }
            if (wrong) atomicAdd(bad, wrong);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    const int blocks = 256, iters = 50;
    float* buf; unsigned* cnt; unsigned* bad; unsigned long long* cyc;
    hipMalloc(&buf, sizeof(float) * 6144 * blocks * iters); hipMalloc(&cnt, 4 * 64 * iters); hipMalloc(&bad, 4); hipMalloc(&cyc, 8);
    hipMemset(cnt, 0, 4 * 64 * iters); hipMemset(bad, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(buf, cnt, bad, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned hb; unsigned long long c; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.3f ms total, %.2f us per iteration (block 0: %.0f cycles), mismatches %u\n", name, ms, ms * 1e3 / iters, (double)c / iters, hb);
    hipFree(buf); hipFree(cnt); hipFree(bad); hipFree(cyc);
}

int main() {
    run<0>("stores only");
    run<1>("agent-scope fence + counter, last reads");
    run<2>("workgroup-scope fence + counter, last reads");
    run<0>("stores only");
    run<1>("agent-scope fence + counter, last reads");
    run<3>("sc1 stores / loads + counter, no fence");
    run<3>("sc1 stores / loads + counter, no fence");
    return 0;
}
