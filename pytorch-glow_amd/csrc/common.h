// common.h -- shared device helpers and host-side error plumbing for libglowhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/glowhip.h"

namespace glowhip {

void set_error(const char* fmt, ...);

#define GH_REQUIRE(cond, ...)                    \
    do {                                         \
        if (!(cond)) {                           \
            ::glowhip::set_error(__VA_ARGS__);   \
            return GLOWHIP_EINVAL;               \
        }                                        \
    } while (0)

#define GH_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            ::glowhip::set_error("%s: %s", name, hipGetErrorString(e_));             \
            return GLOWHIP_ELAUNCH;                                                  \
        }                                                                            \
    } while (0)

#define GH_TRY(expr)               \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != GLOWHIP_OK) return rc_; \
    } while (0)

constexpr float LOG_2PI_F = 1.8378770664093453f;
constexpr float LOGSCALE = 3.0f;  // network/module.py:10 logscale_factor

// ---- per-sample log-determinant accumulators ---------------------------------------------------
// Kept as signed Q31.32 fixed point in a u64 so that atomic accumulation from many workgroups is
// order-independent (bitwise reproducible), unlike float atomics.
constexpr double FIX_SCALE = 4294967296.0;
constexpr int ACC_EXTRA = 7;
// acc: 2 N words -- acc[n] the accumulator of sample n, acc[N + n] its STICKY NON-FINITE FLAG -- and, in a plan's workspace,
// ACC_EXTRA more accumulator rows acc[(2 + k) N + n] that k_finalize adds to acc[n]: the finishing kernel of the product path
// spreads its one-per-workgroup atomics over them (1 024 workgroups on the four cache lines of 64 samples' accumulators queue up).  A NaN / inf partial sum cannot
// be represented in fixed point (the conversion would silently turn NaN into 0): it raises the flag instead, and k_finalize
// reports NaN / +inf / -inf for a flagged sample -- an fp16-range overflow in the split-half kernels, diverged weights or out-of-distribution
// input surface as a non-finite nll exactly as in the reference, never as a finite wrong one.
// flag bits: 1 = a NaN term, 2 = a +inf term, 4 = a -inf term (k_finalize: NaN wins, +inf and -inf together give NaN, as in the
// reference's floating-point sum)
__device__ __forceinline__ void fix_flag_nonfinite(unsigned long long* acc, long n, int N, double v) {
    atomicOr(acc + N + n, isnan(v) ? 1ull : (v > 0 ? 2ull : 4ull));
}
__device__ __forceinline__ void fix_atomic_add(unsigned long long* acc, long n, int N, double v) {
    if (!isfinite(v)) { fix_flag_nonfinite(acc, n, N, v); return; }
    long long q = __double2ll_rn(v * FIX_SCALE);
    atomicAdd(acc + n, (unsigned long long)q);
}
__device__ __forceinline__ double fix_to_double(unsigned long long a) {
    return (double)(long long)a * (1.0 / FIX_SCALE);
}

// ---- wave64 / workgroup reductions (fixed tree => deterministic) ------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// Sum over a workgroup of NT threads; result valid in thread 0.  red: NT/64 doubles of LDS.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) t += red[i];
    }
    __syncthreads();
    return t;
}

// XCD-aware remap (8 XCDs, private L2s): hardware places block b on XCD b%8; give each XCD a contiguous
// run of logical tiles so the tiles that share an operand panel share an L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// ReLU that PROPAGATES NaN like the reference's (torch.relu(NaN) = NaN); fmaxf(NaN, 0) = 0 would turn an fp16-range overflow
// (inf - inf = NaN in a split-half operand) into a finite wrong activation
__device__ __forceinline__ float relu_(float v) { return v < 0.f ? 0.f : v; }

// ---- debug build only (-DGLOWHIP_DEBUG_STAMPS): s_memtime stamps of workgroup 0 / wave 0, read back by scripts/stamps.py
#ifdef GLOWHIP_DEBUG_STAMPS
// (one array + one extern "C" reader per translation unit: no relocatable device code in this build)
#define GH_STAMPS_DEFINE(name)                                                                                  \
    namespace glowhip { __device__ unsigned long long g_stamps_local[8 * 64]; }                                 \
    extern "C" int glowhip_debug_read_stamps_##name(unsigned long long* dst) {                                  \
        return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(glowhip::g_stamps_local), sizeof(unsigned long long) * 64); \
    }                                                                                                           \
    extern "C" int glowhip_debug_read_stamps_all_##name(unsigned long long* dst) {   /* [wave][stamp] */        \
        return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(glowhip::g_stamps_local), sizeof(unsigned long long) * 8 * 64); \
    }
// launch timeline: every workgroup of the last stamped launch records (s_memrealtime at start, at end, HW_ID, XCC_ID) -- 100 MHz
// chip-wide clock, comparable across CUs
#define GH_WGTIMES_DEFINE(name)                                                                                 \
    namespace glowhip { __device__ unsigned long long g_wg_times[4096 * 4]; }                                   \
    extern "C" int glowhip_debug_read_wgtimes_##name(unsigned long long* dst, int n) {                          \
        return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(glowhip::g_wg_times), sizeof(unsigned long long) * 4 * n); \
    }
#define GH_WG_BEGIN() do { if ((threadIdx.x) == 0) { const int b_ = blockIdx.y * gridDim.x + blockIdx.x; if (b_ < 4096) { \
        g_wg_times[b_ * 4] = __builtin_amdgcn_s_memrealtime(); g_wg_times[b_ * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); \
        g_wg_times[b_ * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } } } while (0)
#define GH_WG_END() do { if ((threadIdx.x) == 0) { const int b_ = blockIdx.y * gridDim.x + blockIdx.x; if (b_ < 4096) \
        g_wg_times[b_ * 4 + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#ifndef GH_STAMP_BLOCK
#define GH_STAMP_BLOCK 0
#endif
// lane 0 of EVERY wave of the stamped workgroup records: [wave][stamp] (wave 0's row is what the one-wave readers see)
#define GH_STAMP(i) do { if (blockIdx.x == GH_STAMP_BLOCK && blockIdx.y == 0 && (threadIdx.x & 63) == 0) g_stamps_local[(threadIdx.x >> 6) * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#define GH_STAMP_VAL(i, v) do { if (blockIdx.x == GH_STAMP_BLOCK && blockIdx.y == 0 && (threadIdx.x & 63) == 0) g_stamps_local[(threadIdx.x >> 6) * 64 + (i)] = (unsigned long long)(v); } while (0)
#else
#define GH_STAMPS_DEFINE(name)
#define GH_WGTIMES_DEFINE(name)
#define GH_WG_BEGIN() do { } while (0)
#define GH_WG_END() do { } while (0)
#define GH_STAMP(i) do { } while (0)
#define GH_STAMP_VAL(i, v) do { } while (0)
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace glowhip
