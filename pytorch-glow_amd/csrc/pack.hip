// pack.hip -- batched re-derivation of parameter-dependent data (glowhip_plan_pack): ONE launch for all
// exp(3*logs) tables and ONE for all MFMA weight images of a plan, driven by device-resident job tables
// (a 96-step Glow has ~390 scale jobs and ~290 repack jobs; launching them one by one cost 11 ms per pack).
#include "kernels.h"
#include "conv_mfma.h"
#include "sh.h"

namespace glowhip {

__device__ __forceinline__ size_t align_up_dev(size_t v, size_t a) { return (v + a - 1) / a * a; }

__global__ void __launch_bounds__(256) k_pack_scales_batched(const ScaleJob* __restrict__ jobs, char* packed) {
    const ScaleJob j = jobs[blockIdx.y];
    float* scale = (float*)(packed + j.scale_off);
    float* inv = j.has_inv ? (float*)(packed + j.inv_off) : nullptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < j.n; i += gridDim.x * 256) {
        const float l3 = j.logs[i] * LOGSCALE;
        scale[i] = expf(l3);
        if (inv) inv[i] = expf(-l3);
    }
}

// wide: wt[k][o] = w[o][k] (k = ci*k*k + tap), zero rows k >= K   -- K-major image for k_conv_wide
// tail: wp[(((chunk*8 + c4)*9 + tap)*MT + mt)*64 + kq*16 + i] = w[o(mt*16+i)][chunk*32 + c4*4 + kq][tap]
__global__ void __launch_bounds__(256) k_repack_batched(const RepackJob* __restrict__ jobs, char* packed) {
    const RepackJob j = jobs[blockIdx.y];
    if (j.kind >= REPACK_SH2_GEMM) return;   // k_repack_sh2_batched
    float* out = (float*)(packed + j.out_off);
    if (j.kind == REPACK_WIDE) {
        const long total = (long)j.Kpad * j.Cout;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int k = (int)(e / j.Cout), o = (int)(e - (long)k * j.Cout);
            out[e] = (k < j.K) ? j.w[(long)o * j.K + k] : 0.f;
        }
    } else if (j.kind == REPACK_FIRST) {
        // wf[(chunk*54 + tap*6 + cl)][o] = w[o][chunk*6 + cl][tap]   -- (6-channel chunk, tap, channel) K order
        const long total = (long)9 * j.Cin * j.Cout;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int o = (int)(e % j.Cout);
            const int k = (int)(e / j.Cout);
            const int chunk = k / 54, r = k - chunk * 54;
            const int tap = r / 6, cl = r - tap * 6;
            const int ci = chunk * 6 + cl;
            const float wv = j.transposed ? j.w[((long)ci * j.Cout + o) * 9 + (8 - tap)] : j.w[((long)o * j.Cin + ci) * 9 + tap];
            out[e] = j.fold_logs ? wv * expf(j.fold_logs[o] * LOGSCALE) : wv;
        }
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < j.Cout; o += (long)gridDim.x * 256)
            out[total + o] = j.fold_logs ? j.fold_bias[o] * expf(j.fold_logs[o] * LOGSCALE) : 0.f;
    } else {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < j.total; e += (long)gridDim.x * 256) {
            const int i = (int)(e & 15), kq = (int)((e >> 4) & 3);
            long t = e >> 6;
            const int mt = (int)(t % j.MT); t /= j.MT;
            const int tap = (int)(t % 9); t /= 9;
            const int c4 = (int)(t % (TAIL_CK / 4));
            const int chunk = (int)(t / (TAIL_CK / 4));
            const int ci = chunk * TAIL_CK + c4 * 4 + kq;
            const int o = tail_row_channel(mt * 16 + i, j.Cout, j.paired);
            out[e] = (o >= 0 && ci < j.Cin)
                         ? (j.transposed ? j.w[((long)ci * j.Cout + o) * 9 + (8 - tap)] : j.w[((long)o * j.Cin + ci) * 9 + tap])
                         : 0.f;
        }
    }
}

// SH2 images (sh.h): half [plane][Kp/8][M][8] of w'[r][k] * 2^e[r], then M floats row scale, then M floats bias.  ONE WAVE PER
// OUTPUT ROW: the row's largest |w'| fixes its exponent e (largest value in [2^12, 2^13)), which needs the whole row first.
//   SH2_GEMM  (f.2): row o, k = input channel;           w' = w[o][k] exp(3 logs[o]);  rowscale = 2^-e, bias = b' * 16
//   SH2_FIRST (f.0): row o, k = (8-channel chunk, tap, 8 channels), tap fastest; same folding; j.K = G groups
//   SH2_TAIL  (f.4): row m = tap * Cout + co (< j.Kpad = Mpad4 rows, zero beyond 9 Cout), k = input channel;
//                    rowscale = 2^-e / 16 (undoes the activation scale as well), no bias
__device__ __forceinline__ const float* repack_src(const RepackJob& j, const char* packed) {
    return j.w ? j.w : reinterpret_cast<const float*>(packed + j.w_off);
}

template <int KIND>
__device__ __forceinline__ void repack_sh2_rows(const RepackJob& j, char* packed) {
    const float* jw = repack_src(j, packed);
    // a wave takes EIGHT consecutive rows: lane = (k group within the pass) * 8 + row, so the eight 16-byte groups of one k group
    // are 128 contiguous bytes of the image (a single row per wave wrote 16 bytes every M * 16)
    const int lane = threadIdx.x & 63, r8 = lane & 7, gl = lane >> 3;
    const int M = KIND == REPACK_SH2_TAIL ? j.Kpad : j.Cout;
    const int Kp = KIND == REPACK_SH2_FIRST ? j.K * 8 : j.Cin;
    const int ngroups = Kp / 8;
    _Float16* oh = (_Float16*)(packed + j.out_off);
    float* rowscale = (float*)(packed + j.out_off + (size_t)2 * Kp * M * sizeof(_Float16));
    float* rbias = rowscale + M;
    const int nchunk = (j.Cin + 7) / 8;
    for (int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; r0 < M; r0 += gridDim.x * 32) {
        const int r = r0 + r8;
        const bool rv = r < M;
        float fold = 1.f;
        if (rv && KIND != REPACK_SH2_TAIL && j.fold_logs) fold = expf(j.fold_logs[r] * LOGSCALE);
        // the 8 values of k group gi of this lane's row
        auto group = [&](int gi, float (&v)[8]) {
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) v[k8] = 0.f;
            if (!rv || gi >= ngroups) return;
            if (KIND == REPACK_SH2_GEMM) {          // contiguous in memory: two 16-byte loads when the row length allows
                // (k-permuted image: the group's two halves are 4 consecutive k each, 8 apart -- sh.h sh2_kperm_src)
                const float* src = jw + (long)r * j.Cin + (j.kperm ? sh2_kperm_src(gi * 8) : gi * 8);
                const int h2o = j.kperm ? 8 : 4;
                if ((j.Cin & 3) == 0) {
                    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src), b = *reinterpret_cast<const f32x4_t*>(src + h2o);
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) { v[k8] = a[k8] * fold; v[4 + k8] = b[k8] * fold; }
                } else {
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) v[k8] = src[(k8 & 3) + (k8 >> 2) * h2o] * fold;
                }
            } else if (KIND == REPACK_SH2_FIRST) {
                const int ch = gi / 9, tap = gi - ch * 9;
                if (ch < nchunk) {
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8)
                        if (ch * 8 + k8 < j.Cin) v[k8] = jw[((long)r * j.Cin + ch * 8 + k8) * 9 + tap] * fold;
                }
            } else if (r < 9 * j.Cout) {
                const int tap = r / j.Cout, co = r - tap * j.Cout;
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) v[k8] = jw[((long)co * j.Cin + gi * 8 + k8) * 9 + tap];
            }
        };
        // pass 1: the row's largest magnitude (the values are re-read in pass 2: they come from L2, and holding them would cost
        // 64+ registers per lane -- an earlier version did and ran at one wave per SIMD)
        float mx = 0.f;
        for (int gi = gl; gi < ngroups; gi += 8) {
            float v[8];
            group(gi, v);
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) mx = fmaxf(mx, fabsf(v[k8]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);          // mx = f * 2^ex, f in [0.5, 1)
            e = 13 - ex;                    // mx * 2^e in [2^12, 2^13)
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float up = ldexpf(1.f, e);
        for (int gi = gl; gi < ngroups; gi += 8) {
            float v[8];
            group(gi, v);
            h8 hi, lo;
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                _Float16 a, b;
                sh2_split(v[k8] * up, a, b);
                hi[k8] = a; lo[k8] = b;
            }
            if (rv) {
                *reinterpret_cast<h8*>(oh + ((long)gi * M + r) * 8) = hi;
                *reinterpret_cast<h8*>(oh + ((long)(ngroups + gi) * M + r) * 8) = lo;
            }
        }
        if (gl == 0 && rv) {
            if (KIND == REPACK_SH2_TAIL) {
                rowscale[r] = ldexpf(1.f, -e) * SH2_ACT_INV;
                rbias[r] = 0.f;
            } else {
                rowscale[r] = ldexpf(1.f, -e);
                rbias[r] = (j.fold_bias ? j.fold_bias[r] * fold : 0.f) * SH2_ACT_SCALE;
            }
        }
    }
}

// f.0 and f.4 weights are [o][i][3][3]: one (8 input channels) x (9 taps) BRICK of a row is 72 contiguous floats.  A lane loads
// whole bricks (wide loads where the alignment allows) and emits their nine k groups; gathering tap by tap instead fetched a
// 128-byte line per 4-byte load and ran at the L2 -> L1 rate (169 us per pack for the f.4 images alone).
__device__ __forceinline__ void load_brick(const float* src, int nk, bool valid, float (&b)[72]) {
#pragma unroll
    for (int i = 0; i < 72; ++i) b[i] = 0.f;
    if (!valid) return;
    const int count = nk * 9;
    if (nk == 8 && ((size_t)src & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const f32x4_t v = reinterpret_cast<const f32x4_t*>(src)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[i * 4 + q] = v[q];
        }
    } else if ((count & 1) == 0 && ((size_t)src & 7) == 0) {
#pragma unroll
        for (int i = 0; i < 36; ++i)
            if (2 * i < count) {
                const f32x2_t v = reinterpret_cast<const f32x2_t*>(src)[i];
                b[2 * i] = v[0]; b[2 * i + 1] = v[1];
            }
    } else {
#pragma unroll
        for (int i = 0; i < 72; ++i)
            if (i < count) b[i] = src[i];
    }
}

__device__ __forceinline__ int sh2_row_exponent(float mx) {
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) {
        int ex;
        (void)frexpf(mx, &ex);
        e = 13 - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    return e;
}

// group tap of a brick -> the (hi, lo) pair of 16-byte groups
__device__ __forceinline__ void brick_emit(const float (&b)[72], int tap, float mul, h8& hi, h8& lo) {
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
        _Float16 x, y;
        sh2_split(b[k8 * 9 + tap] * mul, x, y);
        hi[k8] = x; lo[k8] = y;
    }
}

// SH2_FIRST: one lane per output row, a loop over the row's bricks (two passes: row maximum, then the split)
__device__ __forceinline__ void repack_sh2_first(const RepackJob& j, char* packed) {
    const int M = j.Cout, G = j.K, Kp = G * 8;
    _Float16* oh = (_Float16*)(packed + j.out_off);
    float* rowscale = (float*)(packed + j.out_off + (size_t)2 * Kp * M * sizeof(_Float16));
    float* rbias = rowscale + M;
    const int nchunk = (j.Cin + 7) / 8;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < M; r += gridDim.x * 256) {
        const float fold = j.fold_logs ? expf(j.fold_logs[r] * LOGSCALE) : 1.f;
        const float* row = repack_src(j, packed) + (long)r * j.Cin * 9;
        float mx = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) {
            float b[72];
            load_brick(row + ch * 72, min(8, j.Cin - ch * 8), true, b);
#pragma unroll
            for (int i = 0; i < 72; ++i) mx = fmaxf(mx, fabsf(b[i] * fold));
        }
        const int e = sh2_row_exponent(mx);
        const float up = ldexpf(1.f, e);
        for (int ch = 0; ch < nchunk; ++ch) {
            float b[72];
            load_brick(row + ch * 72, min(8, j.Cin - ch * 8), true, b);
#pragma unroll
            for (int i = 0; i < 72; ++i) b[i] *= fold;      // same rounding order as the row maximum: (w * fold) * 2^e
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                h8 hi, lo;
                brick_emit(b, tap, up, hi, lo);
                const int gi = ch * 9 + tap;
                *reinterpret_cast<h8*>(oh + ((long)gi * M + r) * 8) = hi;
                *reinterpret_cast<h8*>(oh + ((long)(G + gi) * M + r) * 8) = lo;
            }
        }
        h8 z;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) z[k8] = (_Float16)0.f;
        for (int gi = nchunk * 9; gi < G; ++gi) {           // the k padding up to a whole number of pipeline steps
            *reinterpret_cast<h8*>(oh + ((long)gi * M + r) * 8) = z;
            *reinterpret_cast<h8*>(oh + ((long)(G + gi) * M + r) * 8) = z;
        }
        rowscale[r] = ldexpf(1.f, -e);
        rbias[r] = (j.fold_bias ? j.fold_bias[r] * fold : 0.f) * SH2_ACT_SCALE;
    }
}

// SH2_FIRST with many input channels (the deep levels' f.4 as a direct 3x3 image: 64 chunks per row; their f.0: 12 / 24): a WAVE takes
// eight rows, lane = (brick-in-pass) * 8 + row, so a row's bricks are spread over eight lanes (one lane per row walked 64 bricks
// twice: 0.67 ms per pack at config E).  Same values as repack_sh2_first: the row maximum is a maximum, the split is per element.
__device__ __forceinline__ void repack_sh2_first_wide(const RepackJob& j, char* packed) {
    const int M = j.Cout, G = j.K, Kp = G * 8;
    _Float16* oh = (_Float16*)(packed + j.out_off);
    float* rowscale = (float*)(packed + j.out_off + (size_t)2 * Kp * M * sizeof(_Float16));
    float* rbias = rowscale + M;
    const int nchunk = (j.Cin + 7) / 8;
    const int lane = threadIdx.x & 63, r8 = lane & 7, gl = lane >> 3;
    for (int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; r0 < M; r0 += gridDim.x * 32) {
        const int r = r0 + r8;
        const bool rv = r < M;
        const float fold = (rv && j.fold_logs) ? expf(j.fold_logs[r] * LOGSCALE) : 1.f;
        const float* row = repack_src(j, packed) + (long)(rv ? r : 0) * j.Cin * 9;
        float mx = 0.f;
        for (int ch = gl; ch < nchunk; ch += 8) {
            float b[72];
            load_brick(row + ch * 72, min(8, j.Cin - ch * 8), rv, b);
#pragma unroll
            for (int i = 0; i < 72; ++i) mx = fmaxf(mx, fabsf(b[i] * fold));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const int e = sh2_row_exponent(mx);
        const float up = ldexpf(1.f, e);
        for (int ch = gl; ch < nchunk; ch += 8) {
            float b[72];
            load_brick(row + ch * 72, min(8, j.Cin - ch * 8), rv, b);
#pragma unroll
            for (int i = 0; i < 72; ++i) b[i] *= fold;      // same rounding order as the row maximum: (w * fold) * 2^e
            if (!rv) continue;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                h8 hi, lo;
                brick_emit(b, tap, up, hi, lo);
                const int gi = ch * 9 + tap;
                *reinterpret_cast<h8*>(oh + ((long)gi * M + r) * 8) = hi;
                *reinterpret_cast<h8*>(oh + ((long)(G + gi) * M + r) * 8) = lo;
            }
        }
        if (!rv) continue;
        h8 z;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) z[k8] = (_Float16)0.f;
        for (int gi = nchunk * 9 + gl; gi < G; gi += 8) {   // the k padding up to a whole number of pipeline steps
            *reinterpret_cast<h8*>(oh + ((long)gi * M + r) * 8) = z;
            *reinterpret_cast<h8*>(oh + ((long)(G + gi) * M + r) * 8) = z;
        }
        if (gl == 0) {
            rowscale[r] = ldexpf(1.f, -e);
            rbias[r] = (j.fold_bias ? j.fold_bias[r] * fold : 0.f) * SH2_ACT_SCALE;
        }
    }
}

// k-permuted group gi of a row of bricks (sh.h sh2_kperm_src): channels 4 kl .. 4 kl + 3 of the 32-block's bricks 2 s and 2 s + 1 --
// two half bricks of 36 contiguous floats each; b[k8 * 9 + tap] as load_brick leaves it
__device__ __forceinline__ void load_brick_kperm(const float* row, int gi, bool valid, float (&b)[72]) {
#pragma unroll
    for (int i = 0; i < 72; ++i) b[i] = 0.f;
    if (!valid) return;
    const int k0 = sh2_kperm_src(gi * 8);            // first of the group's first four channels; the other four are k0 + 8 ...
    const float* s0 = row + (long)k0 * 9;
    const float* s1 = row + (long)(k0 + 8) * 9;
    if ((((size_t)s0 | (size_t)s1) & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const f32x4_t v = reinterpret_cast<const f32x4_t*>(s0)[i], w = reinterpret_cast<const f32x4_t*>(s1)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) { b[i * 4 + q] = v[q]; b[36 + i * 4 + q] = w[q]; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 36; ++i) { b[i] = s0[i]; b[36 + i] = s1[i]; }
    }
}

// SH2_TAIL: a block takes EIGHT output channels (lane = brick-in-pass * 8 + channel, the four waves split the bricks) and emits
// their 9 x 8 rows; row maxima go through LDS.  Rows beyond 9 Cout (the padding to whole row tiles) are zero-filled.
__device__ __forceinline__ void repack_sh2_tail(const RepackJob& j, char* packed) {
    __shared__ float s_mx[4][9][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r8 = lane & 7, gl = lane >> 3;
    const int M = j.Kpad, Kp = j.Cin, ngroups = Kp / 8, rows = 9 * j.Cout;
    _Float16* oh = (_Float16*)(packed + j.out_off);
    float* rowscale = (float*)(packed + j.out_off + (size_t)2 * Kp * M * sizeof(_Float16));
    float* rbias = rowscale + M;
    for (int co0 = blockIdx.x * 8; co0 < j.Cout; co0 += gridDim.x * 8) {
        const int co = co0 + r8;
        const bool rv = co < j.Cout;
        const float* row = repack_src(j, packed) + (long)(rv ? co : 0) * j.Cin * 9;
        float mx[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) mx[t] = 0.f;
        for (int gi = wave * 8 + gl; gi < ngroups; gi += 32) {
            float b[72];
            if (j.kperm) load_brick_kperm(row, gi, rv, b); else load_brick(row + gi * 72, 8, rv, b);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) mx[t] = fmaxf(mx[t], fabsf(b[k8 * 9 + t]));
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 8, 64));
            mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 16, 64));
            mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 32, 64));
            if (gl == 0) s_mx[wave][t][r8] = mx[t];
        }
        __syncthreads();
        int e[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            e[t] = sh2_row_exponent(fmaxf(fmaxf(s_mx[0][t][r8], s_mx[1][t][r8]), fmaxf(s_mx[2][t][r8], s_mx[3][t][r8])));
        __syncthreads();
        for (int gi = wave * 8 + gl; gi < ngroups; gi += 32) {
            float b[72];
            if (j.kperm) load_brick_kperm(row, gi, rv, b); else load_brick(row + gi * 72, 8, rv, b);
            if (rv) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    h8 hi, lo;
                    brick_emit(b, t, ldexpf(1.f, e[t]), hi, lo);
                    const int r = t * j.Cout + co;
                    *reinterpret_cast<h8*>(oh + ((long)gi * M + r) * 8) = hi;
                    *reinterpret_cast<h8*>(oh + ((long)(ngroups + gi) * M + r) * 8) = lo;
                }
            }
        }
        if (wave == 0 && gl == 0 && rv) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                rowscale[t * j.Cout + co] = ldexpf(1.f, -e[t]) * SH2_ACT_INV;
                rbias[t * j.Cout + co] = 0.f;
            }
        }
    }
    const int pad = M - rows;
    h8 z;
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) z[k8] = (_Float16)0.f;
    for (int el = blockIdx.x * 256 + threadIdx.x; el < pad * 2 * ngroups; el += gridDim.x * 256) {
        const int g2 = el / pad, r = rows + (el - g2 * pad);
        *reinterpret_cast<h8*>(oh + ((long)g2 * M + r) * 8) = z;
    }
    for (int r = rows + blockIdx.x * 256 + threadIdx.x; r < M; r += gridDim.x * 256) {
        rowscale[r] = SH2_ACT_INV;
        rbias[r] = 0.f;
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) k_repack_sh2_batched(const RepackJob* __restrict__ jobs, char* packed) {
    const RepackJob j = jobs[blockIdx.y];
    if (j.kind != KIND) return;
    if (KIND == REPACK_SH2_GEMM) repack_sh2_rows<REPACK_SH2_GEMM>(j, packed);
    else if (KIND == REPACK_SH2_FIRST) { if (j.Cin >= 64) repack_sh2_first_wide(j, packed); else repack_sh2_first(j, packed); }
    else repack_sh2_tail(j, packed);
}

// dst[i][o][ks-1-tap] = src[o][i][tap]: a workgroup moves a 32 (o) x 32 (i) tile through LDS -- reads are runs of 32 ks floats of
// one source row, writes runs of 32 ks floats of one destination row
// (jobs come in triples -- f.4, f.2, f.0 of a FlowStep -- with 16 .. 32, 256 and 16 tiles at hidden 512: one launch per member
// (first, stride 3) with that member's tile count, not one grid of 256 x jobs whose two other thirds were 45 k workgroups that
// looked at their job and left: 317 -> 1xx us per pack)
__global__ void __launch_bounds__(256) k_flipT_batched(const FlipJob* __restrict__ jobs, char* packed, int first, int stride) {
    __shared__ float tile[32][32 * 9 + 1];
    const FlipJob j = jobs[first + blockIdx.y * stride];
    const int ti = (j.I + 31) / 32, to = (j.O + 31) / 32;
    if ((int)blockIdx.x >= ti * to) return;
    const int o0 = (blockIdx.x / ti) * 32, i0 = (blockIdx.x % ti) * 32;
    const int ks = j.ks, run = 32 * ks;
    float* dst = reinterpret_cast<float*>(packed + j.dst_off);
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int ol = e / run, r = e - ol * run;
        const int o = o0 + ol, i = i0 + r / ks;
        tile[ol][r] = (o < j.O && i < j.I) ? j.src[((long)o * j.I + i0) * ks + r] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int il = e / run, r = e - il * run;
        const int ol = r / ks, tap = r - ol * ks;
        const int i = i0 + il, o = o0 + ol;
        if (i < j.I && o < j.O) dst[((long)i * j.O + o0) * ks + r] = tile[ol][il * ks + (ks - 1 - tap)];
    }
}

int launch_flipT_batched(const FlipJob* jobs_dev, int n_jobs, const int* max_tiles3, void* packed, hipStream_t s) {
    if (n_jobs == 0) return GLOWHIP_OK;
    GH_REQUIRE(n_jobs % 3 == 0, "flipT: jobs come in triples");
    for (int k = 0; k < 3; ++k) {
        hipLaunchKernelGGL(k_flipT_batched, dim3(std::max(1, max_tiles3[k]), n_jobs / 3), dim3(256), 0, s, jobs_dev, (char*)packed, k, 3);
        GH_LAUNCH_CHECK("k_flipT_batched");
    }
    return GLOWHIP_OK;
}

int launch_pack_batched(const ScaleJob* sj_dev, int n_scale, const RepackJob* rj_dev, const int* n_kind, int tail_blocks, void* packed,
                        hipStream_t s, hipStream_t s_legacy, int first_blocks) {
    if (n_scale > 0) {
        hipLaunchKernelGGL(k_pack_scales_batched, dim3(2, n_scale), dim3(256), 0, s, sj_dev, (char*)packed);
        GH_LAUNCH_CHECK("k_pack_scales_batched");
    }
    // the jobs are sorted by kind (legacy | SH2 GEMM | SH2 FIRST | SH2 TAIL): every kernel gets exactly its own jobs -- a grid over
    // ALL jobs with an early exit per foreign job cost the f.4 image kernel 4 300 empty workgroups at two per CU (its registers)
    const RepackJob* rj = rj_dev;
    if (n_kind[0] > 0) {
        hipLaunchKernelGGL(k_repack_batched, dim3(64, n_kind[0]), dim3(256), 0, s_legacy, rj, (char*)packed);
        GH_LAUNCH_CHECK("k_repack_batched");
    }
    rj += n_kind[0];
    if (n_kind[1] > 0) hipLaunchKernelGGL(k_repack_sh2_batched<REPACK_SH2_GEMM>, dim3(16, n_kind[1]), dim3(256), 0, s, rj, (char*)packed);
    rj += n_kind[1];
    if (n_kind[2] > 0) hipLaunchKernelGGL(k_repack_sh2_batched<REPACK_SH2_FIRST>, dim3(first_blocks > 2 ? first_blocks : 2, n_kind[2]), dim3(256), 0, s, rj, (char*)packed);
    rj += n_kind[2];
    if (n_kind[3] > 0)
        hipLaunchKernelGGL(k_repack_sh2_batched<REPACK_SH2_TAIL>, dim3(tail_blocks > 0 ? tail_blocks : 1, n_kind[3]), dim3(256), 0, s, rj, (char*)packed);
    GH_LAUNCH_CHECK("k_repack_sh2_batched");
    return GLOWHIP_OK;
}

int launch_repack_sh2_gemm(const RepackJob* rj_dev, int n, void* packed, hipStream_t s) {
    if (n <= 0) return GLOWHIP_OK;
    hipLaunchKernelGGL(k_repack_sh2_batched<REPACK_SH2_GEMM>, dim3(16, n), dim3(256), 0, s, rj_dev, (char*)packed);
    GH_LAUNCH_CHECK("k_repack_sh2_batched (after LU)");
    return GLOWHIP_OK;
}

}  // namespace glowhip
