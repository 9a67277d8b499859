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
    } else if (j.kind == REPACK_SH_GEMM) {
        // split-half GEMM image (sh.h): half [plane][K/8][M][8] of w[o][k] * exp(3 logs[o]), then M floats bias * exp(3 logs)
        const long total = (long)j.K * j.Cout;
        _Float16* oh = (_Float16*)out;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int k8 = (int)(e & 7);
            const int o = (int)((e >> 3) % j.Cout);
            const int k = (int)((e >> 3) / j.Cout) * 8 + k8;
            const float wv = (j.transposed ? j.w[(long)k * j.Cout + o] : j.w[(long)o * j.K + k]) *
                             (j.fold_logs ? expf(j.fold_logs[o] * LOGSCALE) : 1.f);
            _Float16 hi, lo;
            sh_split(wv, hi, lo);
            oh[e] = hi;
            oh[total + e] = lo;
        }
        float* fb = (float*)((char*)out + align_up_dev((size_t)2 * total * sizeof(_Float16), 16));
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < j.Cout; o += (long)gridDim.x * 256)
            fb[o] = j.fold_logs ? j.fold_bias[o] * expf(j.fold_logs[o] * LOGSCALE) : 0.f;
    } else if (j.kind == REPACK_SH_FIRST) {
        // split-half f.0 image (first_sh.hip): half [plane][G][Cout][8]; group g = tap*nchunk + chunk, channel ci = chunk*8 + k8;
        // zero for padded groups / channels; ActNorm scale folded; then Cout floats bias * exp(3 logs).  j.K = G
        const int nchunk = (j.Cin + 7) / 8, G = j.K;
        const long total = (long)G * j.Cout * 8;
        _Float16* oh = (_Float16*)out;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int k8 = (int)(e & 7);
            const int o = (int)((e >> 3) % j.Cout);
            const int g = (int)((e >> 3) / j.Cout);
            const int tap = g / nchunk, ci = (g - tap * nchunk) * 8 + k8;
            float wv = 0.f;
            if (g < 9 * nchunk && ci < j.Cin)
                wv = j.w[((long)o * j.Cin + ci) * 9 + tap] * (j.fold_logs ? expf(j.fold_logs[o] * LOGSCALE) : 1.f);
            _Float16 hi, lo;
            sh_split(wv, hi, lo);
            oh[e] = hi;
            oh[total + e] = lo;
        }
        float* fb = (float*)((char*)out + align_up_dev((size_t)2 * total * sizeof(_Float16), 16));
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < j.Cout; o += (long)gridDim.x * 256)
            fb[o] = j.fold_logs ? j.fold_bias[o] * expf(j.fold_logs[o] * LOGSCALE) : 0.f;
    } else if (j.kind == REPACK_SH_TAIL) {
        // split-half tail image (tail_sh.hip): half [group][plane][Cin/8][Mpad][8], row m = tap*Cg + co_in_group, zero rows
        // m >= 9*Cg; j.MT = channel groups, j.Kpad = Mpad
        const int Mpad = j.Kpad, groups = j.MT, Cg = j.Cout / groups;
        const long per_plane = (long)j.Cin * Mpad, total = per_plane * groups;
        _Float16* oh = (_Float16*)out;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int g = (int)(e / per_plane);
            const long el = e - (long)g * per_plane;
            const int k8 = (int)(el & 7);
            const int m = (int)((el >> 3) % Mpad);
            const int k = (int)((el >> 3) / Mpad) * 8 + k8;
            const int tap = m / Cg, co = g * Cg + (m - tap * Cg);
            const float wv = m < 9 * Cg ? j.w[((long)co * j.Cin + k) * 9 + tap] : 0.f;
            _Float16 hi, lo;
            sh_split(wv, hi, lo);
            oh[(long)g * 2 * per_plane + el] = hi;
            oh[(long)g * 2 * per_plane + per_plane + el] = lo;
        }
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
__global__ void __launch_bounds__(256) k_repack_sh2_batched(const RepackJob* __restrict__ jobs, char* packed) {
    const RepackJob j = jobs[blockIdx.y];
    if (j.kind < REPACK_SH2_GEMM) return;
    // a wave takes EIGHT consecutive rows: lane = (k group within the pass) * 8 + row, so the eight 16-byte groups of one k group
    // are 128 contiguous bytes of the image (a single row per wave wrote 16 bytes every M * 16)
    const int lane = threadIdx.x & 63, r8 = lane & 7, gl = lane >> 3;
    const int M = j.kind == REPACK_SH2_TAIL ? j.Kpad : j.Cout;
    const int Kp = j.kind == REPACK_SH2_FIRST ? j.K * 8 : j.Cin;
    const int ngroups = Kp / 8;
    _Float16* oh = (_Float16*)(packed + j.out_off);
    float* rowscale = (float*)(packed + j.out_off + (size_t)2 * Kp * M * sizeof(_Float16));
    float* rbias = rowscale + M;
    const int nchunk = (j.Cin + 7) / 8;
    for (int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; r0 < M; r0 += gridDim.x * 32) {
        const int r = r0 + r8;
        const bool rv = r < M;
        float fold = 1.f;
        if (rv && j.kind != REPACK_SH2_TAIL && j.fold_logs) fold = expf(j.fold_logs[r] * LOGSCALE);
        auto value = [&](int k) -> float {
            if (!rv) return 0.f;
            if (j.kind == REPACK_SH2_GEMM) return j.w[(long)r * j.Cin + k] * fold;
            if (j.kind == REPACK_SH2_FIRST) {
                const int gi = k >> 3, k8 = k & 7;
                const int ch = gi / 9, tap = gi - ch * 9, ci = ch * 8 + k8;
                return (ch < nchunk && ci < j.Cin) ? j.w[((long)r * j.Cin + ci) * 9 + tap] * fold : 0.f;
            }
            const int tap = r / j.Cout, co = r - tap * j.Cout;
            return r < 9 * j.Cout ? j.w[((long)co * j.Cin + k) * 9 + tap] : 0.f;
        };
        constexpr int KEEP = 8;               // passes whose values stay in registers (Kp <= 512); beyond that they are re-read
        float keep[KEEP][8];
        float mx = 0.f;
        // f.2 rows are contiguous in memory: a lane's 8 values are two 16-byte loads (the other kinds gather with stride 9)
        const bool vec = j.kind == REPACK_SH2_GEMM && (j.Cin & 7) == 0;
#pragma unroll
        for (int it = 0; it < KEEP; ++it) {           // compile-time indices: `keep` stays in registers
            const int gi = gl + 8 * it;
            if (vec) {
                f32x4_t lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = lo4;
                if (rv && gi < ngroups) {
                    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(j.w + (long)r * j.Cin + gi * 8);
                    lo4 = src[0]; hi4 = src[1];
                }
#pragma unroll
                for (int k8 = 0; k8 < 4; ++k8) { keep[it][k8] = lo4[k8] * fold; keep[it][4 + k8] = hi4[k8] * fold; }
            } else {
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) keep[it][k8] = gi < ngroups ? value(gi * 8 + k8) : 0.f;
            }
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) mx = fmaxf(mx, fabsf(keep[it][k8]));
        }
        for (int gi = gl + 8 * KEEP; gi < ngroups; gi += 8)
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) mx = fmaxf(mx, fabsf(value(gi * 8 + k8)));
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
        auto emit = [&](int gi, const float (&v)[8]) {
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
        };
#pragma unroll
        for (int it = 0; it < KEEP; ++it)
            if (gl + 8 * it < ngroups) emit(gl + 8 * it, keep[it]);
        for (int gi = gl + 8 * KEEP; gi < ngroups; gi += 8) {
            float v[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) v[k8] = value(gi * 8 + k8);
            emit(gi, v);
        }
        if (gl == 0 && rv) {
            if (j.kind == REPACK_SH2_TAIL) {
                rowscale[r] = ldexpf(1.f, -e) * SH2_ACT_INV;
                rbias[r] = 0.f;
            } else {
                rowscale[r] = ldexpf(1.f, -e);
                rbias[r] = (j.fold_bias ? j.fold_bias[r] * fold : 0.f) * SH2_ACT_SCALE;
            }
        }
    }
}

int launch_pack_batched(const ScaleJob* sj_dev, int n_scale, const RepackJob* rj_dev, int n_repack, void* packed,
                        hipStream_t s) {
    if (n_scale > 0) {
        hipLaunchKernelGGL(k_pack_scales_batched, dim3(2, n_scale), dim3(256), 0, s, sj_dev, (char*)packed);
        GH_LAUNCH_CHECK("k_pack_scales_batched");
    }
    if (n_repack > 0) {
        hipLaunchKernelGGL(k_repack_batched, dim3(64, n_repack), dim3(256), 0, s, rj_dev, (char*)packed);
        GH_LAUNCH_CHECK("k_repack_batched");
        hipLaunchKernelGGL(k_repack_sh2_batched, dim3(16, n_repack), dim3(256), 0, s, rj_dev, (char*)packed);
        GH_LAUNCH_CHECK("k_repack_sh2_batched");
    }
    return GLOWHIP_OK;
}

}  // namespace glowhip
