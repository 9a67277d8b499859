// lu.hip -- in-kernel LU of the invertible 1x1 convolution weight (replaces torch.det and
// Tensor.inverse at network/module.py:357,365).  One workgroup per matrix runs Gauss-Jordan
// elimination with partial pivoting in fp64 on the augmented matrix [W | I] held in LDS (C <= 64)
// or in an L2-resident scratch buffer (larger C): log|det W| = sum log|pivot|, W^-1 = right half.
#include <algorithm>

#include "kernels.h"

namespace glowhip {

size_t invconv_scratch_bytes(int C) { return (size_t)C * 2 * C * sizeof(double); }

constexpr int LU_LDS_MAX_C = 64;  // 64 * 128 * 8 B = 64 KiB
constexpr int LU_LDS_ONLY_MAX_C = 128;   // log-det only: 128 * 128 * 8 B = 128 KiB

// matrix -> fp64 working copy, eight requests in flight per trip (a plain copy loop waits out one memory round trip per element)
__device__ __forceinline__ void lu_load_matrix(const float* __restrict__ w, int n, double* A) {
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = w[min(e0 + 256 * k, n - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (e0 + 256 * k < n) A[e0 + 256 * k] = (double)v[k];
    }
}

// Workgroup-wide Gauss-Jordan on A = [W | I] (C x 2C doubles, LDS or global).  Returns log|det W| (thread 192).
// Three barriers per pivot: (1) pivot row and value published by wave 0; (2) rows k and p swapped with the pivot row scaled
// on the way (two rows, one pass); (3) column k eliminated from every other row.  Column k of the left half is never read
// again, so it is not cleared.  log|pivot| is summed by a lane of the LAST wave while the others eliminate.
__device__ double lu_gauss_jordan(const float* __restrict__ w, int C, float* __restrict__ winv, double* A) {
    __shared__ int s_piv;
    __shared__ double s_pivval;
    const int tid = threadIdx.x, W2 = 2 * C;
    __syncthreads();  // A may still be read by a previous use in this workgroup
    {
        // [W | I]: the W half eight requests at a time (clamped addresses, selected values)
        for (int e0 = tid; e0 < C * W2; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = min(e0 + 256 * k, C * W2 - 1), r = e / W2, c = e - r * W2;
                v[k] = w[r * C + min(c, C - 1)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + 256 * k;
                if (e >= C * W2) continue;
                const int r = e / W2, c = e - r * W2;
                A[e] = (c < C) ? (double)v[k] : ((c - C == r) ? 1.0 : 0.0);
            }
        }
    }
    double logdet = 0.0;   // meaningful in thread 192
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        // partial pivoting: first wave finds argmax |A[r][k]|, r >= k (ties -> lowest row: deterministic)
        if (tid < 64) {
            double best = -1.0;
            int bi = k;
            for (int r = k + tid; r < C; r += 64) {
                double v = fabs(A[r * W2 + k]);
                if (v > best) { best = v; bi = r; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                double ov = __shfl_down(best, o, 64);
                int oi = __shfl_down(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (tid == 0) { s_piv = bi; s_pivval = A[bi * W2 + k]; }
        }
        __syncthreads();
        const int pr = s_piv;
        const double piv = s_pivval;
        if (tid == 192) logdet += log(fabs(piv));
        // rows k <-> pr, the new row k divided by the pivot (each column owned by one thread)
        for (int c = tid; c < W2; c += 256) {
            const double tk = A[pr * W2 + c], tp = A[k * W2 + c];
            A[k * W2 + c] = tk / piv;
            if (pr != k) A[pr * W2 + c] = tp;
        }
        __syncthreads();
        // eliminate column k from every other row: A[r][c] -= A[r][k] * A[k][c] for c != k (column k itself is dead)
        {
            int r = tid / W2, c = tid - r * W2;
            const int dr = 256 / W2, dc = 256 - dr * W2;
            for (int e = tid; e < C * W2; e += 256) {
                if (r != k && c != k) A[e] = fma(-A[r * W2 + k], A[k * W2 + c], A[e]);
                r += dr; c += dc;
                if (c >= W2) { c -= W2; ++r; }
            }
        }
        __syncthreads();
    }
    if (winv)
        for (int e = tid; e < C * C; e += 256) {
            int r = e / C, c = e - r * C;
            winv[e] = (float)A[r * W2 + C + c];
        }
    __shared__ double s_logdet;
    if (tid == 192) s_logdet = logdet;
    __syncthreads();
    return s_logdet;
}

// log|det W| ONLY (what encode / glow_forward need; W^-1 is for decode and the backward pass): LU with partial pivoting on the
// C x C matrix alone, the rank-1 update restricted to the trailing (C-k-1)^2 block -- C^3/3 element updates instead of
// Gauss-Jordan's 2 C^3 on the augmented matrix.  For C = 384 (config E's last level) that is the difference between 150 ms
// and a few ms per pack.  A: C*C doubles (LDS for C <= 128, global scratch above).
__device__ double lu_logdet_only(const float* __restrict__ w, int C, double* A) {
    __shared__ int s_piv2;
    __shared__ double s_pivval2;
    const int tid = threadIdx.x;
    __syncthreads();
    lu_load_matrix(w, C * C, A);
    double logdet = 0.0;   // meaningful in thread 192
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        if (tid < 64) {
            double best = -1.0;
            int bi = k;
            for (int r = k + tid; r < C; r += 64) {
                double v = fabs(A[r * C + k]);
                if (v > best) { best = v; bi = r; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                double ov = __shfl_down(best, o, 64);
                int oi = __shfl_down(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (tid == 0) { s_piv2 = bi; s_pivval2 = A[bi * C + k]; }
        }
        __syncthreads();
        const int pr = s_piv2;
        const double piv = s_pivval2;
        if (tid == 192) logdet += log(fabs(piv));
        const int rem = C - k - 1;
        if (rem == 0) break;
        // swap the trailing parts of rows k and pr (columns > k), and turn column k below the diagonal into multipliers
        if (pr != k)
            for (int c = k + 1 + tid; c < C; c += 256) {
                const double t = A[pr * C + c];
                A[pr * C + c] = A[k * C + c];
                A[k * C + c] = t;
            }
        for (int r = k + 1 + tid; r < C; r += 256) {
            const double akk = (r == pr && pr != k) ? A[k * C + k] : A[r * C + k];   // row pr now holds old row k
            A[r * C + k] = akk / piv;
        }
        __syncthreads();
        // trailing update: A[r][c] -= l[r] * A[k][c], r, c > k; threads as a 16 x 16 grid over (r, c) -- no index division (a
        // flat index cost a 64-bit division per element, more than the update itself)
        {
            const int tx = tid & 15, ty = tid >> 4;
            for (int r = k + 1 + ty; r < C; r += 16) {
                const double l = A[r * C + k];
                for (int c = k + 1 + tx; c < C; c += 16) A[r * C + c] = fma(-l, A[k * C + c], A[r * C + c]);
            }
        }
        __syncthreads();
    }
    __shared__ double s_logdet2;
    if (tid == 192) s_logdet2 = logdet;
    __syncthreads();
    return s_logdet2;
}

// log|det W| for matrices too large for LDS (128 < C): BLOCKED right-looking LU with partial pivoting, one workgroup per matrix.
// A (C x C doubles) lives in an L2-resident scratch buffer; per panel of NB columns
//   (a) the panel (all rows below the diagonal block, NB columns) is copied to LDS and factored there (pivot search, row swap,
//       scale + rank-1 update: three barriers per column, LDS only),
//   (b) its row swaps are applied to the columns right of the panel (a thread owns whole columns: no barrier),
//   (c) column block by column block (CB columns): U12 = L11^-1 A12 by forward substitution (a thread per column, the result kept
//       in LDS), then the trailing update A22 -= L21 U12 with 4 x 4 register tiles (L21 and U12 both read from LDS).
// The unblocked version above walks the trailing matrix in global memory once per COLUMN with a 64-bit division per element:
// 33 ms for config E's C = 384 matrices, half of that config's forward step.  This one reads / writes it once per PANEL.
constexpr int LU_NB = 32, LU_CB = 64;
__host__ __device__ inline size_t lu_blocked_lds_bytes(int C) {
    return ((size_t)C * (LU_NB + 1) + (size_t)LU_NB * (LU_CB + 1)) * sizeof(double) + (LU_NB + 4) * sizeof(int);
}
constexpr int LU_BLOCKED_MAX_C = 448;    // panel of 448 rows: 118 KiB + 16.6 KiB U block

__device__ double lu_logdet_blocked(const float* __restrict__ w, int C, double* __restrict__ A, double* lds) {
    constexpr int NB = LU_NB, CB = LU_CB, PS = NB + 1, US = CB + 1;    // padded LDS row strides (doubles)
    double* P = lds;                             // panel: [rows below k0][PS]
    double* U = lds + (size_t)C * PS;            // U12 column block: [NB][US]
    int* piv = reinterpret_cast<int*>(U + NB * US);   // [NB] pivot rows (relative to k0), then [1] scratch
    __shared__ int s_p;
    __shared__ double s_pv;
    const int tid = threadIdx.x;
    __syncthreads();
    lu_load_matrix(w, C * C, A);
    double logdet = 0.0;   // meaningful in thread 192
    __syncthreads();
    for (int k0 = 0; k0 < C; k0 += NB) {
        const int nb = min(NB, C - k0), m = C - k0;
        // (a) panel -> LDS, factor
        for (int e = tid; e < m * nb; e += 256) {
            const int i = e / nb, j = e - i * nb;
            P[i * PS + j] = A[(long)(k0 + i) * C + k0 + j];
        }
        __syncthreads();
        for (int j = 0; j < nb; ++j) {
            if (tid < 64) {
                double best = -1.0;
                int bi = j;
                for (int r = j + tid; r < m; r += 64) {
                    const double v = fabs(P[r * PS + j]);
                    if (v > best) { best = v; bi = r; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double ov = __shfl_down(best, o, 64);
                    const int oi = __shfl_down(bi, o, 64);
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                if (tid == 0) { s_p = bi; s_pv = P[bi * PS + j]; piv[j] = bi; }
            }
            __syncthreads();
            const int pr = s_p;
            const double pv = s_pv;
            if (tid == 192) logdet += log(fabs(pv));
            if (pr != j && tid < nb) {           // swap rows j and pr of the panel (all nb columns: the left ones are the L part)
                const double t = P[pr * PS + tid];
                P[pr * PS + tid] = P[j * PS + tid];
                P[j * PS + tid] = t;
            }
            __syncthreads();
            // rows below the diagonal: multiplier, then the rank-1 update of the panel's remaining columns (a thread owns a row)
            for (int r = j + 1 + tid; r < m; r += 256) {
                const double l = P[r * PS + j] / pv;
                P[r * PS + j] = l;
                for (int c = j + 1; c < nb; ++c) P[r * PS + c] = fma(-l, P[j * PS + c], P[r * PS + c]);
            }
            __syncthreads();
        }
        const int n2 = C - k0 - nb;              // columns right of the panel
        if (n2 <= 0) break;
        // (b) the panel's row swaps on the right part, in order; a thread owns its columns
        for (int c = k0 + nb + tid; c < C; c += 256) {
            for (int j = 0; j < nb; ++j) {
                const int pr = piv[j];
                if (pr != j) {
                    const double t = A[(long)(k0 + pr) * C + c];
                    A[(long)(k0 + pr) * C + c] = A[(long)(k0 + j) * C + c];
                    A[(long)(k0 + j) * C + c] = t;
                }
            }
        }
        __syncthreads();
        // (c) column blocks of the right part
        const int m2 = m - nb;                   // rows below the diagonal block
        for (int c0 = k0 + nb; c0 < C; c0 += CB) {
            const int cb = min(CB, C - c0);
            // U12 block: forward substitution with the unit lower triangle L11 (panel rows 0..nb-1)
            if (tid < cb) {
                double x[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) x[i] = i < nb ? A[(long)(k0 + i) * C + c0 + tid] : 0.0;
#pragma unroll
                for (int i = 1; i < NB; ++i) {
                    if (i < nb) {
                        double v = x[i];
#pragma unroll
                        for (int t = 0; t < i; ++t) v = fma(-P[i * PS + t], x[t], v);
                        x[i] = v;
                    }
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) U[i * US + tid] = x[i];
            }
            __syncthreads();
            // trailing update of this column block: 4 x 4 register tiles
            const int tr = (m2 + 3) / 4, tc = (cb + 3) / 4;
            for (int tix = tid; tix < tr * tc; tix += 256) {
                const int ti = tix / tc, tj = tix - ti * tc;
                const int r0 = nb + ti * 4, cc0 = tj * 4;           // panel-relative row, block-relative column
                double acc[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
                for (int t = 0; t < nb; ++t) {
                    double l[4], u[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) l[a] = P[min(r0 + a, m - 1) * PS + t];
#pragma unroll
                    for (int b = 0; b < 4; ++b) u[b] = U[t * US + min(cc0 + b, cb - 1)];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] = fma(l[a], u[b], acc[a][b]);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (r0 + a < m && cc0 + b < cb) {
                            const long idx = (long)(k0 + r0 + a) * C + c0 + cc0 + b;
                            A[idx] -= acc[a][b];
                        }
            }
            __syncthreads();
        }
    }
    __shared__ double s_logdet3;
    if (tid == 192) s_logdet3 = logdet;
    __syncthreads();
    return s_logdet3;
}

template <bool USE_LDS>
__global__ void __launch_bounds__(256) k_invconv_prepare(const float* __restrict__ w, int C, float* __restrict__ winv,
                                                         float* __restrict__ logabsdet, double* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) double lds_aug[];
    const double ld = lu_gauss_jordan(w, C, winv, USE_LDS ? lds_aug : scratch);
    if (threadIdx.x == 0 && logabsdet) logabsdet[0] = (float)ld;
}

// Batched form used by glowhip_plan_pack: one workgroup per FlowStep; also produces the step's
// data-independent log-det term  konst = 3*sum(actnorm.logs)*HW + log|det W|*HW  (network/module.py:76-82,356-357).
__global__ void __launch_bounds__(256) k_step_prepare_batched(const StepPrepJob* __restrict__ jobs, char* packed, int want_inverse) {
    extern __shared__ __attribute__((aligned(16))) double lds_aug[];
    __shared__ double red[4];
    const StepPrepJob j = jobs[blockIdx.x];
    double lad = 0.0;
    if (j.w) {
        if (want_inverse) {
            double* A = j.C <= LU_LDS_MAX_C ? lds_aug : (double*)(packed + j.scratch_off);
            lad = lu_gauss_jordan(j.w, j.C, (float*)(packed + j.winv_off), A);
        } else if (j.C > LU_LDS_ONLY_MAX_C && j.C <= LU_BLOCKED_MAX_C) {   // forward only, too large for LDS: blocked LU
            lad = lu_logdet_blocked(j.w, j.C, (double*)(packed + j.scratch_off), lds_aug);
        } else {      // forward only: log|det W| without the inverse (C^3/3 instead of 2 C^3 element updates)
            double* A = j.C <= LU_LDS_ONLY_MAX_C ? lds_aug : (double*)(packed + j.scratch_off);
            lad = lu_logdet_only(j.w, j.C, A);
        }
    }
    double acc = 0.0;
    for (int k = threadIdx.x; k < j.C; k += 256) acc += (double)(j.an_logs[k] * LOGSCALE);
    const double tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) {
        if (j.w) *(float*)(packed + j.logabsdet_off) = (float)lad;
        *(double*)(packed + j.konst_off) = tot * (double)j.HW + (j.w ? (double)(float)lad * (double)j.HW : 0.0);
    }
}

// ---- log|det W| of the SMALL matrices of a plan (C = 12 / 24 / 48: every FlowStep of the 64 x 64 model), one WAVE per matrix with
// the matrix in REGISTERS: lane r holds row r as C doubles.  The workgroup-wide LU above pays three barriers and an LDS round trip
// per pivot -- 74 us per pack for the 96 matrices of config B, every step of a forward that re-derives its weights.  Measured
// (rocprofv3, scripts/time_pack.py): 64 us for this form -- a modest gain: what bounds both is the serial chain of a pivot step
// (cross-lane argmax, one fp64 division, the pivot row lane by lane), ~2 800 cycles per pivot here.  Same algorithm, same operations in the same order (partial pivoting with the
// lowest-row tie-break, multipliers l = a / pivot, trailing update fma(-l, pivot row, a), sum of log|pivot| in pivot order): the
// result is the workgroup kernel's, bit for bit.  Rows are never moved: a lane keeps the POSITION its row currently has, a swap
// exchanges two positions, and the pivot row reaches the other lanes by v_readlane (its lane is wave-uniform).
template <int C>
__device__ __forceinline__ double lu_logdet_wave(const float* __restrict__ w, const int lane) {
    double a[C];
    const int row = lane < C ? lane : C - 1;
#pragma unroll
    for (int c = 0; c < C; ++c) a[c] = (double)w[row * C + c];
    int pos = lane;
    double mypiv = 1.0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        double best = (lane < C && pos >= k) ? fabs(a[k]) : -1.0;
        int bi = pos;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {       // (butterfly: every lane ends with the same winner -- the largest value, lowest position among equals)
            const double ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const int pr = __builtin_amdgcn_readfirstlane(bi);
        const unsigned long long owner = __ballot(lane < C && pos == pr);
        const int P = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(owner));      // the lane that holds the pivot row
        auto from_pivot = [&](double v) {
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double_as_longlong(v), P);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)__double_as_longlong(v) >> 32), P);
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        const double piv = from_pivot(a[k]);
        if (lane == k) mypiv = piv;              // (the logarithms are taken once, all lanes at a time, behind the elimination: log() is a
                                                 // few hundred dependent instructions, and in front of every step it was a third of the kernel)
        const bool was_k = pos == k;
        if (lane == P) pos = k;
        else if (was_k) pos = pr;
        if (k == C - 1) break;
        const bool below = lane < C && pos > k;
        const double l = a[k] / piv;
#pragma unroll
        for (int c = k + 1; c < C; ++c) {
            const double rk = from_pivot(a[c]);
            if (below) a[c] = fma(-l, rk, a[c]);
        }
    }
    // sum of log|pivot| in pivot order, as the workgroup kernel adds them
    const double lg = log(fabs(mypiv));
    double logdet = 0.0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double_as_longlong(lg), k);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)__double_as_longlong(lg) >> 32), k);
        logdet += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return logdet;
}

// one wave per FlowStep (four per workgroup); launch_step_prepare_batched selects it when every job of the plan is one it takes
__global__ void __launch_bounds__(256) k_step_prepare_small(const StepPrepJob* __restrict__ jobs, int n, char* packed) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const StepPrepJob j = jobs[i];
    double lad = 0.0;
    if (j.w) lad = j.C == 12 ? lu_logdet_wave<12>(j.w, lane) : (j.C == 24 ? lu_logdet_wave<24>(j.w, lane) : lu_logdet_wave<48>(j.w, lane));
    // (the same tree as the workgroup kernel's block_sum: its waves 1 - 3 add exact zeros for C <= 64)
    const double tot = wave_sum(lane < j.C ? (double)(j.an_logs[lane] * LOGSCALE) : 0.0);
    if (lane == 0) {
        if (j.w) *(float*)(packed + j.logabsdet_off) = (float)lad;
        *(double*)(packed + j.konst_off) = tot * (double)j.HW + (j.w ? (double)(float)lad * (double)j.HW : 0.0);
    }
}
bool step_prepare_small_takes(const StepPrepJob& j) { return j.C <= 64 && (!j.w || j.C == 12 || j.C == 24 || j.C == 48); }

// plan-wide total of the per-step terms: fetched in parallel, summed in layer order by one thread (deterministic)
__global__ void __launch_bounds__(256) k_sum_konst(const StepPrepJob* __restrict__ jobs, int n, char* packed) {
    __shared__ double v[256];
    double t = 0.0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        v[threadIdx.x] = i < n ? *(const double*)(packed + jobs[i].konst_off) : 0.0;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int q = 0; q < 256 && base + q < n; ++q) t += v[q];
        __syncthreads();
    }
    if (threadIdx.x == 0) *(double*)packed = t;
}

int launch_step_prepare_batched(const StepPrepJob* jobs_dev, int n, int max_lds_c, void* packed, hipStream_t s, int want_inverse,
                                int max_c, int all_small) {
    if (n == 0) {
        (void)hipMemsetAsync(packed, 0, sizeof(double), s);
        return GLOWHIP_OK;
    }
    if (!want_inverse && all_small) {      // log|det W| only and every matrix is 12 / 24 / 48 wide: one wave per matrix, in registers
        hipLaunchKernelGGL(k_step_prepare_small, dim3((n + 3) / 4), dim3(256), 0, s, jobs_dev, n, (char*)packed);
        GH_LAUNCH_CHECK("k_step_prepare_small");
        hipLaunchKernelGGL(k_sum_konst, dim3(1), dim3(256), 0, s, jobs_dev, n, (char*)packed);
        GH_LAUNCH_CHECK("k_sum_konst");
        return GLOWHIP_OK;
    }
    size_t lds = invconv_scratch_bytes(max_lds_c > 0 ? max_lds_c : 1);
    if (!want_inverse && max_c > LU_LDS_MAX_C) {    // the log-det-only factorisation of matrices up to 128 x 128 runs in LDS,
                                                    // the blocked one above that keeps a panel + a U block there
        lds = std::max(lds, (size_t)std::min(max_c, LU_LDS_ONLY_MAX_C) * std::min(max_c, LU_LDS_ONLY_MAX_C) * sizeof(double));
        if (max_c > LU_LDS_ONLY_MAX_C) lds = std::max(lds, lu_blocked_lds_bytes(std::min(max_c, LU_BLOCKED_MAX_C)));
    }
    if (lds > 32 * 1024)
        (void)hipFuncSetAttribute((const void*)k_step_prepare_batched, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_step_prepare_batched, dim3(n), dim3(256), lds, s, jobs_dev, (char*)packed, want_inverse);
    GH_LAUNCH_CHECK("k_step_prepare_batched");
    hipLaunchKernelGGL(k_sum_konst, dim3(1), dim3(256), 0, s, jobs_dev, n, (char*)packed);
    GH_LAUNCH_CHECK("k_sum_konst");
    return GLOWHIP_OK;
}

int launch_invconv_prepare(const float* w, int C, float* winv, float* logabsdet, void* scratch, hipStream_t s) {
    GH_REQUIRE(C > 0 && C <= 1024, "invconv_prepare: C=%d unsupported", C);
    if (C <= LU_LDS_MAX_C) {
        size_t lds = invconv_scratch_bytes(C);
        if (lds > 32 * 1024)
            (void)hipFuncSetAttribute((const void*)k_invconv_prepare<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        hipLaunchKernelGGL(k_invconv_prepare<true>, dim3(1), dim3(256), lds, s, w, C, winv, logabsdet, (double*)nullptr);
    } else {
        GH_REQUIRE(scratch != nullptr, "invconv_prepare: scratch buffer required for C=%d", C);
        hipLaunchKernelGGL(k_invconv_prepare<false>, dim3(1), dim3(256), 0, s, w, C, winv, logabsdet, (double*)scratch);
    }
    GH_LAUNCH_CHECK("k_invconv_prepare");
    return GLOWHIP_OK;
}

}  // namespace glowhip
