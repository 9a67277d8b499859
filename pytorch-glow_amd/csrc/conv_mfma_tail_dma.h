// conv_mfma_tail_dma.h -- shared pieces of the tail convolution: geometry helper, fused epilogue, and the LDS-DMA
// kernel template (instantiated in conv_mfma_tail_dma_a.hip / _b.hip so the build parallelises).
#pragma once
#include <type_traits>

#include "conv_mfma.h"

namespace glowhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int tail_chs(int TR, int W) {   // channel stride == 16 (mod 32) floats
    return (TR + 2) * (W + 8) + ((16 - ((TR + 2) * (W + 8)) % 32) + 32) % 32;
}

__device__ __forceinline__ float gauss_logp1_(float mean, float logs, float x) {
    const float d = x - mean;
    return -0.5f * (LOG_2PI_F + 2.0f * logs + (d * d) / expf(2.0f * logs));
}

// Epilogue shared by the tail kernels: (conv + bias) * exp(3 logs), then the coupling / prior arithmetic on the
// lane-local {even, odd} channel pairs.  Returns this lane's contribution to the per-sample log-det term.
template <int MT, int NTW>
__device__ __forceinline__ double tail_epilogue(const TailConvArgs& a, const f32x4 (&acc)[MT][NTW], long n, int p0,
                                                int HW, int wn, int mt0, int paired, int lane) {
    double ld = 0.0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mrow = (mt0 + m) * 16 + (lane >> 4) * 4;
        float hb[4], hs[4];
        int oc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            oc[r] = tail_row_channel(mrow + r, a.Cout, paired);
            hb[r] = (oc[r] >= 0 && a.bias) ? a.bias[oc[r]] : 0.f;
            hs[r] = oc[r] >= 0 ? (a.scale ? a.scale[oc[r]] : 1.f) : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int p = p0 + (wn * NTW + nt) * 16 + (lane & 15);
            float hv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (acc[m][nt][r] + hb[r]) * hs[r];
            if (a.hout) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (oc[r] >= 0) a.hout[(n * a.Cout + oc[r]) * HW + p] = hv[r];
            }
            if (paired) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (oc[e] < 0) continue;
                    const int c = oc[e] >> 1;  // coupling channel
                    const float A_ = hv[e], B_ = hv[2 + e];
                    const long zi = n * a.z2_in_bs + (long)c * HW + p;
                    const long zo = n * a.z2_out_bs + (long)c * HW + p;
                    if (a.mode == TAIL_AFFINE_FWD) {
                        const float sc = sigmoidf_(B_ + 2.0f);
                        a.z2_out[zo] = (a.z2_in[zi] + A_) * sc;
                        ld += (double)logf(sc);
                    } else if (a.mode == TAIL_AFFINE_REV) {
                        const float sc = sigmoidf_(B_ + 2.0f);
                        a.z2_out[zo] = a.z2_in[zi] / sc - A_;
                        ld -= (double)logf(sc);
                    } else if (a.mode == TAIL_SPLIT_FWD) {
                        ld += (double)gauss_logp1_(A_, B_, a.z2_in[zi]);
                    } else {  // TAIL_SPLIT_REV
                        a.z2_out[zo] = A_ + expf(B_) * a.z2_in[zi];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (oc[r] < 0) continue;
                    const long zo = n * a.z2_out_bs + (long)oc[r] * HW + p;
                    if (a.mode == TAIL_PLAIN) a.z2_out[zo] = hv[r];
                    else {
                        const float z2 = a.z2_in[n * a.z2_in_bs + (long)oc[r] * HW + p];
                        a.z2_out[zo] = a.mode == TAIL_ADD_FWD ? z2 + hv[r] : z2 - hv[r];
                    }
                }
            }
        }
    }
    return ld;
}

// ================================================================================================
// k_conv_tail_dma: the same computation with both operand images streamed by the LDS-DMA path
// (global_load_lds_dwordx4) into a 3-stage ring of 16-channel chunks -- no staging VGPRs, no ds_write bursts,
// one raw barrier per chunk placed MID-chunk (the matrix pipe still has queued MFMAs), refill pieces issued one
// per k-step behind an executing MFMA.  The X window of a stage is written as a LINEAR LDS image (a DMA piece
// fills 1 KiB = 64 consecutive 16-byte slots); slots that are padding columns / rows outside the image read a
// 16-byte zero block in global memory instead of being skipped, so the zero padding is re-established by the
// DMA itself.  Needs Cin % 16 == 0 and compile-time geometry (WFIX).
// ================================================================================================
template <int MT, int NTW, int WN, int WK, int WFIX>
__global__ void __launch_bounds__(256) k_conv_tail_dma(TailConvArgs a, int paired) {
    static_assert(WN * WK == 4 && WFIX > 0, "4 waves per block, compile-time width");
    constexpr int TP = 16 * NTW * WN, CK = 16, NST = 3;
    constexpr int RS = WFIX + 8, TR = TP / WFIX, W4 = WFIX / 4, CHS = tail_chs(TR, WFIX);
    constexpr int X_FLOATS = (CK * CHS + 255) / 256 * 256, PX = X_FLOATS / 256;
    constexpr int A_FLOATS = (CK / 4) * 9 * MT * 64, PA = A_FLOATS / 256;
    constexpr int P = PX + PA, PPW = (P + 3) / 4;          // DMA pieces per stage / per wave (dummies pad to PPW)
    constexpr int STAGE = X_FLOATS + A_FLOATS;
    constexpr int CPW = (CK / 4) / WK, STEPS = 9 * CPW, MID = STEPS / 2;
    static_assert(PPW <= STEPS - MID, "not enough k-steps to spread the refill");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // NST*STAGE | 256 dummy | 8 floats (red)
    float* dummy = lds + NST * STAGE;
    double* red = reinterpret_cast<double*>(dummy + 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % WN, wk = wid / WN;
    const int HW = a.H * a.W;
    const long gp0 = (long)blockIdx.x * TP;
    const long n = gp0 / HW;
    const int p0 = (int)(gp0 - n * HW);
    const int y0 = p0 / a.W;
    const float* xin = a.x + n * a.x_bs;
    const int mt_total = gridDim.y * MT, mt0 = blockIdx.y * MT;
    const int nchunks = a.Cin / CK;

    // per-lane source of every DMA piece this wave issues (piece q = wid + 4*i of each stage)
    const float* src0[PPW];
    int stride[PPW];     // floats per chunk
    int ldso[PPW];       // LDS float offset inside the stage (wave-uniform), or -1 for the dummy area
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wid + 4 * i;
        if (q < PX) {
            const int slot = q * 64 + lane;
            const int c = slot / (CHS / 4), rem = slot - c * (CHS / 4);
            const int r = rem / (RS / 4), col4 = rem - r * (RS / 4);
            const int yy = y0 - 1 + r;
            const bool ok = c < CK && r < TR + 2 && col4 >= 1 && col4 <= W4 && yy >= 0 && yy < a.H;
            src0[i] = ok ? xin + (long)c * HW + (long)yy * a.W + (col4 - 1) * 4 : a.zeros;
            stride[i] = ok ? CK * HW : 0;
            ldso[i] = q * 256;
        } else if (q < P) {
            const int f = (q - PX) * 256 + lane * 4;
            const int row = f / (MT * 64), col = f - row * (MT * 64);
            src0[i] = a.wp + ((long)row * mt_total + mt0) * 64 + col;
            stride[i] = (CK / 4) * 9 * mt_total * 64;
            ldso[i] = X_FLOATS + (q - PX) * 256;
        } else {
            src0[i] = a.zeros;
            stride[i] = 0;
            ldso[i] = -1;
        }
    }
    auto issue_piece = [&](int ch, int i) {
        float* dst = ldso[i] >= 0 ? lds + (ch % NST) * STAGE + ldso[i] : dummy;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0[i] + (long)ch * stride[i]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    int boff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int q = (wn * NTW + nt) * 16 + (lane & 15);
        const int r = q / WFIX, x = q - r * WFIX;
        boff[nt] = (r + 1) * RS + x + 4 + (lane >> 4) * CHS;
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(0, i);
    if (nchunks > 1) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(1, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    float av[2][MT], bv[2][NTW];
    auto fetch = [&](const float* Xs, const float* As, int st, int slot) {
        const int tap = st / CPW, c4 = wk + (st % CPW) * WK;
        const int toff = (tap / 3 - 1) * RS + (tap % 3 - 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) av[slot][m] = As[((c4 * 9 + tap) * MT + m) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bv[slot][nt] = Xs[boff[nt] + c4 * 4 * CHS + toff];
    };
    fetch(lds, lds + X_FLOATS, 0, 0);
    // chunk body; PAR = parity of the fragment slot its first k-step reads (alternates from chunk to chunk when
    // STEPS is odd -- slot indices must stay compile-time constants, so the two parities are two instantiations)
    auto chunk_body = [&](auto par_c, int ch) {
        constexpr int PAR = decltype(par_c)::value;
        const float* Xs = lds + (ch % NST) * STAGE;
        const float* As = Xs + X_FLOATS;
        const float* Xn = lds + ((ch + 1) % NST) * STAGE;
        const float* An = Xn + X_FLOATS;
        const bool has_next = ch + 1 < nchunks;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = (st + PAR) & 1;
            if (st == MID && has_next) {
                // chunk ch+1 was requested half a chunk ago or earlier; nothing younger is in flight yet
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // ... and every wave is past chunk ch-1: its stage is free
                asm volatile("" ::: "memory");
            }
            if (st + 1 < STEPS) fetch(Xs, As, st + 1, cur ^ 1);
            else if (has_next) fetch(Xn, An, 0, cur ^ 1);
            const bool dma = st >= MID && st - MID < PPW && ch + 2 < nchunks;
            if (dma) issue_piece(ch + 2, st - MID);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][m], bv[cur][nt], acc[m][nt], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
            if (st >= MID && st - MID < PPW) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
        }
    };
    if constexpr (STEPS % 2 == 0) {
        for (int ch = 0; ch < nchunks; ++ch) chunk_body(std::integral_constant<int, 0>{}, ch);
    } else {
        for (int ch = 0; ch < nchunks; ch += 2) {
            chunk_body(std::integral_constant<int, 0>{}, ch);
            if (ch + 1 < nchunks) chunk_body(std::integral_constant<int, 1>{}, ch + 1);
        }
    }
    __syncthreads();

    if (WK > 1) {
        float* part = lds;
        if (wk > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        part[(((wk - 1) * WN + wn) * (MT * NTW * 4) + (m * NTW + nt) * 4 + r) * 64 + lane] = acc[m][nt][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int k = 1; k < WK; ++k)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[m][nt][r] += part[(((k - 1) * WN + wn) * (MT * NTW * 4) + (m * NTW + nt) * 4 + r) * 64 + lane];
        }
    }
    double ld = 0.0;
    if (wk == 0) ld = tail_epilogue<MT, NTW>(a, acc, n, p0, HW, wn, mt0, paired, lane);
    if (a.acc && (a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV || a.mode == TAIL_SPLIT_FWD)) {
        const double tot = block_sum<256>(ld, red);
        if (tid == 0) fix_atomic_add(a.acc, n, a.N, tot);
    }
}

template <int MT, int NTW, int WN, int WK, int WFIX>
static int launch_tail_dma(const TailConvArgs& a, int paired, hipStream_t s, int msplit) {
    constexpr int TP = 16 * NTW * WN;
    constexpr int CHS = tail_chs(TP / WFIX, WFIX);
    constexpr int X_FLOATS = (16 * CHS + 255) / 256 * 256, A_FLOATS = 4 * 9 * MT * 64;
    size_t lds = ((size_t)3 * (X_FLOATS + A_FLOATS) + 256 + 8) * sizeof(float);
    const size_t red = (size_t)(WK - 1) * WN * MT * NTW * 4 * 64 * sizeof(float);
    if (red > lds) lds = red;
    const long total_px = (long)a.N * a.H * a.W;
    (void)hipFuncSetAttribute((const void*)k_conv_tail_dma<MT, NTW, WN, WK, WFIX>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_conv_tail_dma<MT, NTW, WN, WK, WFIX>), dim3((unsigned)(total_px / TP), msplit), dim3(256), lds, s,
                       a, paired);
    GH_LAUNCH_CHECK("k_conv_tail_dma");
    return GLOWHIP_OK;
}

}  // namespace glowhip
