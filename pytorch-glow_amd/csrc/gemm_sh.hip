// gemm_sh.hip -- f.2 (1x1 convolution hidden -> hidden + ActNorm + ReLU, network/module.py:300-319) on split-half operands
// (sh.h): three v_mfma_f32_32x32x16_f16 per k-step (hi*hi into the main accumulator, hi*lo + lo*hi into the cross
// accumulator) instead of eight v_mfma_f32_32x32x2_f32 -- 16x the matrix rate at 3x the instructions.
//
// Workgroup = 128 out-channels x 128 pixels, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA tiles x 2 accumulators
// (128 accumulator registers).  Operands stream HBM/L2 -> LDS through the DMA path (global_load_lds_dwordx4) into a 4-stage ring
// of 16-deep k-tiles (64 KiB: two workgroups per CU, so one's epilogue overlaps the other's loop); the LDS image of a stage is
// [A|B][plane][chunk 2][128 rows][8 halfs] -- written linearly by the DMA (one wave-instruction = 64 rows x 16 B = 1 KiB) and
// read as conflict-free 16-byte fragments (16 consecutive lanes = 256 contiguous bytes = every bank once).  A stage is
// consumed behind a counted s_waitcnt vmcnt + ONE raw s_barrier per 12 MFMAs; that barrier also proves the previous stage is
// drained, so its slot is refilled right behind it.  In the product path this kernel serves the levels the fused f.0+f.2
// kernel (f02_sh.hip) does not take, and f.2 of the training step (forward and input gradient).
#include "sh.h"
#include "conv_mfma.h"

GH_STAMPS_DEFINE(gemm)

namespace glowhip {

constexpr int SH_BM = 128, SH_BN = 128, SH_BK = 16, SH_ST = 4;
constexpr int SH_A_HALFS = 2 * (SH_BK / 8) * 128 * 8;              // [plane][chunk 2][128][8] = 4096 halfs
constexpr int SH_STAGE_HALFS = 2 * SH_A_HALFS;                    // A + B: 16 KiB -> 64 KiB ring, two workgroups per CU

// Epilogue shared by the GEMM variants: main + cross / 2^11 + folded ActNorm bias, ReLU; C[row = channel][col = pixel]:
// lane (kl, ml) holds pixel ml and channels 8*(r>>2) + 4*kl + (r&3) of each 32-row tile.
template <bool OUT_SH>
__device__ __forceinline__ void gemm_sh_epilogue(const f32x16_t (&accm)[2][2], const f32x16_t (&accx)[2][2],
                                                 const float* __restrict__ bias, float* __restrict__ Yf,
                                                 _Float16* __restrict__ Ysh, long P, int M, int HW, int relu, float out_scale, int tile_m,
                                                 int tile_n, int wr, int wc, int kl, int ml) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long px = (long)tile_n * SH_BN + wc * 64 + j * 32 + ml;
        const bool ok = px < P;
        const long n = px / HW;
        const int p = (int)(px - n * HW);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o0 = tile_m * SH_BM + wr * 64 + i * 32 + 8 * g + 4 * kl;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = (accm[i][j][4 * g + q] + accx[i][j][4 * g + q] * SH_LO_INV) * out_scale;   // power of two: exact
                    v[q] = relu ? relu_(t) : t;
                }
                if (OUT_SH) {
                    h4 hi, lo;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        _Float16 a, b;
                        sh_split(v[q], a, b);
                        hi[q] = a; lo[q] = b;
                    }
                    if (ok) {
                        _Float16* dst = Ysh + sh_off(M >> 3, 0, o0 >> 3, px) + (o0 & 7);
                        *reinterpret_cast<h4*>(dst) = hi;
                        *reinterpret_cast<h4*>(dst + (long)(M >> 3) * SH_CHUNK_STEP) = lo;
                    }
                } else if (ok) {
                    float* dst = Yf + (n * M + o0) * (long)HW + p;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[(long)q * HW] = v[q];
                }
            }
        }
    }
}

template <bool OUT_SH>
__global__ void __launch_bounds__(256, 2)
k_gemm_sh(const _Float16* __restrict__ X, long P, const _Float16* __restrict__ Wsh, const float* __restrict__ bias,
          float* __restrict__ Yf, _Float16* __restrict__ Ysh, int K, int M, int HW, int relu, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];   // SH_ST stages
    GH_STAMP(16);
    const int tiles_m = M / SH_BM;
    const int tiles_n = (int)((P + SH_BN - 1) / SH_BN);
    const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = logical % tiles_m, tile_n = logical / tiles_m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int kl = lane >> 5, ml = lane & 31;
    const long w_plane = (long)K * M;
    const int NCK = K >> 3;                         // 8-channel chunks of the input tensor

    // DMA sources of this lane: wave w streams (plane = w>>1, chunk = w&1) of every stage: 2 A pieces (64-row halves) + 2 B pieces
    const int dpl = wid >> 1, dch = wid & 1;
    const _Float16* a_src = Wsh + dpl * w_plane + ((long)dch * M + tile_m * SH_BM + lane) * 8;
    long px0 = (long)tile_n * SH_BN + lane, px1 = px0 + 64;
    px0 = px0 < P ? px0 : P - 1;   // ragged last tile: clamp the fetch, the stores are predicated
    px1 = px1 < P ? px1 : P - 1;
    const _Float16* b_src0 = X + sh_off(NCK, dpl, dch, px0);
    const _Float16* b_src1 = X + sh_off(NCK, dpl, dch, px1);
    const int nkt = K / SH_BK;

    auto issue_piece = [&](int kt, int piece) {   // piece 0,1: A halves; 2,3: B halves
        _Float16* st = smem_h + (kt % SH_ST) * SH_STAGE_HALFS + ((dpl * 2 + dch) * 128 + (piece & 1) * 64) * 8;
        if (piece < 2) {
            const _Float16* src = a_src + ((long)kt * 2 * M + (piece & 1) * 64) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)st, 16, 0, 0);
        } else {
            const _Float16* src = ((piece & 1) ? b_src1 : b_src0) + (long)kt * 2 * SH_CHUNK_STEP;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(st + SH_A_HALFS), 16, 0, 0);
        }
    };

    // the folded ActNorm bias is the accumulators' initial value: its load latency hides behind the first DMA wait and the
    // epilogue holds no per-row parameter
    f32x16_t accm[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bias + tile_m * SH_BM + wr * 64 + i * 32 + 8 * g + 4 * kl);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) { accm[i][j][4 * g + q] = b4[q]; accx[i][j][4 * g + q] = 0.f; }
        }

#pragma unroll
    for (int t = 0; t < SH_ST - 1; ++t)
        if (t < nkt) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) issue_piece(t, pc);
        }

    GH_STAMP(17);
    const int a_off = (kl * 128 + wr * 64 + ml) * 8, b_off = SH_A_HALFS + (kl * 128 + wc * 64 + ml) * 8;
    for (int kt = 0; kt < nkt; ++kt) {
        if (SH_ST == 4 && kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt < 8) GH_STAMP(24 + kt);
        const _Float16* st = smem_h + (kt % SH_ST) * SH_STAGE_HALFS;
        h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const h8*>(st + a_off + i * 256);
            al[i] = *reinterpret_cast<const h8*>(st + a_off + i * 256 + 2 * 128 * 8);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[j] = *reinterpret_cast<const h8*>(st + b_off + j * 256);
            bl[j] = *reinterpret_cast<const h8*>(st + b_off + j * 256 + 2 * 128 * 8);
        }
        if (kt + SH_ST - 1 < nkt) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) issue_piece(kt + SH_ST - 1, pc);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accm[i][j], 0, 0, 0);
                accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
            }
    }

    GH_STAMP(18);
    gemm_sh_epilogue<OUT_SH>(accm, accx, bias, Yf, Ysh, P, M, HW, relu, out_scale, tile_m, tile_n, wr, wc, kl, ml);
    GH_STAMP(19);
}

bool gemm_sh_supported(int K, int M, int H, int W) {
    return M % SH_BM == 0 && K % SH_BK == 0 && K >= SH_BK && (H * W) % 64 == 0;   // SH tensors are tiled by 64 pixels
}

size_t gemm_sh_packed_bytes(int K, int M) {
    return align_up((size_t)2 * K * M * sizeof(_Float16), 16) + (size_t)M * sizeof(float);
}

int launch_gemm_sh(const _Float16* x_sh, const void* wsh, float* y_f32, _Float16* y_sh, int N, int K, int HW, int M, int relu,
                   hipStream_t s, float out_scale) {
    GH_REQUIRE(gemm_sh_supported(K, M, HW, 1), "gemm_sh: unsupported shape K=%d M=%d HW=%d", K, M, HW);
    GH_REQUIRE((y_f32 != nullptr) != (y_sh != nullptr), "gemm_sh: exactly one output");
    if (N == 0) return GLOWHIP_OK;
    const long P = (long)N * HW;
    const _Float16* w = (const _Float16*)wsh;
    const float* bias = (const float*)((const char*)wsh + align_up((size_t)2 * K * M * sizeof(_Float16), 16));
    const size_t lds = (size_t)SH_ST * SH_STAGE_HALFS * sizeof(_Float16);
    const unsigned grid = (unsigned)((M / SH_BM) * ((P + SH_BN - 1) / SH_BN));
    if (y_sh) {
        (void)hipFuncSetAttribute((const void*)k_gemm_sh<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_gemm_sh<true>, dim3(grid), dim3(256), lds, s, x_sh, P, w, bias, nullptr, y_sh, K, M, HW, relu, out_scale);
    } else {
        (void)hipFuncSetAttribute((const void*)k_gemm_sh<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_gemm_sh<false>, dim3(grid), dim3(256), lds, s, x_sh, P, w, bias, y_f32, nullptr, K, M, HW, relu, out_scale);
    }
    GH_LAUNCH_CHECK("k_gemm_sh");
    return GLOWHIP_OK;
}

}  // namespace glowhip
