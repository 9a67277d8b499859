// conv_mfma.hip -- the coupling network's convolutions on the CDNA4 matrix cores.
//
// Arithmetic: fp32-input MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32).  These are exact
// fp32 (bitwise an fmaf chain), which is what the <=1e-4 parity bar needs -- a single bf16 pass is
// measurably outside it.  Peak 157.3 TFLOP/s.
//
//   k_conv_wide  : implicit GEMM  Y[o][pix] = sum_k Wt[k][o] * im2col(X)[k][pix]  for the two convolutions
//                  with hidden (512) output channels: f.0 (3x3, K = 9*Cin) and f.2 (1x1, K = 512), ActNorm
//                  (+bias, *exp(3 logs)) and ReLU fused into the epilogue (network/module.py:252-259,314-317).
//                  Block tile BM x BN (out-channels x pixels), 4 waves in 2x2, each wave (BM/2)x(BN/2) as
//                  32x32 MFMA tiles; operands K-major in LDS so every fragment read is 32 consecutive floats
//                  (conflict-free ds_read_b32); next K-tile prefetched into registers under the MFMAs.
//   k_conv_tail  : 3x3 convolution to a few (<=48) output channels (f.4 / Split2d prior), 16x16x4 MFMA with
//                  out-channels on M and 16-pixel runs on N; input staged once per 32-channel chunk as a
//                  zero-padded halo tile in LDS, the 9 taps are LDS address offsets (no im2col expansion).
//                  The coupling / prior arithmetic and the per-sample log-det reduction run in the epilogue
//                  on the accumulator registers (network/model.py:105-113,131-139; module.py:526-536).
#include "conv_mfma.h"

namespace glowhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ================================================================================================
// k_conv_wide
// ================================================================================================
template <int KS, int BM, int BN, int BK>
__global__ void __launch_bounds__(256)
k_conv_wide(const float* __restrict__ X, long x_bs, const float* __restrict__ Wt, const float* __restrict__ bias,
            const float* __restrict__ scale, float* __restrict__ Y, int N, int Cin, int H, int W, int M, int K,
            int Kpad, int relu, float* __restrict__ Ypart) {
    // gridDim.y > 1: split-K -- workgroup (x, y) reduces k-tiles [y, y+1) * nkt / gridDim.y and writes its RAW partial sums to
    // Ypart[y] (k_splitk_finish adds them up in a fixed order and applies the epilogue).  For the deep levels of the large
    // configurations: 16 images x 4 x 4 pixels x 384 channels is 24 output tiles for a reduction of 4 608.
    constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
    constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 MFMA tiles per wave
    constexpr int A_F4 = BM / 4, B_F4 = BN / 4;  // float4 per tile row
    constexpr int A_RPP = 256 / A_F4, B_RPP = 256 / B_F4;  // rows per pass
    constexpr int A_PASSES = BK / A_RPP, B_PASSES = BK / B_RPP;
    static_assert(A_PASSES >= 1 && B_PASSES >= 1, "tile too wide for 256 threads");
    // double-buffered K-major operand tiles: ONE barrier per K-tile (the write of tile t+1 goes to the
    // buffer nobody reads during tile t)
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * BM + 2 * BK * BN];
    float (*As)[BK][BM] = reinterpret_cast<float (*)[BK][BM]>(smem);
    float (*Bs)[BK][BN] = reinterpret_cast<float (*)[BK][BN]>(smem + 2 * BK * BM);

    const int HW = H * W;
    const long total_px = (long)N * HW;
    const int tiles_m = M / BM;
    const int tiles_n = (int)((total_px + BN - 1) / BN);
    const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = logical % tiles_m, tile_n = logical / tiles_m;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    // ---- global -> register staging coordinates
    const int a_row = tid / A_F4, a_c4 = tid % A_F4;
    const float* a_src = Wt + (long)a_row * M + (long)tile_m * BM + a_c4 * 4;
    const int b_row = tid / B_F4, b_c4 = tid % B_F4;
    const long b_gp = (long)tile_n * BN + b_c4 * 4;   // first of this thread's 4 pixels (flattened n*HW+p)
    const bool b_ok = b_gp < total_px;                // HW % 4 == 0 => the 4 pixels are in or out together
    const long b_n = b_ok ? b_gp / HW : 0;
    const int b_p = b_ok ? (int)(b_gp - b_n * HW) : 0;
    const int b_y = b_p / W, b_x = b_p - b_y * W;
    const float* b_img = X + b_n * x_bs;

    f32x4 ra[A_PASSES], rb[B_PASSES];  // native vectors: HIP's float4 struct defeats SROA here (array lands in scratch/LDS)
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps)
            ra[ps] = *reinterpret_cast<const f32x4*>(a_src + (long)(kt * BK + ps * A_RPP) * M);
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            const int k = kt * BK + ps * B_RPP + b_row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (KS == 1) {
                if (b_ok) v = *reinterpret_cast<const f32x4*>(b_img + (long)k * HW + b_p);
            } else {
                const int ci = k / 9, tap = k - ci * 9;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int yy = b_y + dy;
                if (b_ok && k < K && yy >= 0 && yy < H) {
                    const float* row = b_img + (long)ci * HW + yy * W;
                    const int x0 = b_x + dx;
                    v[0] = (x0 >= 0) ? row[x0] : 0.f;
                    v[1] = row[x0 + 1];
                    v[2] = row[x0 + 2];
                    v[3] = (x0 + 3 < W) ? row[x0 + 3] : 0.f;
                }
            }
            rb[ps] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps)
            *reinterpret_cast<f32x4*>(&As[buf][ps * A_RPP + a_row][a_c4 * 4]) = ra[ps];
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps)
            *reinterpret_cast<f32x4*>(&Bs[buf][ps * B_RPP + b_row][b_c4 * 4]) = rb[ps];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt_all = Kpad / BK;
    const int kt0 = (int)((long)blockIdx.y * nkt_all / gridDim.y), nkt = (int)((long)(blockIdx.y + 1) * nkt_all / gridDim.y);
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    const int kl = lane >> 5, ml = lane & 31;
    for (int kt = kt0; kt < nkt; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);  // HBM/L2 latency hidden under this tile's MFMAs
        // fragment reads are software-pipelined one k-step ahead of the MFMAs that consume them
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[buf][kl][wr * WM + i * 32 + ml];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bs[buf][kl][wc * WN + j * 32 + ml];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = As[buf][(kk + 1) * 2 + kl][wr * WM + i * 32 + ml];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = Bs[buf][(kk + 1) * 2 + kl][wc * WN + j * 32 + ml];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            // the refill of the OTHER LDS buffer is spread over the last NW k-steps, one ds_write_b128 per step
            // between MFMA groups: as one burst at the end of the tile it held the LDS for ~400 cycles and the
            // next tile's first fragment reads (hence the matrix pipe) queued behind it (-14 % measured)
            constexpr int NW = A_PASSES + B_PASSES;
            const int wsel = kk - (BK / 2 - NW);
            if (wsel >= 0 && kt + 1 < nkt) {
                if (wsel < A_PASSES)
                    *reinterpret_cast<f32x4*>(&As[buf ^ 1][(wsel % A_PASSES) * A_RPP + a_row][a_c4 * 4]) = ra[wsel % A_PASSES];
                else
                    *reinterpret_cast<f32x4*>(&Bs[buf ^ 1][((wsel - A_PASSES) % B_PASSES) * B_RPP + b_row][b_c4 * 4]) =
                        rb[(wsel - A_PASSES) % B_PASSES];
            }
            // pin the issue order: next step's LDS reads go out BEFORE this step's MFMAs (hipcc otherwise sinks
            // them behind the MFMAs and waits lgkmcnt(0) with the matrix pipe idle)
            if (kk + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            if (wsel >= 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: ActNorm (+bias, *scale) + ReLU on the accumulators, C[row=o][col=pixel], then a transpose
    // through LDS (the operand buffers are dead) so every lane stores 16 contiguous bytes: 4x fewer store
    // instructions than the natural one-dword-per-lane layout, which was store-ISSUE-bound.
    float* stage = smem + wid * (WM * WN);
    static_assert(2 * BK * (BM + BN) >= 4 * WM * WN, "operand buffers too small to stage the output tile");
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            const int o = tile_m * BM + wr * WM + row;
            const bool part = gridDim.y > 1;
            const float bo = (bias && !part) ? bias[o] : 0.f, so = (scale && !part) ? scale[o] : 1.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float v = part ? acc[i][j][r] : (acc[i][j][r] + bo) * so;
                stage[row * WN + j * 32 + ml] = (relu && !part) ? relu_(v) : v;
            }
        }
    }
    // each wave reads back only what it wrote: no workgroup barrier needed, the LDS ops of one wave are ordered
    constexpr int F4_PER_ROW = WN / 4, ROWS_PER_IT = 64 / F4_PER_ROW;
    const int rrow = lane / F4_PER_ROW, rc4 = lane % F4_PER_ROW;
    const long gp = (long)tile_n * BN + wc * WN + rc4 * 4;
    if (gp < total_px) {
        const long n = gp / HW;
        const int p = (int)(gp - n * HW);
        float* yn = (gridDim.y > 1 ? Ypart + (long)blockIdx.y * N * M * HW : Y) + n * (long)M * HW + p;
#pragma unroll
        for (int it = 0; it < WM / ROWS_PER_IT; ++it) {
            const int row = it * ROWS_PER_IT + rrow;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * WN + rc4 * 4);
            const int o = tile_m * BM + wr * WM + row;
            *reinterpret_cast<f32x4*>(yn + (long)o * HW) = v;
        }
    }
}

// ================================================================================================
// k_gemm_glds: the 1x1 convolution (f.2) with operand tiles streamed HBM/L2 -> LDS by the DMA path
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass) into a 4-stage ring of 16-deep K-tiles.
// Tile t+3 is requested while tile t is multiplied; a stage is consumed after a COUNTED s_waitcnt vmcnt(8)
// (two younger tiles stay in flight across the barrier) + one raw s_barrier per tile.  All LDS is one array
// (a second __shared__ object makes hipcc drain vmcnt(0) before every ds_read of such a pipeline).
// LDS image of a tile = [16][128] floats K-major, written linearly (a wave-instruction fills 1 KiB = two rows),
// which is exactly the conflict-free layout the MFMA fragment reads want: no swizzle needed.
// ================================================================================================
constexpr int GL_BK = 16, GL_ST = 4, GL_BM = 128, GL_BN = 128;

__global__ void __launch_bounds__(256)
k_gemm_glds(const float* __restrict__ X, long x_bs, const float* __restrict__ Wt, const float* __restrict__ bias,
            const float* __restrict__ scale, float* __restrict__ Y, int N, int K, int HW, int M, int relu) {
    constexpr int BM = GL_BM, BN = GL_BN, BK = GL_BK;
    constexpr int WM = 64, WN = 64, TM = 2, TN = 2;
    constexpr int STAGE = BK * BM + BK * BN;   // floats per stage (16 KiB)
    extern __shared__ __attribute__((aligned(16))) float smem[];   // GL_ST * STAGE floats = 64 KiB

    const long total_px = (long)N * HW;
    const int tiles_m = M / BM;
    const int tiles_n = (int)(total_px / BN);
    const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = logical % tiles_m, tile_n = logical / tiles_m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int kl = lane >> 5, ml = lane & 31;

    // per-lane DMA sources: wave w fills rows 4w..4w+3 of the A tile and of the B tile (two 1-KiB pieces each)
    const float* a_src = Wt + (long)(4 * wid + kl) * M + (long)tile_m * BM + ml * 4;
    const long gp = (long)tile_n * BN + ml * 4;
    const long n = gp / HW;
    const int p = (int)(gp - n * HW);
    const float* b_src = X + n * x_bs + p + (long)(4 * wid + kl) * HW;
    const int nkt = K / BK;

    // one DMA piece (1 KiB) of tile kt: piece 0,1 = this wave's two A row-pairs, 2,3 = its two B row-pairs
    auto issue_piece = [&](int kt, int piece) {
        float* st = smem + (kt % GL_ST) * STAGE;
        const int jj = piece & 1;
        if (piece < 2)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(a_src + (long)(kt * BK + 2 * jj) * M),
                (__attribute__((address_space(3))) void*)(st + (2 * wid + jj) * 256), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(b_src + (long)(kt * BK + 2 * jj) * HW),
                (__attribute__((address_space(3))) void*)(st + BK * BM + (2 * wid + jj) * 256), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nkt) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) issue_piece(t, pc);
        }
    // tile 0 has landed once at most tiles 1 and 2 (4 pieces each per wave) are outstanding
    if (nkt >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nkt == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // One continuous stream of k-steps across tiles.  Mid-tile (k-step 4 of 8) the NEXT tile is made visible
    // (counted vmcnt + raw barrier: the matrix pipe still has this tile's remaining MFMAs queued), so its first
    // fragments are prefetched before this tile's last MFMAs; that barrier also proves every wave is past tile
    // kt-1, whose stage is then refilled with tile kt+3, one DMA piece per k-step (an LDS-DMA issue costs ~60-180
    // cycles: back to back they would idle the pipe, one at a time they hide behind an executing MFMA).
    float a[2][TM], b[2][TN];
    {
        const float* As = smem;
        const float* Bs = As + BK * BM;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[kl * BM + wr * WM + i * 32 + ml];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bs[kl * BN + wc * WN + j * 32 + ml];
    }
    for (int kt = 0; kt < nkt; ++kt) {
        const float* As = smem + (kt % GL_ST) * STAGE;
        const float* Bs = As + BK * BM;
        const float* An = smem + ((kt + 1) % GL_ST) * STAGE;
        const float* Bn = An + BK * BM;
        const bool has_next = kt + 1 < nkt;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk == 4 && has_next) {
                if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = As[((kk + 1) * 2 + kl) * BM + wr * WM + i * 32 + ml];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = Bs[((kk + 1) * 2 + kl) * BN + wc * WN + j * 32 + ml];
            } else if (has_next) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = An[kl * BM + wr * WM + i * 32 + ml];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = Bn[kl * BN + wc * WN + j * 32 + ml];
            }
            const bool dma = kk >= 4 && kt + 3 < nkt;
            if (dma) issue_piece(kt + 3, kk - 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            if (kk >= 4) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
    }
    __syncthreads();   // all waves done with the operand ring before it is reused as the output staging area

    // ---- epilogue (as k_conv_wide): ActNorm + ReLU (all optional), transpose through LDS, 16-byte stores
    float* stage = smem + wid * (WM * WN);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            const int o = tile_m * BM + wr * WM + row;
            const bool part = gridDim.y > 1;
            const float bo = (bias && !part) ? bias[o] : 0.f, so = (scale && !part) ? scale[o] : 1.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float v = part ? acc[i][j][r] : (acc[i][j][r] + bo) * so;
                stage[row * WN + j * 32 + ml] = (relu && !part) ? relu_(v) : v;
            }
        }
    }
    constexpr int F4_PER_ROW = WN / 4, ROWS_PER_IT = 64 / F4_PER_ROW;
    const int rrow = lane / F4_PER_ROW, rc4 = lane % F4_PER_ROW;
    const long gpo = (long)tile_n * BN + wc * WN + rc4 * 4;
    const long no = gpo / HW;
    const int po = (int)(gpo - no * HW);
    float* yn = Y + no * (long)M * HW + po;
#pragma unroll
    for (int it = 0; it < WM / ROWS_PER_IT; ++it) {
        const int row = it * ROWS_PER_IT + rrow;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * WN + rc4 * 4);
        const int o = tile_m * BM + wr * WM + row;
        *reinterpret_cast<f32x4*>(yn + (long)o * HW) = v;
    }
}

static bool g_disable_glds = false;   // testing hook
void conv_mfma_wide_disable_glds(int off) { g_disable_glds = off != 0; }

static bool gemm_glds_ok(int Cin, int HW, int Cout, long total_px) {
    return !g_disable_glds && Cout % GL_BM == 0 && Cin % GL_BK == 0 && Cin >= 3 * GL_BK && HW % 4 == 0 &&
           total_px % GL_BN == 0;
}

static int pick_wide_tile(int Cout, long total_px) {
    // 128x128 when the grid still fills the chip (>= 2 tiles per CU), else 64x64
    if (Cout % 128 == 0 && (Cout / 128) * ((total_px + 127) / 128) >= 512) return 128;
    return 64;
}

bool conv_mfma_wide_supported(int Cin, int H, int W, int Cout, int ksize) {
    if (Cout % 64 != 0) return false;
    if ((H * W) % 4 != 0) return false;
    if (ksize == 1) return Cin % 32 == 0;
    if (ksize == 3) return W % 4 == 0 && Cin >= 1;
    return false;
}

size_t conv_mfma_wide_packed_bytes(int Cin, int Cout, int ksize) {
    return (size_t)wide_kpad(Cin, ksize) * Cout * sizeof(float);
}

// w (Cout, Cin, k, k) -> wt [Kpad][Cout], k = ci*k*k + tap, zero rows beyond K
__global__ void __launch_bounds__(256) k_pack_wide(const float* __restrict__ w, int K, int Kpad, int Cout,
                                                   float* __restrict__ wt) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Kpad * Cout) return;
    const int k = (int)(i / Cout), o = (int)(i - (long)k * Cout);
    wt[i] = (k < K) ? w[(long)o * K + k] : 0.f;
}

int conv_mfma_wide_pack(const float* w, int Cin, int Cout, int ksize, float* wt, hipStream_t s) {
    const int K = Cin * ksize * ksize, Kpad = wide_kpad(Cin, ksize);
    hipLaunchKernelGGL(k_pack_wide, dim3(cdiv((long)Kpad * Cout, 256)), dim3(256), 0, s, w, K, Kpad, Cout, wt);
    GH_LAUNCH_CHECK("k_pack_wide");
    return GLOWHIP_OK;
}

// y[n][o][p] = act((sum_s part[s][n][o][p] + bias[o]) * scale[o]), s ascending (deterministic)
__global__ void __launch_bounds__(256) k_splitk_finish(const float* __restrict__ part, int S, long per, const float* __restrict__ bias,
                                                       const float* __restrict__ scale, int relu, float* __restrict__ Y, int M, int HW) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= per) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + i4);
    for (int sidx = 1; sidx < S; ++sidx) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(part + (long)sidx * per + i4);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    const int o = (int)((i4 / HW) % M);                    // HW % 4 == 0: the four values share the channel
    const float bo = bias ? bias[o] : 0.f, so = scale ? scale[o] : 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = (v[q] + bo) * so; if (relu) v[q] = relu_(v[q]); }
    *reinterpret_cast<f32x4*>(Y + i4) = v;
}

template <int KS, int BMN>
static int launch_wide_cfg(const float* x, long x_bs, const float* wt, const float* bias, const float* scale, float* y,
                           int N, int Cin, int H, int W, int Cout, hipStream_t s, int relu, float* scratch, size_t scratch_floats) {
    const long total_px = (long)N * H * W;
    const int tiles = (Cout / BMN) * (int)((total_px + BMN - 1) / BMN);
    const int K = Cin * KS * KS;
    const int nkt = wide_kpad(Cin, KS) / 32;
    // split-K when the output tiles alone leave most of the chip idle and the reduction is long enough to share
    int S = 1;
    const long per = (long)N * Cout * H * W;
    if (scratch && tiles < 96 && nkt >= 16) {
        S = std::min(std::min(16, 320 / tiles), nkt / 4);
        while (S > 1 && (size_t)S * per > scratch_floats) --S;
    }
    hipLaunchKernelGGL((k_conv_wide<KS, BMN, BMN, 32>), dim3(tiles, S), dim3(256), 0, s, x, x_bs, wt, bias, scale, y, N,
                       Cin, H, W, Cout, K, wide_kpad(Cin, KS), relu, scratch);
    GH_LAUNCH_CHECK("k_conv_wide");
    if (S > 1) {
        hipLaunchKernelGGL(k_splitk_finish, dim3(cdiv(per / 4, 256)), dim3(256), 0, s, scratch, S, per, bias, scale, relu, y, Cout, H * W);
        GH_LAUNCH_CHECK("k_splitk_finish");
    }
    return GLOWHIP_OK;
}

int launch_conv_mfma_wide(const float* x, long x_bs, const float* wt, const float* post_bias, const float* post_scale,
                          float* y, int N, int Cin, int H, int W, int Cout, int ksize, hipStream_t s, int relu,
                          float* splitk_scratch, size_t splitk_floats) {
    GH_REQUIRE(conv_mfma_wide_supported(Cin, H, W, Cout, ksize), "conv_mfma_wide: unsupported shape");
    if (N == 0) return GLOWHIP_OK;
    const int t = pick_wide_tile(Cout, (long)N * H * W);
    if (ksize == 1 && t == 128 && gemm_glds_ok(Cin, H * W, Cout, (long)N * H * W)) {
        const long total_px = (long)N * H * W;
#ifndef GLOWHIP_EXP_LDSPAD
#define GLOWHIP_EXP_LDSPAD 0
#endif
        const size_t lds = (size_t)GL_ST * (GL_BK * GL_BM + GL_BK * GL_BN) * sizeof(float) + GLOWHIP_EXP_LDSPAD;
        (void)hipFuncSetAttribute((const void*)k_gemm_glds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_gemm_glds, dim3((unsigned)((Cout / GL_BM) * (total_px / GL_BN))), dim3(256), lds, s, x, x_bs, wt,
                           post_bias, post_scale, y, N, Cin, H * W, Cout, relu);
        GH_LAUNCH_CHECK("k_gemm_glds");
        return GLOWHIP_OK;
    }
    if (ksize == 1) {
        if (t == 128) return launch_wide_cfg<1, 128>(x, x_bs, wt, post_bias, post_scale, y, N, Cin, H, W, Cout, s, relu, splitk_scratch, splitk_floats);
        return launch_wide_cfg<1, 64>(x, x_bs, wt, post_bias, post_scale, y, N, Cin, H, W, Cout, s, relu, splitk_scratch, splitk_floats);
    }
    if (t == 128) return launch_wide_cfg<3, 128>(x, x_bs, wt, post_bias, post_scale, y, N, Cin, H, W, Cout, s, relu, splitk_scratch, splitk_floats);
    return launch_wide_cfg<3, 64>(x, x_bs, wt, post_bias, post_scale, y, N, Cin, H, W, Cout, s, relu, splitk_scratch, splitk_floats);
}

// ================================================================================================
// k_conv_tail  (brought up after the wide kernels; see conv_mfma_tail.hip)
// ================================================================================================

}  // namespace glowhip
