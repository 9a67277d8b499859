// cnet1w_sh.hip -- the coupling network  h = f(z1) = f.4(relu(f.2(relu(f.0(z1)))))  (network/module.py:300-319) of one FlowStep with
// ONE wave per SIMD: four waves per workgroup, 512 registers each, and NO activation ever leaves the register file.
//
// k_cnet (cnet_sh.hip) gives a wave a block of h2 ROWS for all 128 pixels of the tile; h1 and h2 then have to travel between waves
// through LDS (split + store epilogues, barriers, a hand-over of h2 before f.4), and those phases issue no MFMA: 35 k of a level-1
// workgroup's 116 k cycles.  Here a wave owns 32 PIXELS and every row of them:
//   * the accumulator layout of v_mfma_f32_32x32x16_f16 (lane = pixel column, 16 registers = rows 8 g + 4 (lane / 32) + t) IS the
//     B-operand layout of the next layer's MFMA (lane = pixel column, 8 consecutive k) up to a permutation of k inside a 32-row
//     block -- and a contraction does not care in which order its k are visited as long as A and B agree.  The f.2 / f.4 weight
//     images are therefore k-PERMUTED at pack time (sh.h sh2_kperm_src), and relu + (hi, lo) split of a 32-row block of h1 / h2 turns 16 accumulator registers into the two k-steps of B fragments
//     the next layer multiplies -- in registers.  h1 exists 32 channels at a time (16 registers), h2 as the 256 accumulator
//     registers of the wave (AGPRs), T = f.4's taps-as-rows output as 16 * NRT4 more.
//   * all four waves need every weight, so the weights stream L2 -> LDS ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4:
//     no staging registers) into a three-slot ring of 32 KiB k-steps of the f.2 image, later the f.4 image; f.0's rows of the
//     next chunk go to a small double buffer.  Weight bytes per MFMA are those of k_cnet's 128-pixel tile.
//   * f.0 of chunk c + 1 (15 - 27 MFMAs) and its epilogue (64 VALU instructions) ride between the 96 MFMAs f.2 spends on chunk c:
//     one wave per SIMD issues up to five other instructions in the shadow of each MFMA (MI355X_MICROARCH.md), so the matrix pipe
//     stays busy through what used to be separate phases.
// Output: the partial sums `hpart` / `hup` / `hdn` of k_cnet at MS = 1 with 128-pixel tiles -- k_cfinish does not know the difference.
// Instances (launch_cnet1w): product forward / inverse, the training step's taping forward (MODE 1) and its input-gradient launch
// (MODE 2: the transposed network) for the C = 12 levels; no chained prologue, one group of f.4 output channels.
// MS = 2: the h2 rows of a 128-pixel tile split over TWO workgroups (grid.y), each computing all of h1 again (f.0 is the small
// layer) and half of f.2 / a K-half of f.4 -- for the levels whose 128-pixel tiles alone leave half the CUs idle (C = 24 at 16 x 16
// pixels and batch 64: 128 tiles).  Weight bytes per MFMA stay those of a 128-pixel tile (a 64-pixel tile needs twice that, and at
// one workgroup per CU the launch is then bound by the L2 -> LDS stream); the partial sums are k_cnet's MS = 2 layout.
// (Measured 3 % slower than k_cnet's 64-pixel tiles where it applies -- each half repeats f.0, 29 % of its MFMAs -- and therefore
// selected only behind the debug switch 0x20000: cnet_sh.hip cnet_select.)
#include "sh.h"
#include <algorithm>
#include <type_traits>

#include "conv_mfma.h"
#include "cnet_geo.h"
#include "cnet_fin.h"

GH_STAMPS_DEFINE(cnet1w)
GH_WGTIMES_DEFINE(cnet1w)

namespace glowhip {

__device__ __forceinline__ void c1_dma16(const void* gsrc, void* ldst) {      // 64 lanes x 16 bytes -> 1 KiB at ldst (wave-uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void c1_dma16o(const void* gsrc, void* ldst) {   // the immediate offset OFF is added to BOTH addresses
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldst, 16, OFF, 0);
}
template <int OFF>
__device__ __forceinline__ void c1_bdma16o(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, void* ldst) {      // the same through a buffer descriptor
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)ldst, 16, voff, soff, OFF, 0);
}
// a partial sum of f.4 to the scratch buffer: plain store, or -- fused finishing -- an agent-scope relaxed atomic store (sc1: written
// through the XCD's L2, so that a workgroup on another XCD reads it with the matching load)
template <bool COH>
__device__ __forceinline__ void c1_publish(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
#define C1_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0)
#define C1_FENCE() __builtin_amdgcn_sched_barrier(0)

// f.0's accumulator block lives in VGPRs: the 256 AGPRs hold h2, and hipcc selects the AGPR form for every MFMA builtin of a kernel
// that may use AGPRs (a 17th accumulator block made it shuttle blocks between the two files: 224 v_accvgpr_* per chunk).  So f.0's
// MFMAs are inline asm with "v" operands.  hipcc pads nothing around an asm statement (cdna_hip_programming.md 5.7): the `s_nop 1`
// covers a compiler VALU copy into an operand right before it; a chain on one accumulator needs no states; and every reader of
// the result goes through c1_settle first (an 8-pass MFMA's result may be read 12 states after it issued).
__device__ __forceinline__ void c1_mfma_v0(f32x16_t& acc, const h8& a, const h8& b) {
#ifndef C1_ASM_MFMA_PAD
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
#else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ void c1_mfma_v(f32x16_t& acc, const h8& a, const h8& b) {
#ifndef C1_ASM_MFMA_PAD
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ void c1_settle(f32x16_t& acc) { asm volatile("s_nop 12" : "+v"(acc)); }

// 16 accumulator values of a 32-row block (already scaled / biased / rectified, NEGATED activations: sh.h) -> the B fragments of
// the two k-steps the block spans: bh[s], bl[s] = (hi, lo) of registers 8 s .. 8 s + 7
__device__ __forceinline__ void c1_frag(const f32x4_t& v, h8& bh, h8& bl, int half) {
    h4 hi, lo;
    sh2_split4<true>(v, hi, lo);
#pragma unroll
    for (int t = 0; t < 4; ++t) { bh[4 * half + t] = hi[t]; bl[4 * half + t] = lo[t]; }
}

// f.0's MFMA j (k-step j / 3, product j % 3) of the chunk riding along: its slot among the chunk's 24 NQ f.2 slots.  Four quads per
// k-step (MS = 1): k-step st in quad st, behind slots 3, 7, 11.  Two (MS = 2): one per slot from slot 0 on -- 27 of them (G0 = 18)
// in the 24 slots of the chunk's first k-step, every ninth doubled up -- so that the epilogue has the second k-step's 24 slots.
// (MS = 1 with more than five k-steps of f.0 -- the backward launch's first layer has C = 12 inputs, nine k-steps: every other slot.)
__host__ __device__ constexpr int c1_f0slot(int nq, int nst0, int j) {
    return nq == 4 ? (nst0 <= 5 ? 12 * (j / 3) + 3 + 4 * (j % 3) : 2 * j + 1) : j - (j + 1) / 9;
}

// TAPE (the training forward, plan_train.hip): h1 and h2 also go to memory as fp16 [pixel / 32][row][pixel % 32] and their signs as
// 16-bit words -- the formats k_cnet MODE 1 writes and the backward k_cnet / the weight-gradient GEMMs read (cnet_sh.hip).  A lane
// holds two consecutive rows of ONE pixel per packed register; a quad-permute with the neighbouring lane turns that into one row of
// TWO pixels (even lanes the even row, odd lanes the odd one): one 4-byte store per pair of values instead of two 2-byte ones.
// The stores ride in the epilogue pipeline (five more stages); they count in vmcnt like the stream's pieces, in order, so the
// counted waits of the stream allow for the stores issued behind the piece they wait for.
// BWD (MODE 2; the input-gradient chain of the same network, cnet_sh.hip MODE 2: x = d L / d(f.4 output), the images are those of the
// transposed weights): the "activation" of the first two layers is g_u = g_h * (h > 0) -- the ReLU masks READ from the tape's sign
// words, one 16-bit word per lane and block, requested a block ahead -- and g_u2 / g_u0 go to memory as fp32 [pixel / 32][row][pixel % 32]
// for the weight-gradient GEMMs: a lane's value of a row is 4 bytes of that row's 128-byte line, so a wave's store of one register
// writes two full lines (no exchange between lanes), 16 stores per block.
// FIN (product instance, MS = 1): the launch finishes the step itself (sh.h CnetArgs::fin_cnt; the tail of this kernel) -- no
// finishing kernel, one kernel boundary per FlowStep instead of two.
template <int HID, int G0, int NRT4, int MODE = 0, int MS = 1, bool FIN = false>
__global__ void __launch_bounds__(256) k_cnet1w(CnetArgs a, CnetGeo g) {
    static_assert(!FIN || (MODE == 0 && MS == 1), "fused finishing: the product instance without row split");
    constexpr bool TAPE = MODE == 1, BWD = MODE == 2;
    constexpr int NT = 256, LPXT = 7;
    constexpr int NCH = HID / 32;             // 32-channel chunks of the hidden width = row tiles of h1
    constexpr int NKS = HID / 16;             // k-steps of f.2
    constexpr int MR = HID / MS;              // h2 rows of this workgroup
    constexpr int NRT2 = MR / 32;             // ... as row tiles = accumulator blocks of the wave
    constexpr int NQ = NRT2 / 4;              // quads of row tiles = quads of 12 slots per k-step of f.2
    constexpr int NKS4 = MR / 16;             // k-steps of f.4 over this workgroup's h2 rows
    constexpr int SLOT = MR * 64;             // ring slot: one k-step of the workgroup's rows of the f.2 image, both planes (32 / 16 KiB)
    constexpr int PPF = 2 * NQ;               // 1-KiB pieces per wave of such a fill
    constexpr int NST0 = G0 / 2;              // k-steps of f.0
    constexpr int NP0 = (G0 + 3) / 4;         // DMA pieces per wave of a chunk's f.0 rows (G0 KiB)
    constexpr int MP4 = NRT4 * 32;            // rows of the taps-as-rows f.4 image
    constexpr int K4 = SLOT / (MP4 * 64) >= 8 ? 8 : (SLOT / (MP4 * 64) >= 4 ? 4 : (SLOT / (MP4 * 64) >= 2 ? 2 : 1));      // k-steps of the f.4 image per ring slot
    constexpr int NF4 = NKS4 / K4;            // fills of the f.4 image
    constexpr int PP4 = K4 * NRT4;            // 1-KiB pieces per plane of such a fill
    constexpr int PPW4 = (2 * PP4 + 3) / 4;   // ... per wave (both planes over four waves; a surplus piece repeats the last)
    constexpr bool RUN4 = PP4 % 2 == 0;       // a wave's pieces of an f.4 fill are one run of one plane (else: placed piece by piece)
    static_assert(PPW4 == PPF && NF4 >= 3 && NCH % 2 == 0 && NQ >= 2 && NKS4 % K4 == 0, "ring bookkeeping");
    static_assert(MODE == 0 || MS == 1, "taping / backward: every workgroup would store its share of h1 (a run-time count of stores in the counted waits)");
    constexpr int FL = NKS + NF4 - 1;         // last fill of the stream: fills 0 .. NKS - 1 = f.2 k-steps, NKS .. FL = f.4 fills
    constexpr int NF0 = 3 * NST0;             // MFMAs of f.0 per chunk
    constexpr int EP0 = NQ == 4 ? 60 : 24;    // first of the 24 slots of f.0's epilogue among the chunk's 24 NQ
    static_assert(c1_f0slot(NQ, NST0, NF0 - 1) < EP0 && EP0 + 24 <= 24 * NQ, "f.0 and its epilogue inside the chunk");
    // byte offsets of the LDS regions: tables | ring | f.0 double buffer | window.  (T, staged at the end, starts behind the tables.)
    constexpr int TABS = ((4 * HID + MP4) * 4 + 1023) / 1024 * 1024;
    constexpr int RINGB = TABS;
    constexpr int W0B = RINGB + 3 * SLOT;
    constexpr int WINB = W0B + 2 * G0 * 1024;

    extern __shared__ __attribute__((aligned(16))) char lds1[];
    _Float16* win = reinterpret_cast<_Float16*>(lds1 + WINB);
    float* t_rs0 = reinterpret_cast<float*>(lds1);
    float* t_b0 = t_rs0 + HID;
    float* t_rs2 = t_b0 + HID;
    float* t_b2 = t_rs2 + HID;
    float* t_rs4 = t_b2 + HID;
    const int ms_row0 = MS > 1 ? (int)blockIdx.y * MR : 0;       // first h2 row of this workgroup

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the wave's pixel tile
    const int kl = lane >> 5, ml = lane & 31;
    const int W = a.W, H = a.H, HW = g.HW;
    const int tb = blockIdx.x;
    const long gp0 = (long)tb * 128;
    const long n0 = g.NI == 1 ? gp0 / HW : (long)tb * g.NI;
    const int y0 = g.NI == 1 ? (int)((gp0 - n0 * HW) >> g.wshift) : 0;
    const int submask = (1 << g.lsub) - 1;
    // TAPE: this wave's 32-pixel tile of the fp16 tensors (uniform) + the lane's place in a pair of rows; its sign words
    [[maybe_unused]] char* tb1 = nullptr;
    [[maybe_unused]] char* tb2 = nullptr;
    [[maybe_unused]] unsigned short* mb1 = nullptr;
    [[maybe_unused]] unsigned short* mb2 = nullptr;
    [[maybe_unused]] const unsigned tlane = ((4 * kl + (lane & 1)) * 32 + (ml & ~1)) * 2;
    [[maybe_unused]] const unsigned psel = (lane & 1) ? 0x03020706u : 0x05040100u;
    [[maybe_unused]] const long mstride = (long)a.N * HW * 2;      // sign words from one 32-row tile to the next
    [[maybe_unused]] const float nscale = -a.out_scale;
    [[maybe_unused]] const unsigned blane = kl * 512 + ml * 4;      // BWD: the lane's 4 bytes in the 128-byte line of row 4 kl (+ row * 128)
    if constexpr (MODE != 0) {
        const long tile32 = (gp0 >> 5) + wid;
        tb1 = reinterpret_cast<char*>(a.tape_h1) + tile32 * (HID * (BWD ? 128 : 64));
        tb2 = reinterpret_cast<char*>(a.tape_h2) + tile32 * (HID * (BWD ? 128 : 64));
        // (the sign words of the FIRST hidden layer this launch computes: written in TAPE -- h1's; read in BWD -- h2's, whose
        // gradient the transposed network's first layer produces)
        mb1 = (BWD ? a.mask2 : a.mask1) + (gp0 + wid * 32) * 2;
        mb2 = (BWD ? a.mask1 : a.mask2) + (gp0 + wid * 32) * 2;
    }
    // (values v0, v1 = the NEGATED, scaled activations of rows r, r + 1 of this lane's pixel)
    [[maybe_unused]] auto tape_half = [&](unsigned x) -> unsigned {      // packed hi halves (of -16 h) -> packed fp16 h
        const h2 ns = {(_Float16)nscale, (_Float16)nscale};
        return __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, x) * ns);
    };
    [[maybe_unused]] auto tape_pack = [&](float v0, float v1) -> unsigned {
        const f32x2_t vv = {v0, v1};
        return tape_half(__builtin_bit_cast(unsigned, __builtin_convertvector(vv, h2)));
    };
    [[maybe_unused]] auto tape_nbr = [&](unsigned tx) -> unsigned { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)tx, 0xB1, 0xF, 0xF, true); };      // quad_perm [1, 0, 3, 2]
    [[maybe_unused]] auto tape_sel = [&](unsigned nb, unsigned tx) -> unsigned { return __builtin_amdgcn_perm(nb, tx, psel); };
#ifdef C1_DBG_TAPE_NOSTORE
    constexpr bool TST = false, BST = false;           // (timing experiments only: the tape's arithmetic without its stores)
#else
    constexpr bool TST = TAPE, BST = BWD;
#endif
    [[maybe_unused]] auto bwd_put = [&](char* tb, int c, int p, const f32x2_t& v) {      // pair p of chunk c as fp32: rows 32 c + 8 (p / 2) + 4 kl + 2 (p % 2), + 1
        float* q = reinterpret_cast<float*>(tb + (unsigned)(32 * c + 8 * (p >> 1) + 2 * (p & 1)) * 128u + blane);
#ifdef C1_BWD_NT_STORES
        if (BST) { __builtin_nontemporal_store(v[0], q); __builtin_nontemporal_store(v[1], q + 32); }
#else
        if (BST) { q[0] = v[0]; q[32] = v[1]; }
#endif
        else asm volatile("" ::"v"(v));
    };
    [[maybe_unused]] auto mask_get = [&](const unsigned short* mb, int c) -> unsigned { return (unsigned)mb[(long)c * mstride + ml * 2 + kl]; };
    [[maybe_unused]] auto tape_put = [&](char* tb, int c, int p, unsigned to) {       // pair p of chunk c: rows 32 c + 8 (p / 2) + 4 kl + 2 (p % 2) (+ 1)
        if (TST) *reinterpret_cast<unsigned*>(tb + (unsigned)(32 * c + 8 * (p >> 1) + 2 * (p & 1)) * 64u + tlane) = to;
        else asm volatile("" ::"v"(to));
    };
    [[maybe_unused]] auto mask_put = [&](unsigned short* mb, int c, unsigned w) {
        if (TST) mb[(long)c * mstride + ml * 2 + kl] = (unsigned short)w;
        else asm volatile("" ::"v"(w));
    };

    [[maybe_unused]] const _Float16* W0 = (const _Float16*)a.w0;
    const long w0_plane = (long)G0 * HID * 8;
    const float* rs0 = (const float*)((const char*)a.w0 + sh2_rowscale_off(G0 * 8, HID));
    [[maybe_unused]] const _Float16* W2 = (const _Float16*)a.w2;
    constexpr long w2_plane = (long)HID * HID;
    const float* rs2 = (const float*)((const char*)a.w2 + sh2_rowscale_off(HID, HID));
    [[maybe_unused]] const _Float16* W4 = (const _Float16*)a.w4;
    constexpr long w4_plane = (long)HID * MP4;
    const float* rs4 = (const float*)((const char*)a.w4 + sh2_rowscale_off(HID, MP4));

    // ---- the weight stream.  Fill f of the ring: f < NKS the k-step f of the f.2 image (16 pieces per plane), else fill f - NKS of
    // the f.4 image (K4 k-steps, PP4 pieces per plane).  A wave issues 8 pieces per fill (a surplus piece repeats the wave's last).
    // The pieces of a wave are contiguous in the image and in the slot (1 KiB apart in both), and the instruction's immediate offset
    // applies to BOTH addresses: one base pair (source pointer, LDS offset) per fill serves four pieces each -- per piece the wave
    // issues the load itself instead of three scalar address instructions in front of it.
#ifndef C1_GLOBAL_DMA
    // (through buffer descriptors: scalar base + scalar offset + lane offset are added by the address unit; an LDS-DMA piece issued
    // this way takes 10 - 11 cycles out of the MFMA stream, 17 - 27 as global_load_lds -- scripts/ubench/mfma_burst.hip)
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w2), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w4 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w4), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w0), 0, 0x7fffffff, 0x00020000);
    int f_w2 = 1;                         // the fill in progress reads the f.2 image (else f.4)
    int f_soff = 0;                       // byte offset of this wave's first piece in that image
#else
    const _Float16* f_src = nullptr;      // of the fill in progress: this wave's first source half (uniform; + lane * 8)
#endif
    int f_dst = 0;                        // ... and its LDS byte offset
    // KIND (compile time): 0 = the fill is one of f.2's; 2 = one of f.4's; 1 = either (run time: the last two chunks of the f.2 loop
    // request the first f.4 fills).  The loop bodies must not branch -- a branch splits the slot structure the scheduling barriers
    // hold together -- so KIND 1 selects between the two forms with scalar selects.
    auto ring_begin = [&](int f, int slot_off, auto kind) {
        constexpr int KIND = decltype(kind)::value;
        const bool w2f = KIND == 0 || (KIND == 1 && f < NKS);
        // f.2: the wave's PPF pieces are one run -- plane wid / 2, k group wid % 2 of the k-step, the workgroup's MR rows of it
        // f.4: fill f - NKS = K4 k-steps of the workgroup's K range; RUN4: PPW4 pieces of one plane, else see ring_piece
        const int p = wid >> 1;
        const long o2 = (p * w2_plane + ((long)(2 * f + (wid & 1)) * HID + ms_row0) * 8) * 2;
        const long o4 = ((RUN4 ? p * w4_plane + (wid & 1) * (PPW4 * 512) : 0) + ((long)(ms_row0 / 16) + (long)(f - NKS) * K4) * (2 * MP4 * 8)) * 2;
#ifndef C1_GLOBAL_DMA
        f_w2 = w2f ? 1 : 0;
        f_soff = (int)(w2f ? o2 : o4);
#else
        f_src = w2f ? W2 + o2 / 2 : W4 + o4 / 2;
#endif
        f_dst = RINGB + slot_off + ((w2f || RUN4) ? wid * (PPF * 1024) : 0);
    };
    auto ring_piece = [&](int i, auto kind) {       // piece i (0 .. PPF - 1) of the fill in progress
        constexpr int KIND = decltype(kind)::value;
        char* dst = lds1 + f_dst;
        if constexpr (!RUN4 && KIND != 0) {
            // f.4 with an odd piece count per plane (MS = 2: 7 + 7 pieces of a k-step): piece wid + 4 i of the 2 PP4, plane and place
            // worked out per piece (wave-uniform scalars; a few pieces per k-step of 21 MFMAs: their issue cost does not matter there)
            const int j = min(wid + 4 * i, 2 * PP4 - 1);
            const int pl = j >= PP4 ? 1 : 0, r = j - pl * PP4;
            const int o4 = (int)((pl * w4_plane + r * 512) * 2), d4 = j * 1024;
#ifndef C1_GLOBAL_DMA
            const bool w2f = KIND == 1 && f_w2;
            c1_bdma16o<0>(w2f ? rs_w2 : rs_w4, lane * 16, f_soff + (w2f ? i * 1024 : o4), dst + (w2f ? i * 1024 : d4));
#else
            static_assert(KIND == 2 || RUN4, "global-form DMA: debug builds of the full-height instances only");
            c1_dma16(f_src + o4 / 2 + lane * 8, dst + d4);
#endif
        } else {
#ifndef C1_GLOBAL_DMA
            const __amdgpu_buffer_rsrc_t rs = KIND == 0 ? rs_w2 : (KIND == 2 ? rs_w4 : (f_w2 ? rs_w2 : rs_w4));
            const int sb = i < 4 ? f_soff : f_soff + 4096;        // (the piece index is a constant after unrolling: the switch folds)
            char* db = i < 4 ? dst : dst + 4096;
            switch (i & 3) {
            case 0: c1_bdma16o<0>(rs, lane * 16, sb, db); break;
            case 1: c1_bdma16o<1024>(rs, lane * 16, sb, db); break;
            case 2: c1_bdma16o<2048>(rs, lane * 16, sb, db); break;
            default: c1_bdma16o<3072>(rs, lane * 16, sb, db); break;
            }
#else
            const _Float16* src = f_src + lane * 8;
            const _Float16* sb = i < 4 ? src : src + 2048;
            char* db = i < 4 ? dst : dst + 4096;
            switch (i & 3) {
            case 0: c1_dma16o<0>(sb, db); break;
            case 1: c1_dma16o<1024>(sb, db); break;
            case 2: c1_dma16o<2048>(sb, db); break;
            default: c1_dma16o<3072>(sb, db); break;
            }
#endif
        }
    };
    using K_F2 = std::integral_constant<int, 0>;
    using K_ANY = std::integral_constant<int, 1>;
    using K_F4 = std::integral_constant<int, 2>;
    // f.0 rows of chunk c (32 rows, both planes, G0 groups: G0 KiB) -> buffer c & 1 as [plane][group][32 rows][8]; piece j = groups
    // 2 j', 2 j' + 1 of plane j / (G0 / 2) (the two half-waves read one group each)
    auto w0_piece = [&](int c, int i) {
        const int j = min(wid + 4 * i, G0 - 1);
        const int p = j >= G0 / 2 ? 1 : 0, gp = j - p * (G0 / 2);
#ifndef C1_GLOBAL_DMA
        c1_bdma16o<0>(rs_w0, (kl * HID + ml) * 16, (int)((p * w0_plane + ((long)(2 * gp) * HID + 32 * c) * 8) * 2), lds1 + W0B + (c & 1) * (G0 * 1024) + j * 1024);
#else
        const _Float16* src = W0 + p * w0_plane + ((long)(2 * gp + kl) * HID + 32 * c + ml) * 8;
        c1_dma16(src, lds1 + W0B + (c & 1) * (G0 * 1024) + j * 1024);
#endif
    };
    GH_STAMP(0);
    GH_WG_BEGIN();
    // first what f.0 of chunk 0 needs (its rows, the tables, the window)
#pragma unroll
    for (int i = 0; i < NP0; ++i) w0_piece(0, i);

    // ---- P0: tables and the z1 window (tile rows + one halo row / column each side, zero padded) as (hi, lo) halves in LDS; slot e =
    // (8-channel chunk, sub-image, window pixel).  Every load of the round is issued from a clamped address before the first store.
    const int nwin = g.NI * g.Wpx;
    const int nslots = g.nchunk * nwin;
    auto slot_src = [&](int e, bool& in, int& ch) {
        ch = (int)__umulhi((unsigned)e, g.m_nwin);
        const int rem = e - ch * nwin;
        const int sub = (int)__umulhi((unsigned)rem, g.m_Wpx), wp = rem - sub * g.Wpx;
        const int r = (int)__umulhi((unsigned)wp, g.m_WP), c = wp - r * g.WP;
        const int yy = y0 - 1 + r, xx = c - 1;
        const long n = n0 + sub;
        in = yy >= 0 && yy < H && xx >= 0 && xx < W && n < a.N;
        return a.x + (n < a.N ? n : (long)a.N - 1) * a.x_bs + min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1);
    };
    {
        constexpr int N0 = 2 * HID / NT, N2 = HID / NT;
        float tv0[N0], tv2[N2], tvb[N2], tv4;
#pragma unroll
        for (int i = 0; i < N0; ++i) tv0[i] = rs0[tid + NT * i];                  // rs0 | b0 are adjacent in the image
#pragma unroll
        for (int i = 0; i < N2; ++i) { tv2[i] = rs2[tid + NT * i]; tvb[i] = rs2[HID + tid + NT * i]; }
        tv4 = rs4[min(tid, MP4 - 1)];
        // the window in rounds of NT slots; the stream's first fills are REQUESTED between the first round's loads and their use:
        // they need nothing, nobody needs them before the loop, and issued here their 21 pieces and their trip to L2 overlap the
        // window's own trip and its conversion instead of standing in front of f.0 of chunk 0
        // (two rounds per pass: a window of more than NT slots -- 16-pixel rows with two channel chunks, 128-pixel rows -- has both
        // rounds' loads in flight together instead of one global round trip behind the other)
        auto win_put = [&](int e, bool live, bool in, int ch, const float (&v)[8]) {
            h8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float vv = (in && ch * 8 + q < a.Cin) ? canon_nan(v[q] * (MODE != 0 ? a.in_scale : SH2_ACT_SCALE)) : 0.f;
                _Float16 x0, x1;
                sh2_split(vv, x0, x1);
                hi[q] = x0; lo[q] = x1;
            }
            if (live) {
                *reinterpret_cast<h8*>(win + (long)e * 8) = hi;
                *reinterpret_cast<h8*>(win + g.winplane + (long)e * 8) = lo;
            }
        };
        for (int e0 = 0; e0 < nslots; e0 += 2 * NT) {
            const bool two = e0 + NT < nslots;
            const int eA = min(e0 + tid, nslots - 1), eB = min(e0 + NT + tid, nslots - 1);
            bool inA, inB = false; int chA, chB = 0;
            const float* xa = slot_src(eA, inA, chA);
            float vA[8], vB[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) vA[q] = xa[(long)min(chA * 8 + q, a.Cin - 1) * HW];
            if (two) {
                const float* xb = slot_src(eB, inB, chB);
#pragma unroll
                for (int q = 0; q < 8; ++q) vB[q] = xb[(long)min(chB * 8 + q, a.Cin - 1) * HW];
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) vB[q] = 0.f;
            }
            if (e0 == 0) {
                C1_FENCE();
                ring_begin(0, 0, K_F2{});
#pragma unroll
                for (int i = 0; i < PPF; ++i) ring_piece(i, K_F2{});
#pragma unroll
                for (int i = 0; i < NP0; ++i) w0_piece(1, i);
                ring_begin(1, SLOT, K_F2{});
#pragma unroll
                for (int i = 0; i < PPF; ++i) ring_piece(i, K_F2{});
                ring_begin(2, 2 * SLOT, K_F2{});
                ring_piece(0, K_F2{});
                ring_piece(1, K_F2{});
                C1_FENCE();
            }
            win_put(eA, e0 + tid < nslots, inA, chA, vA);
            if (two) win_put(eB, e0 + NT + tid < nslots, inB, chB, vB);
        }
        // (signs: the activations travel NEGATED from the first epilogue on -- sh.h nrelu_bits, cnet_sh.hip)
#pragma unroll
        for (int i = 0; i < N0; ++i) t_rs0[tid + NT * i] = canon_nan(-tv0[i]);
#pragma unroll
        for (int i = 0; i < N2; ++i) { t_rs2[tid + NT * i] = canon_nan(tv2[i]); t_b2[tid + NT * i] = canon_nan(-tvb[i]); }
        if (tid < MP4) t_rs4[tid] = canon_nan(-tv4);
    }
    // f.0's rows of chunk 0 have landed: everything but the 2 PPF + 2 + NP0 pieces of the fills requested behind them (in-order counter)
    // (a plain barrier: __syncthreads() puts a vmcnt(0) in front of it -- it counts the LDS-DMA pieces as LDS writes to be fenced --
    // and would wait here for the 18 + NP0 pieces that have the whole of f.0 of chunk 0 to land)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPF + 2 + NP0) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GH_STAMP(1);

    // ---- per-lane constants of the contractions
    // f.0: window byte address of this lane's pixel for k-step st (the lane's group 2 st + kl = (chunk, tap); past 9 * nchunk: zero
    // weights, offset 0)
    const int qpix = wid * 32 + ml;
    const int pbase = [&]() {
        const int sub = qpix >> g.lsub, qq = qpix & submask;
        return (sub * g.Wpx + (qq >> g.wshift) * g.WP + (qq & (W - 1))) * 16;
    }();
    int woff[NST0];
#pragma unroll
    for (int st = 0; st < NST0; ++st) {
        const int gk = 2 * st + kl;
        const int ch = gk / 9, tap = gk - ch * 9;
        const int dy = tap / 3, dx = tap - dy * 3;
        woff[st] = WINB + pbase + (ch < g.nchunk ? (ch * g.NI * g.Wpx + dy * g.WP + dx) * 16 : 0);
    }
    const int wlo = g.winplane * 2;                        // bytes from the window's hi plane to its lo plane
    const int a0lane = W0B + lane * 16;                    // f.0 A fragment: + buffer * G0 KiB + (plane * G0 + 2 st) * 512
    const int a2lane = RINGB + (kl * MR + ml) * 16;        // f.2 A fragment: + slot + plane * SLOT / 2 + row tile * 512
    const int a4lane = RINGB + (kl * MP4 + ml) * 16;       // f.4 A fragment: + slot + plane * K4 MP4 32 + k-step * MP4 32 + row tile * 512

    auto ldA0 = [&](int buf, int st, h8& hi, h8& lo) {
        const char* p = lds1 + a0lane + buf * (G0 * 1024) + 2 * st * 512;
        hi = *reinterpret_cast<const h8*>(p);
        lo = *reinterpret_cast<const h8*>(p + G0 * 512);
    };
    auto ldB0 = [&](int st, h8& hi, h8& lo) {
        const char* p = lds1 + woff[st];
        hi = *reinterpret_cast<const h8*>(p);
        lo = *reinterpret_cast<const h8*>(p + wlo);
    };
    [[maybe_unused]] unsigned E_mw = 0u;     // TAPE: the sign word being built (value k = 4 gq + t shifted in at bit 0: ends up in bit 15 - k)
    // epilogue of f.0 for group gq (rows 8 gq + 4 kl + t of chunk c): -h1 = -relu(.) as halves of the B fragments
    auto epi1 = [&](const f32x16_t& acc, int c, int gq, h8 (&bh)[2], h8 (&bl)[2], [[maybe_unused]] unsigned mw) {
        const int o = 32 * c + 8 * gq + 4 * kl;
        const f32x4_t rs = *reinterpret_cast<const f32x4_t*>(t_rs0 + o);
        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(t_b0 + o);
        f32x4_t v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float tt = fmaf(acc[4 * gq + t], rs[t], bb[t]);
            v[t] = BWD ? ((mw >> (15 - (4 * gq + t))) & 1u ? tt : 0.f) : nrelu_bits(tt);
        }
        c1_frag(v, bh[gq >> 1], bl[gq >> 1], gq & 1);
        if constexpr (BWD) {
#pragma unroll
            for (int h = 0; h < 2; ++h) bwd_put(tb1, c, 2 * gq + h, f32x2_t{v[2 * h], v[2 * h + 1]} * nscale);
        }
        if constexpr (TAPE) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned tx = tape_pack(v[2 * h], v[2 * h + 1]);
                tape_put(tb1, c, 2 * gq + h, tape_sel(tape_nbr(tx), tx));
                E_mw = __builtin_amdgcn_alignbit(E_mw, __float_as_uint(v[2 * h]), 31);
                E_mw = __builtin_amdgcn_alignbit(E_mw, __float_as_uint(v[2 * h + 1]), 31);
            }
            if (gq == 3) mask_put(mb1, c, E_mw);
        }
    };

    // ---- f.0 of chunk 0 on its own (the only MFMAs of the kernel without f.2 or f.4 MFMAs around them)
    h8 Bh[2], Bl[2];           // B fragments of the chunk f.2 is multiplying: k-steps 2 c, 2 c + 1
    // f.0's B fragments -- this wave's window pixels, (hi, lo) per k-step -- are the same for every chunk of h1 rows: they stay in
    // registers for the whole kernel (8 NST0 of the 512; re-read per chunk they were a fifth of the LDS traffic of the f.2 loop,
    // which runs within 20 - 30 % of the LDS bandwidth)
    h8 B0H[NST0], B0L[NST0];
#pragma unroll
    for (int st = 0; st < NST0; ++st) ldB0(st, B0H[st], B0L[st]);
    {
        f32x16_t acc1;
#pragma unroll
        for (int st = 0; st < NST0; ++st) {
            h8 ah, al;
            ldA0(0, st, ah, al);
            if (st == 0) c1_mfma_v0(acc1, ah, B0H[st]); else c1_mfma_v(acc1, ah, B0H[st]);
            c1_mfma_v(acc1, ah, B0L[st]);
            c1_mfma_v(acc1, al, B0H[st]);
        }
        c1_settle(acc1);
        unsigned mw0 = 0u;
        if constexpr (BWD) mw0 = mask_get(mb1, 0);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) epi1(acc1, 0, gq, Bh, Bl, mw0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first fills have landed ...
    __syncthreads();          // ... everybody's; and every wave is done with f.0 buffer 0 before the loop refills it
    GH_STAMP(2);

    // ---- P2 with P1 of the next chunk inside: acc2[rt] += W2'[rows of tile rt, chunk c] h1[chunk c]
    f32x16_t acc2[NRT2];
#pragma unroll
    for (int i = 0; i < NRT2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;

    int s_cur = 0;             // ring slot (byte offset) of the k-step being multiplied
    int s_fill = 2 * SLOT;     // ... of the fill in progress
    // ---- issue discipline.  With one wave per SIMD nothing but this wave's own stream keeps the matrix pipe busy: an MFMA occupies it
    // for 32 cycles, and whatever the wave issues before its NEXT MFMA has to fit into those 32 cycles (a 16-byte LDS read ~ 8 of
    // them, a VALU instruction ~ 4, an LDS-DMA piece more).  hipcc clusters loads (eight ds_read_b128 back to back = 64 cycles of
    // issue behind one MFMA: the first version of this loop ran at 0.73 of the MFMA rate).  So the stream is written SLOT by slot:
    // one MFMA + at most a few fillers, closed by a scheduling barrier that nothing crosses.
    // A fragments of a quad = four row tiles x (hi, lo), SINGLE-buffered: the hi fragment of tile i is read by slots i and 4 + i of a
    // quad and reloaded (next quad's) in slot 4 + i; the lo fragment by slot 8 + i, reloaded there.  Every read of a ring slot is thus
    // issued before the TOP that hands the slot back to the stream (the TOP sits at the head of a k-step's last quad).
    h8 AH[4], AL[4];
    auto ldA2one = [&](int slot_off, int q4, int i, int pl, h8& dst) {
        dst = *reinterpret_cast<const h8*>(lds1 + a2lane + slot_off + (q4 * 4 + i) * 512 + pl * (SLOT / 2));
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) { ldA2one(0, 0, i, 0, AH[i]); ldA2one(0, 0, i, 1, AL[i]); }
    // f.0's A fragments, single-buffered as well: the three MFMAs of a k-step are (A0h, B0l), (A0h, B0h), (A0l, B0h), and each A fragment
    // is reloaded for the next k-step right behind its last use
    h8 A0h, A0l;
    ldA0(1, 0, A0h, A0l);
    // ---- the epilogue of a 32-row block (16 values per lane -> the B fragments of two k-steps), software-pipelined over 24 slots.
    // With one wave per SIMD a DEPENDENT VALU instruction waits out its producer's latency in the issue stream, and that wait comes
    // straight out of the next MFMA's time (five dependent instructions per pair of values, as one piece behind one MFMA, cost 6 k of
    // the 74 k cycles of the f.2 loop).  So the 8 pairs of values move through five stages -- S1 scale + bias, S2 relu, S3 hi halves,
    // S4 residuals, S5 lo halves -- one stage per slot: pair p is at stage S in tick 2 p + 2 + S, every tick issues four
    // INDEPENDENT instructions (S1, S3, S5 of three pairs in odd ticks; S2, S4 of two pairs in even ones), and the tables of row
    // group gq are requested in tick 4 gq, three ticks ahead of their first use.
    // (agpr: the block is one of h2's, in AGPRs; its values are read with an explicit v_accvgpr_read in S1 -- left to the register
    // allocator, the copies of ALL blocks of the fully unrolled f.4 phase move to the head of that one basic block, 256 VGPRs live
    // at once, and the loop-invariant addresses of the whole kernel get spilled)
    f32x4_t E_rs[2], E_bb[2];         // [gq & 1]
    float E_v[8][2], E_m[8][2];       // per pair: value (S1, S2), residual (S4)
    unsigned E_x[8];                  // per pair: the packed hi halves (S3)
    // TAPE: four more stages per pair behind S3 -- T1 the hi halves times -1/16 (one packed fp16 multiply: exact but for results in
    // fp16's subnormal range, |h| < 6.1e-5, where it rounds a second time -- k_cnet converts the fp32 product, one rounding; both are
    // fp16 roundings of h within 2^-24) + the pair's two sign bits, T2 the neighbouring lane's, T3 select, T4 store: pair p is at
    // stage T in tick 2 p + 5 + T (the last store in tick 23), the block's sign word goes out in tick 22
    // BWD: S2 is the mask select (the block's sign word mw: value k in bit 15 - k), and two more stages per pair behind it -- T1 the
    // two values times -out_scale, T2 their two stores: pair p at stage T in tick 2 p + 4 + T (the last stores in tick 20)
    [[maybe_unused]] unsigned E_tx[8], E_tn[8];
    [[maybe_unused]] f32x2_t E_tm[8];
    auto epi_tick = [&](const f32x16_t& acc, const float* trs, const float* tbb, int c, int e, h8 (&bh)[2], h8 (&bl)[2], auto agpr,
                        [[maybe_unused]] char* tb, [[maybe_unused]] unsigned short* mb, [[maybe_unused]] unsigned mw) {
        if ((e & 3) == 0 && e < 16) {
            const int gq = e >> 2, o = 32 * c + 8 * gq + 4 * kl;
            E_rs[gq & 1] = *reinterpret_cast<const f32x4_t*>(trs + o);
            E_bb[gq & 1] = *reinterpret_cast<const f32x4_t*>(tbb + o);
        }
#pragma unroll
        for (int S = 1; S <= 5; ++S) {
            if ((e - 2 - S) & 1) continue;
            const int p = (e - 2 - S) / 2;
            if (e - 2 - S < 0 || p > 7) continue;
            const int gq = p >> 1, t0 = 2 * (p & 1);
            if (S == 1) {
                float a0, a1;
                if constexpr (decltype(agpr)::value) {
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0) : "a"(acc[4 * gq + t0]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1) : "a"(acc[4 * gq + t0 + 1]));
                } else {
                    a0 = acc[4 * gq + t0]; a1 = acc[4 * gq + t0 + 1];
                }
                E_v[p][0] = fmaf(a0, E_rs[gq & 1][t0], E_bb[gq & 1][t0]);
                E_v[p][1] = fmaf(a1, E_rs[gq & 1][t0 + 1], E_bb[gq & 1][t0 + 1]);
                asm volatile("" : "+v"(E_v[p][0]), "+v"(E_v[p][1]));      // (each stage is computed in ITS slot: the optimizer otherwise sinks it to its use)
            } else if (S == 2) {
                if constexpr (BWD) {      // 0 or all ones from the value's bit of the sign word, then AND
                    E_v[p][0] = __uint_as_float(__float_as_uint(E_v[p][0]) & (unsigned)__builtin_amdgcn_sbfe((int)mw, 15 - 2 * p, 1));
                    E_v[p][1] = __uint_as_float(__float_as_uint(E_v[p][1]) & (unsigned)__builtin_amdgcn_sbfe((int)mw, 14 - 2 * p, 1));
                } else {
                    E_v[p][0] = nrelu_bits(E_v[p][0]);
                    E_v[p][1] = nrelu_bits(E_v[p][1]);
                }
                asm volatile("" : "+v"(E_v[p][0]), "+v"(E_v[p][1]));
            } else if (S == 3) {
                const f32x2_t vv = {E_v[p][0], E_v[p][1]};
                E_x[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(vv, h2));
                asm volatile("" : "+v"(E_x[p]));
            } else if (S == 4) {
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(E_m[p][0]) : "v"(E_x[p]), "v"(E_v[p][0]));
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(E_m[p][1]) : "v"(E_x[p]), "v"(E_v[p][1]));
            } else {
                const f32x2_t mm = {E_m[p][0], E_m[p][1]};
                const h2 y = __builtin_convertvector(mm, h2);
                const h2 x = __builtin_bit_cast(h2, E_x[p]);
                const int e0 = 4 * (gq & 1) + t0;
                bh[gq >> 1][e0] = x[0]; bh[gq >> 1][e0 + 1] = x[1];
                bl[gq >> 1][e0] = y[0]; bl[gq >> 1][e0 + 1] = y[1];
            }
        }
        if constexpr (TAPE) {
#pragma unroll
            for (int T = 1; T <= 4; ++T) {
                const int d = e - 5 - T;
                if (d < 0 || (d & 1) || d / 2 > 7) continue;
                const int p = d / 2;
                if (T == 1) {
                    E_tx[p] = tape_half(E_x[p]);
                    E_mw = __builtin_amdgcn_alignbit(E_mw, __float_as_uint(E_v[p][0]), 31);
                    E_mw = __builtin_amdgcn_alignbit(E_mw, __float_as_uint(E_v[p][1]), 31);
                    asm volatile("" : "+v"(E_tx[p]), "+v"(E_mw));
                } else if (T == 2) {
                    E_tn[p] = tape_nbr(E_tx[p]);
                    asm volatile("" : "+v"(E_tn[p]));
                } else if (T == 3) {
                    E_tx[p] = tape_sel(E_tn[p], E_tx[p]);
                    asm volatile("" : "+v"(E_tx[p]));
                } else {
                    tape_put(tb, c, p, E_tx[p]);
                }
            }
            if (e == 22) mask_put(mb, c, E_mw);
        }
        if constexpr (BWD) {
#pragma unroll
            for (int T = 1; T <= 2; ++T) {
                const int d = e - 4 - T;
                if (d < 0 || (d & 1) || d / 2 > 7) continue;
                const int p = d / 2;
                if (T == 1) {
                    E_tm[p] = f32x2_t{E_v[p][0], E_v[p][1]} * nscale;
                    asm volatile("" : "+v"(E_tm[p]));
                } else {
                    bwd_put(tb, c, p, E_tm[p]);
                }
            }
        }
    };

    // (kind: K_F2 for the chunks whose fills are all f.2's -- all but the last two -- else K_ANY)
    auto p2_chunk = [&](int c, auto kind) {
        const int cn = min(c + 1, NCH - 1);        // chunk whose f.0 rides along (the last iteration repeats chunk NCH - 1 for nothing)
        const int nbuf = cn & 1;
        f32x16_t acc1;
        h8 Bnh[2], Bnl[2];
        [[maybe_unused]] unsigned mwA = 0u;        // BWD: the sign word of chunk cn (requested in slot 17, used from slot EP0 + 4 on)
        // f.0's MFMA j of chunk cn and the operand reloads behind it
        auto f0_mfma = [&](int j) {
            const int st = j / 3, w = j % 3;
            const char* pa = lds1 + a0lane + nbuf * (G0 * 1024) + 2 * (st + 1) * 512;
            if (w == 0) {
                if (st == 0) c1_mfma_v0(acc1, A0h, B0L[st]); else c1_mfma_v(acc1, A0h, B0L[st]);
            } else if (w == 1) {
                c1_mfma_v(acc1, A0h, B0H[st]);
                if (st + 1 < NST0) A0h = *reinterpret_cast<const h8*>(pa);
            } else {
                c1_mfma_v(acc1, A0l, B0H[st]);
                if (st + 1 < NST0) A0l = *reinterpret_cast<const h8*>(pa + G0 * 512);
            }
        };
#pragma unroll
        for (int Q = 0; Q < 2 * NQ; ++Q) {
            const int s = Q / NQ, q4 = Q % NQ;
            const bool lastq = q4 == NQ - 1;        // the k-step's last quad: the TOP sits in front of its fourth slot
            const int f = 2 * c + s;                // k-step
            C1_FENCE();
            const int i0 = lastq ? 0 : 2 + 2 * q4;      // this quad's two pieces of the fill in progress (f + 3 behind the TOP, else f + 2)
            const int nq4 = (q4 + 1) % NQ;          // the next quad (after a TOP: the next k-step's first quad, from the slot that has just landed)
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int sl = 12 * Q + k;          // slot of the chunk
                if (lastq && k == 3) {
                    // TOP: fill f + 1 has landed (this wave's pieces: counted wait; everybody's: barrier) and every wave has read all
                    // it wanted from slot s_cur (lgkmcnt(0) before the barrier) -- which fill f + 3 may then overwrite.  It sits in
                    // front of the quad's FOURTH slot: the last reads of the slot were issued in the previous quad, three MFMAs and
                    // more ago, so the lgkmcnt(0) no longer waits out an LDS round trip (at the head of the quad it did, 32 times
                    // per tile), and the next k-step's first fragments are requested from slot 4 on.
#ifndef C1_DBG_NO_VMWAIT
                    // (TAPE: + the stores issued behind the last piece of the fill waited for -- 6 of the previous chunk's 9 at s = 0,
                    // all 9 of this chunk's at s = 1; one less each: a smaller count only waits for more.  BWD: 6 of the previous
                    // chunk's 16 stores + this chunk's sign-word load at s = 0, all 16 stores at s = 1; one less each.)
                    if (s == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPF + NP0 + (TST ? 5 : 0) + (BST ? 6 : (BWD ? 1 : 0))) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPF + (TST ? 8 : 0) + (BST ? 15 : 0)) : "memory");
#endif
#ifndef C1_DBG_NO_BARRIER
                    __builtin_amdgcn_s_barrier();
#endif
                    asm volatile("" ::: "memory");
                    s_fill = s_cur;
                    s_cur = s_cur + SLOT == 3 * SLOT ? 0 : s_cur + SLOT;
                    ring_begin(f + 3, s_fill, kind);
                    C1_FENCE();
                }
                const int i = k & 3, sw = k >> 2;
                acc2[4 * q4 + i] = C1_MFMA(sw == 2 ? AL[i] : AH[i], sw == 1 ? Bl[s] : Bh[s], acc2[4 * q4 + i]);
#ifndef C1_DBG_NO_AREAD
                if (sw == 1) ldA2one(s_cur, nq4, i, 0, AH[i]);
                if (sw == 2) ldA2one(s_cur, nq4, i, 1, AL[i]);
#endif
                // the stream: two ring pieces per quad; f.0 rows of chunk c + 2 in the quads of the chunk's first k-step but its last
#ifndef C1_DBG_NO_DMA
                if (k == (lastq ? 3 : 0)) ring_piece(i0, kind);
                if (k == (lastq ? 9 : 2)) ring_piece(i0 + 1, kind);
                if (s == 0 && !lastq && (k & 1) && q4 + (k >> 1) * (NQ - 1) < NP0) w0_piece(min(c + 2, NCH - 1), q4 + (k >> 1) * (NQ - 1));
#endif
                if constexpr (BWD) { if (sl == 17) mwA = mask_get(mb1, cn); }
                // f.0 of chunk cn (c1_f0slot); in the chunk's last quad the first operands of the chunk after (its rows landed with
                // that quad's TOP)
#ifndef C1_DBG_NO_P1
#pragma unroll
                for (int j = 0; j < NF0; ++j)
                    if (c1_f0slot(NQ, NST0, j) == sl) f0_mfma(j);
                if (Q == 2 * NQ - 1 && k >= 8) {
                    const char* pa = lds1 + a0lane + (min(c + 2, NCH - 1) & 1) * (G0 * 1024);
                    if (k == 9) A0h = *reinterpret_cast<const h8*>(pa);
                    if (k == 11) A0l = *reinterpret_cast<const h8*>(pa + G0 * 512);
                }
                // its epilogue: 24 slots from EP0 on
                if (sl >= EP0 && sl < EP0 + 24) {
#ifndef C1_DBG_NO_EPI
                    if (sl == EP0) c1_settle(acc1);
                    epi_tick(acc1, t_rs0, t_b0, cn, sl - EP0, Bnh, Bnl, std::false_type{}, tb1, mb1, mwA);
#endif
                }
#endif
                C1_FENCE();
            }
        }
#if !defined(C1_DBG_NO_P1) && !defined(C1_DBG_NO_EPI)
#pragma unroll
        for (int s = 0; s < 2; ++s) { Bh[s] = Bnh[s]; Bl[s] = Bnl[s]; }
#endif
    };
    static_assert(2 * (NCH - 3) + 1 + 3 < NKS, "the fills of the chunks before the last two are f.2's");
#pragma unroll 1
    for (int c = 0; c < NCH - 2; ++c) p2_chunk(c, K_F2{});
#pragma unroll 1
    for (int c = NCH - 2; c < NCH; ++c) p2_chunk(c, K_ANY{});
    GH_STAMP(3);

    // ---- P3: T[m][px] += W4t[m][k] h2[k][px] with h2 = relu + split of the accumulator block k / 32, two k-steps per block.
    // (slot bookkeeping continues: the last TOP of the loop above made fill NKS = the first of the f.4 image readable in s_cur)
    // MS = 1: T in VGPRs like f.0's block (asm MFMAs): h2 holds every AGPR until its last block has been consumed.  MS = 2: h2 is half
    // as large and T (seven row tiles) takes the other AGPRs through the builtin.
    constexpr bool T_ASM = NRT2 * 16 + NRT4 * 16 > 256;
    f32x16_t accT[NRT4];
    h8 Hh[2][2], Hl[2][2];      // [block & 1][k-step of the block]
    const float* trs2 = t_rs2 + ms_row0;
    const float* tbb2 = t_b2 + ms_row0;
    // BWD: the sign word of h2-position block b (mask1 of the forward: h1's), requested a block ahead of its epilogue
    [[maybe_unused]] unsigned mwB[NRT2];
    if constexpr (BWD) { mwB[0] = mask_get(mb2, 0); mwB[1] = mask_get(mb2, 1); }
#pragma unroll
    for (int e = 0; e < 24; ++e) epi_tick(acc2[0], trs2, tbb2, 0, e, Hh[0], Hl[0], std::true_type{}, tb2, mb2, BWD ? mwB[0] : 0u);
    h8 A4H[NRT4], A4L[NRT4];
    auto ldA4one = [&](int slot_off, int kk, int i, int pl, h8& dst) {
        dst = *reinterpret_cast<const h8*>(lds1 + a4lane + slot_off + kk * (MP4 * 32) + i * 512 + pl * (K4 * MP4 * 32));
    };
#pragma unroll
    for (int i = 0; i < NRT4; ++i) { ldA4one(s_cur, 0, i, 0, A4H[i]); ldA4one(s_cur, 0, i, 1, A4L[i]); }
    if constexpr (!T_ASM) {
#pragma unroll
        for (int i = 0; i < NRT4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accT[i][r] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < NKS4; ++ks) {
        const int c = ks >> 1, s = ks & 1;
        const int F = NKS + ks / K4, kk = ks % K4;        // fill being read, k-step inside it
        C1_FENCE();
        // pieces of the fill in progress: two of fill F + 3 behind the TOP (in the fill's last k-step: its last slots), the others of fill
        // F + 2 over the last slots of its first K4 - 1 k-steps -- or, with one k-step per fill, in slots 0 and 2 in front of the TOP
        constexpr int PA = 2;
        static_assert(K4 > 1 || PPW4 - PA <= 2, "one k-step per fill: its other pieces in slots 0 and 2");
        const int ff = kk == K4 - 1 ? F + 3 : F + 2;
        constexpr int PER = K4 > 1 ? (PPW4 - PA + K4 - 2) / (K4 - 1) : 0;
        const int i0 = kk == K4 - 1 ? 0 : PA + kk * PER, i1 = kk == K4 - 1 ? PA : min(PPW4, PA + (kk + 1) * PER);
        const int cn = min(c + 1, NRT2 - 1);
        constexpr int NM = 3 * NRT4;
        constexpr int TPS = (24 + 2 * NM - 1) / (2 * NM);      // epilogue ticks per slot: the next block's 24 inside this block's 2 NM slots
        static_assert(NM - PA >= 4, "the pieces behind the TOP");
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            // TOP (as in the loop above: in front of the k-step's fourth slot -- or, with fewer than three row tiles, in front of the
            // first slot that reloads a fragment for the NEXT k-step, which must come out of the slot that has just landed)
            constexpr int TOPK = NRT4 < 3 ? NRT4 : 3;
            if (kk == K4 - 1 && F + 1 <= FL && k == TOPK) {
                // (TAPE: 19 stores are issued between the last piece of fill F + 1 -- at the end of fill F - 1's third k-step -- and here.
                // BWD: more than a fill's worth -- 4 blocks x (16 stores + a sign-word load) -- so the largest count there is, 63, is valid:
                // the pieces waited for are older than the 63 youngest operations.)
                if (F + 2 <= FL) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BST ? 63 : PPW4 + (TST ? 16 : 0)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                s_fill = s_cur;
                s_cur = s_cur + SLOT == 3 * SLOT ? 0 : s_cur + SLOT;
                if (F + 3 <= FL) ring_begin(F + 3, s_fill, K_F4{});
                C1_FENCE();
            }
            if (K4 == 1 && F + 2 <= FL && k < 3 && !(k & 1) && PA + k / 2 < PPW4) ring_piece(PA + k / 2, K_F4{});
            const int i = k % NRT4, sw = k / NRT4;
            if constexpr (T_ASM) {
                if (ks == 0 && sw == 0) c1_mfma_v0(accT[i], A4H[i], Hh[0][0]);
                else c1_mfma_v(accT[i], sw == 2 ? A4L[i] : A4H[i], sw == 1 ? Hl[c & 1][s] : Hh[c & 1][s]);
            } else {
                accT[i] = C1_MFMA(sw == 2 ? A4L[i] : A4H[i], sw == 1 ? Hl[c & 1][s] : Hh[c & 1][s], accT[i]);
            }
            if (ks + 1 < NKS4 && sw == 1) ldA4one(s_cur, (kk + 1) % K4, i, 0, A4H[i]);
            if (ks + 1 < NKS4 && sw == 2) ldA4one(s_cur, (kk + 1) % K4, i, 1, A4L[i]);
            // h2 of the next block: its epilogue's 24 ticks over this block's two k-steps, TPS per slot
            if constexpr (BWD) { if (s == 0 && k == 0 && c + 2 < NRT2) mwB[c + 2] = mask_get(mb2, c + 2); }
#pragma unroll
            for (int u = 0; u < TPS; ++u) {
                const int e = (s * NM + k) * TPS + u;
                if (c + 1 < NRT2 && e < 24) epi_tick(acc2[cn], trs2, tbb2, cn, e, Hh[cn & 1], Hl[cn & 1], std::true_type{}, tb2, mb2, BWD ? mwB[cn] : 0u);
            }
            if (ff <= FL && k >= NM - (i1 - i0)) ring_piece(i0 + k - (NM - (i1 - i0)), K_F4{});
            C1_FENCE();
        }
    }
    GH_STAMP(4);
    if constexpr (T_ASM) {
#pragma unroll
        for (int i = 0; i < NRT4; ++i) c1_settle(accT[i]);
    }
    __syncthreads();           // every wave is done with the ring: it becomes the staging area of T

    // ---- P4: T -> LDS as fp32, PIXEL-major [pixel][RP rows] (row scale applied), then the 9-tap sums.  A lane's four consecutive
    // rows of a pixel are one 16-byte store, and a thread of the tap sums takes one pixel and FOUR output channels: row m = tap * Cout
    // + co puts them side by side, one 16-byte read per tap -- a quarter of the LDS instructions of the [row][pixel] form in both
    // halves of the phase.  RP / 4 is odd: the 16-byte groups of consecutive pixels fall on distinct banks.
    float* T = reinterpret_cast<float*>(lds1 + TABS);       // (behind the tables: t_rs4 is read while T is written)
    constexpr int CGW = BWD ? 2 : 4;            // output channels a tap-sum thread takes per read (BWD: C / 2 = 6 channels)
    using tvec = std::conditional_t<CGW == 4, f32x4_t, f32x2_t>;
    const int Cout = g.Cg;                      // (a multiple of CGW: cnet1w_takes)
    const int rows = 9 * Cout;
    const int RP = cnet_trow(rows);             // floats per pixel: whole 16-byte groups, an odd number of them
    {
        float* dst = T + (wid * 32 + ml) * RP + 4 * kl;
#pragma unroll
        for (int i = 0; i < NRT4; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int r0 = i * 32 + 8 * gq;
                const f32x4_t rsv = *reinterpret_cast<const f32x4_t*>(t_rs4 + r0 + 4 * kl);
                f32x4_t v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = accT[i][4 * gq + t] * rsv[t];
                if (r0 + 4 * kl < rows) *reinterpret_cast<f32x4_t*>(dst + r0) = v;      // (rows beyond 9 Cout are the image's zero padding)
            }
    }
    __syncthreads();
    GH_STAMP(5);
    // (MS = 2: each half leaves its own copy of the sums, k_cnet's layout -- the finishing kernel adds them in a fixed order)
    const long msN = MS > 1 ? (long)blockIdx.y * a.N : 0;
    const long mstile = MS > 1 ? (long)blockIdx.y * g.tiles : 0;
    float* hpart = a.scratch;
    float* hup = a.scratch + (long)MS * a.N * a.Cout * HW;
    float* hdn = hup + (long)MS * g.tiles * a.Cout * W;
    const int ngrp = Cout / CGW;
    {
        // a thread keeps ITS pixel (tid & 127) and walks the groups of four channels tid >> 7, + 2, ...
        const int q = tid & 127;
        const int sub = q >> g.lsub, qq = q & submask;
        const int r = qq >> g.wshift, x = qq & (W - 1);
        const long n = n0 + sub;
        int off[9];
        bool ok[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int tap = dy * 3 + dx;              // out(r, x) += T[source (r + dy - 1, x + dx - 1)][tap (dy, dx)]
                ok[tap] = r + dy - 1 >= 0 && r + dy - 1 < g.R && x + dx - 1 >= 0 && x + dx - 1 < W;
                off[tap] = (q + (ok[tap] ? (dy - 1) * W + (dx - 1) : 0)) * RP + tap * Cout;
            }
        if (n < a.N) {
            float* hp = hpart + ((msN + n) * a.Cout) * HW + (long)(y0 + r) * W + x;
            for (int cg = tid >> LPXT; cg < ngrp; cg += NT >> LPXT) {
                const float* tp = T + CGW * cg;
                tvec v[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) v[tap] = *reinterpret_cast<const tvec*>(tp + off[tap]);
                tvec sum;
#pragma unroll
                for (int j = 0; j < CGW; ++j) sum[j] = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int j = 0; j < CGW; ++j) sum[j] += ok[tap] ? v[tap][j] : 0.f;
#pragma unroll
                for (int j = 0; j < CGW; ++j) c1_publish<FIN>(hp + (long)(CGW * cg + j) * HW, sum[j]);
            }
        }
    }
    // halo rows (NI = 1 only): what the tile's first row gives to image row y0 - 1, its last row to row y0 + R
    if (g.NI == 1 && g.R < H) {
        const int hitems = 2 * ngrp * W;
        // (items go to the threads of the tap sums' LIGHTER half first: those took one group of channels where the others took two)
        for (int e = (tid + NT / 2) & (NT - 1); e < hitems; e += NT) {
            const int dn = e >= ngrp * W;
            const int rem = e - dn * (ngrp * W);
            const int cg = rem >> g.wshift, x = rem & (W - 1);
            if (dn ? (y0 + g.R >= H) : (y0 == 0)) continue;
            const int rsrc = dn ? g.R - 1 : 0;
            const int dyt = dn ? 0 : 2;                 // filter row applied by the outside pixel to this source row
            tvec v[3]; bool ok[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                ok[dx] = x + dx - 1 >= 0 && x + dx - 1 < W;
                v[dx] = *reinterpret_cast<const tvec*>(T + (rsrc * W + (ok[dx] ? x + dx - 1 : x)) * RP + (dyt * 3 + dx) * Cout + CGW * cg);
            }
#pragma unroll
            for (int j = 0; j < CGW; ++j) {
                const float sacc = (ok[0] ? v[0][j] : 0.f) + (ok[1] ? v[1][j] : 0.f) + (ok[2] ? v[2][j] : 0.f);
                c1_publish<FIN>((dn ? hdn : hup) + ((mstile + tb) * a.Cout + CGW * cg + j) * W + x, sacc);
            }
        }
    }
    GH_STAMP(6);
    if constexpr (FIN) {
        // ---- fused finishing.  The sums above went out as agent-scope atomic stores (written through the XCD's L2); once this
        // wave's are acknowledged (vmcnt) and every wave is here, thread 0 bumps the arrival counter of its own tile and of the
        // neighbouring tiles of the image -- they need this tile's halo rows -- and whoever completes a counter finishes THAT tile
        // with the finishing kernel's own code, reading the other workgroups' sums around the L2 as they were written
        // (scripts/ubench/fence_cost.hip MODE 3: no stale word in 12 800 exchanges, ~6 us for the chain on the last arriver; an
        // agent-scope fence per workgroup instead cost 94 us per round).  Nobody waits: a workgroup finishes zero to three tiles and exits.
        __shared__ long long fin_red[4];
        __shared__ int fin_do[3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                  // (also: every thread is done reading T -- it becomes the finishing's staging area)
        const bool halos = g.R < H;                       // (NI = 1: cnet1w_takes)
        const int tpi = HW >> LPXT, j = tb & (tpi - 1);   // tiles per image (a power of two), this tile's place in its image
        if (tid < 3) {
            const int d = tid - 1, jj = j + d;
            int mine = 0;
            if (jj >= 0 && jj < tpi && (d == 0 || halos)) {
                const unsigned need = halos ? 1u + (jj > 0) + (jj < tpi - 1) : 1u;
                const unsigned old = __hip_atomic_fetch_add(a.fin_cnt + tb + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == need) {
                    mine = 1;
                    __hip_atomic_store(a.fin_cnt + tb + d, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (everyone who counts on it has: ready for the next launch)
                }
            }
            fin_do[tid] = mine;
        }
        __syncthreads();
        if (fin_do[0] | fin_do[1] | fin_do[2]) {
            CfinArgs fa;
            fa.p.scratch = a.scratch; fa.p.MS = 1; fa.p.tiles = g.tiles; fa.p.R = g.R; fa.p.NI = g.NI; fa.p.lpxt = LPXT;
            fa.p.bias = a.bias; fa.p.scale = a.scale; fa.p.mode = a.mode; fa.p.Cout = a.Cout; fa.p.z = a.z_in; fa.p.z_bs = a.z_in_bs;
            fa.p.one_wave = 1; fa.p.finished = 1;
            fa.mix = a.mix; fa.z_out = a.z_out; fa.z_out_bs = a.z_out_bs; fa.acc = a.acc;
            fa.N = a.N; fa.H = H; fa.W = W; fa.HW = HW; fa.wshift = g.wshift; fa.xcd_affine = 0; fa.tape_hout = nullptr;
            const FinSrc fs = fin_src(fa.p, a.N, H, W, HW, g.wshift);
            float* fsm = reinterpret_cast<float*>(lds1 + TABS);
#pragma unroll 1
            for (int d = 0; d < 3; ++d) {
                if (!fin_do[d]) continue;
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {             // a tile = two 64-pixel chunks of the finishing kernel
                    const int chunk = 2 * (tb + d - 1) + h;
                    cfinish_chunk<64, 1, true, true>(fa, fs, chunk, chunk & ACC_EXTRA, fsm, fin_red);
                    __syncthreads();
                }
            }
        }
    }
    GH_WG_END();
}

// ------------------------------------------------------------------------------------------------ host side
static size_t cnet1w_lds_bytes(const CnetGeo& g, int hidden, int ms) {
    const size_t tabs = (((size_t)4 * hidden + g.Mpad4) * sizeof(float) + 1023) / 1024 * 1024;
    const size_t work = (size_t)3 * (hidden / ms) * 64 + (size_t)2 * g.G * 1024 + (size_t)2 * g.winplane * sizeof(_Float16);
    const size_t stage = (size_t)cnet_trow(9 * g.Cg) * 128 * sizeof(float);      // T, pixel-major
    return tabs + std::max(work, stage);
}

// instances: 1 = product, 2 = taping forward (both: C = 12 levels, the workgroup owns all 512 h2 rows), 3 = product with the h2 rows
// split over two workgroups (C = 24 levels), 4 = the backward launch of the C = 12 levels (12 channels in, 6 out)
static int cnet1w_instance(const CnetArgs& a, const CnetGeo& g, int ms) {      // 0: none
    if (a.hidden != 512 || g.ng != 1 || g.pxt != 128) return 0;
    if (a.bwd) return (ms == 1 && g.G == 18 && g.NRT4 == 2 && a.tape_h1) ? 4 : 0;
    if (ms == 1 && g.G == 10 && g.NRT4 == 4) return a.tape_h1 ? 2 : 1;
    if (ms == 2 && g.G == 18 && g.NRT4 == 7 && !a.tape_h1) return 3;
    return 0;
}

bool cnet1w_takes(const CnetArgs& a, const CnetGeo& g, int ms) {
    if (a.pre_on) return false;
    if ((a.tape_h1 || a.bwd) && (!a.tape_h1 || !a.tape_h2 || !a.mask1 || !a.mask2 || g.HW % 128 != 0)) return false;      // taping / backward: whole 32-pixel tiles of the batch per wave
    if (g.NI != 1) return false;                     // (tiles of whole small images stay on k_cnet: no level that large has them)
    const int inst = cnet1w_instance(a, g, ms);
    if (!inst) return false;
    if (cnet1w_lds_bytes(g, a.hidden, ms) > 160 * 1024) return false;
    if (g.Cg % (inst == 4 ? 2 : 4) != 0) return false;      // the tap sums take four (backward: two) output channels per read
    return true;
}

// Does the launch finish the step itself (CnetArgs::fin_cnt)?  The product instance at MS = 1, whole tiles (no tail tile whose
// arrival count would differ), a mixer of at most 48 channels staged beside the values (the staging area is T's), 64-pixel chunks
// that stay inside one image row pair.
bool cnet1w_finishes(const CnetArgs& a, const CnetGeo& g, int ms) {
    if (!a.fin_cnt || !a.z_out || a.tape_h1 || a.bwd || a.pre_on || ms != 1 || cnet1w_instance(a, g, ms) != 1) return false;
    const bool paired = a.mode == TAIL_AFFINE_FWD || a.mode == TAIL_AFFINE_REV;
    const int C = 2 * (paired ? a.Cout / 2 : a.Cout);
    if (a.mix.C != 0 && a.mix.C != C) return false;
    if (g.NI != 1 || g.HW % 128 != 0 || (g.HW & (g.HW - 1)) != 0) return false;
    const size_t stage = (size_t)cnet_trow(9 * g.Cg) * 128 * sizeof(float);
    return ((size_t)C * 64 + (size_t)C * C) * sizeof(float) <= stage;
}

int launch_cnet1w(const CnetArgs& a, const CnetGeo& g, int ms, hipStream_t s) {
    const size_t lds = cnet1w_lds_bytes(g, a.hidden, ms);
    switch (cnet1w_instance(a, g, ms)) {
    case 1:
        if (cnet1w_finishes(a, g, ms)) {
            (void)hipFuncSetAttribute((const void*)k_cnet1w<512, 10, 4, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_cnet1w<512, 10, 4, 0, 1, true>), dim3(g.tiles), dim3(256), lds, s, a, g);
            break;
        }
        (void)hipFuncSetAttribute((const void*)k_cnet1w<512, 10, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cnet1w<512, 10, 4>), dim3(g.tiles), dim3(256), lds, s, a, g);
        break;
    case 2:
        (void)hipFuncSetAttribute((const void*)k_cnet1w<512, 10, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cnet1w<512, 10, 4, 1>), dim3(g.tiles), dim3(256), lds, s, a, g);
        break;
    case 3:
        (void)hipFuncSetAttribute((const void*)k_cnet1w<512, 18, 7, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cnet1w<512, 18, 7, 0, 2>), dim3(g.tiles, 2), dim3(256), lds, s, a, g);
        break;
    case 4:
        (void)hipFuncSetAttribute((const void*)k_cnet1w<512, 18, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cnet1w<512, 18, 2, 2>), dim3(g.tiles), dim3(256), lds, s, a, g);
        break;
    default:
        set_error("cnet1w: no kernel instance");
        return GLOWHIP_EINVAL;
    }
    GH_LAUNCH_CHECK("k_cnet1w");
    return GLOWHIP_OK;
}

}  // namespace glowhip
