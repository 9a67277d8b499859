// plan.hip -- flow plans: the host-side executor that turns a FlowModel (Squeeze2d / FlowStep /
// Split2d stack, network/model.py:230-294) into a fixed sequence of kernel launches on one HIP stream.
// There is no tracing compiler: the layer list is static, so the launch sequence is built once and
// replayed; all scratch comes from a caller-provided workspace, all parameter-derived data (exp(3 logs),
// K-major MFMA weight images, W^-1, log|det W|) from a caller-provided `packed` buffer refreshed by
// glowhip_plan_pack.
#include <math.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "plan_internal.h"

namespace glowhip {

// ---------------------------------------------------------------- pack kernels
__global__ void __launch_bounds__(256) k_pack_scales(const float* __restrict__ logs, int n, float* __restrict__ scale,
                                                     float* __restrict__ inv_scale) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float l3 = logs[i] * LOGSCALE;
    scale[i] = expf(l3);
    if (inv_scale) inv_scale[i] = expf(-l3);
}

static int pack_scales(const float* logs, int n, float* scale, float* inv, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_scales, dim3(cdiv(n, 256)), dim3(256), 0, s, logs, n, scale, inv);
    GH_LAUNCH_CHECK("k_pack_scales");
    return GLOWHIP_OK;
}

// Workspace carving
struct Workspace {
    unsigned* fin_cnt;        // arrival counters of the fused finishing (sh.h CnetArgs::fin_cnt): one word per 64 pixels of the widest level
    unsigned long long* acc;
    float* bufA;
    float* bufB;
    float* h1;
    float* h2;
};

static size_t fin_cnt_words(const glowhip_plan* p, int N) {
    size_t px = 0;
    for (const LayerPlan& L : p->layers)
        if (L.d.kind == GLOWHIP_LAYER_FLOWSTEP) px = std::max(px, (size_t)L.d.H * L.d.W);
    return ((size_t)N * px + 63) / 64 + 64;
}

static size_t workspace_bytes(const glowhip_plan* p, int N) {
    size_t off = 0;
    take(off, fin_cnt_words(p, N) * 4);
    take(off, (size_t)N * 8 * (2 + ACC_EXTRA));
    take(off, (size_t)N * p->max_chw * 4);
    take(off, (size_t)N * p->max_chw * 4);
    take(off, (size_t)N * p->max_hidden * 4);
    take(off, (size_t)N * p->max_hidden * 4);
    return align_up(off, 256);
}

static int carve(const glowhip_plan* p, int N, void* ws, size_t bytes, Workspace& w) {
    if (bytes < workspace_bytes(p, N) || ws == nullptr) {
        set_error("workspace too small: need %zu bytes, got %zu", workspace_bytes(p, N), bytes);
        return GLOWHIP_EWORKSPACE;
    }
    size_t off = 0;
    w.fin_cnt = at<unsigned>(ws, take(off, fin_cnt_words(p, N) * 4));
    w.acc = at<unsigned long long>(ws, take(off, (size_t)N * 8 * (2 + ACC_EXTRA)));
    w.bufA = at<float>(ws, take(off, (size_t)N * p->max_chw * 4));
    w.bufB = at<float>(ws, take(off, (size_t)N * p->max_chw * 4));
    w.h1 = at<float>(ws, take(off, (size_t)N * p->max_hidden * 4));
    w.h2 = at<float>(ws, take(off, (size_t)N * p->max_hidden * 4));
    return GLOWHIP_OK;
}

// ---------------------------------------------------------------- coupling network f() (network/module.py:300-319)
// Runs conv3x3 -> actnorm -> relu -> conv1x1 -> actnorm -> relu -> conv3x3(zeros) and applies the
// coupling to z2.  x1: first-half channels (batch stride x1_bs).
static bool g_pack_one_stream = false;     // testing hook: glowhip_plan_pack without the side-stream fork
void plan_pack_one_stream(int on) { g_pack_one_stream = on != 0; }
static bool g_sh_disabled = false, g_sh_mix_disabled = false, g_fuse_finish_off = true, g_lu_small_off = false;
void plan_disable_sh(int off) {
    g_sh_disabled = (off & 1) != 0;        // the whole split-half path off: every coupling network on the exact-fp32 MFMA kernels
    g_sh_mix_disabled = (off & 16) != 0;   // no mixer of the next step inside the finishing kernel, no squeeze folded into a mixer
    // FUSED FINISHING (k_cnet1w finishing the step itself, cnet1w_sh.hip FIN) is OFF unless asked for: built, bit-identical to the
    // finishing kernel (tests/test_gpu_fused.py), and measured SLOWER -- 7.68 against 6.80 ms per config-B forward, + 27 us per level-1
    // launch: every workgroup arrives last at about one tile, so every CU pays the finishing's latency chain
    // (store drain, counter round trip, a read-around-L2 round trip per 64-pixel chunk) twice per launch, one workgroup at a time,
    // where the finishing kernel runs four workgroups per CU side by side.
    g_fuse_finish_off = (off & 32) == 0;
    g_lu_small_off = (off & 64) != 0;      // log|det W| of the small matrices on the workgroup-wide LU (A/B and bitwise test of the one-wave form)
}

// split-half f16 kernels off for this plan: its own family (glowhip_plan_set_family) or the process-wide testing hook
static bool sh_off(const glowhip_plan* p) { return g_sh_disabled || (p && p->family == GLOWHIP_FAMILY_EXACT_FP32); }
static bool cnet_runs(const glowhip_plan* p, const LayerPlan& L) { return L.cnet && !sh_off(p); }

// deep levels (dnet_sh.hip): layers k_cnet does not take, with an invertible 1x1 convolution, on the product family
static bool dnet_runs(const glowhip_plan* p, const LayerPlan& L) { return L.dnet && !sh_off(p); }
// the run of consecutive FlowSteps of li's shape that take the deep-level kernels, walking in direction `dir` (+1 encode, -1 decode)
static int dnet_run_end(const glowhip_plan* p, int li, int dir) {
    const int nl = (int)p->layers.size();
    const glowhip_layer_desc& d = p->layers[li].d;
    int lj = li;
    while (lj + dir >= 0 && lj + dir < nl) {
        const LayerPlan& Ln = p->layers[lj + dir];
        if (Ln.d.kind != GLOWHIP_LAYER_FLOWSTEP || !dnet_runs(p, Ln) || Ln.d.C != d.C || Ln.d.H != d.H || Ln.d.W != d.W ||
            Ln.d.hidden != d.hidden || Ln.Cout != p->layers[li].Cout) break;
        lj += dir;
    }
    return lj;
}
static int tail_mode(const glowhip_layer_desc& d, int reverse) {
    return d.coupling == GLOWHIP_COUPLING_AFFINE ? (reverse ? TAIL_AFFINE_REV : TAIL_AFFINE_FWD) : (reverse ? TAIL_ADD_REV : TAIL_ADD_FWD);
}
// One level's run of FlowSteps [li .. lj] (encode) on the deep-level kernels: PREP, then MIX -> F0 -> F2 -> F4 -> FIN per step; the
// result is left in `S` (N, C, H, W).
static int run_dnet_forward(glowhip_plan* p, const void* packed, int li, int lj, const float* cur, float* S, int N, const Workspace& w,
                            hipStream_t s) {
    const LayerPlan& L0 = p->layers[li];
    const glowhip_layer_desc& d = L0.d;
    const long chw = (long)d.C * d.H * d.W;
    DnetLevel D{N, d.C, d.H, d.W, d.hidden, L0.Cout, S, chw, w.h1};
    GH_TRY(dnet_level_begin(D, s));
    p->cur_layer = li;
    {
        ScopedTimer t(p, GLOWHIP_K_OTHER, 0, s);
        count_launch(p, "k_dn_prep");
        GH_TRY(dnet_prep(D, cur, chw, d.an_bias, at<float>(packed, L0.an_scale), 0, s));
    }
    for (int k = li; k <= lj; ++k) {
        const LayerPlan& L = p->layers[k];
        p->cur_layer = k;
        {
            ScopedTimer t(p, GLOWHIP_K_CHANMIX, 1, s);
            count_launch(p, "k_dn_gemm(mix)");
            GH_TRY(dnet_mix(D, at<char>(packed, L.dn_mix), nullptr, nullptr, 1, s));
        }
        int ks = 1;
        {
            ScopedTimer t(p, GLOWHIP_K_CONV_F2, 1, s);
            count_launch(p, "k_dn_gemm(f0,f2,f4)");
            GH_TRY(dnet_coupling_net(D, at<char>(packed, L.dn_w0), at<char>(packed, L.dn_w2), at<char>(packed, L.dn_w4), &ks, s));
        }
        const LayerPlan* Ln = k < lj ? &p->layers[k + 1] : nullptr;
        ScopedTimer t(p, GLOWHIP_K_CFINISH, 0, s);
        count_launch(p, "k_dn_fin");
        GH_TRY(dnet_finish(D, ks, L.d.f4_bias, at<float>(packed, L.f4_scale), tail_mode(L.d, 0), w.acc, Ln ? Ln->d.an_bias : nullptr,
                           Ln ? at<float>(packed, Ln->an_scale) : nullptr, Ln != nullptr, s));
    }
    return GLOWHIP_OK;
}
// ... and decode: steps li, li - 1, ..., lj (li >= lj): PREP, then F0 -> F2 -> F4 -> FIN -> MIX^-1 per step
static int run_dnet_reverse(glowhip_plan* p, const void* packed, int li, int lj, const float* cur, float* S, int N, const Workspace& w,
                            hipStream_t s) {
    const LayerPlan& L0 = p->layers[li];
    const glowhip_layer_desc& d = L0.d;
    const long chw = (long)d.C * d.H * d.W;
    DnetLevel D{N, d.C, d.H, d.W, d.hidden, L0.Cout, S, chw, w.h1};
    GH_TRY(dnet_level_begin(D, s));
    p->cur_layer = li;
    {
        ScopedTimer t(p, GLOWHIP_K_OTHER, 0, s);
        count_launch(p, "k_dn_prep");
        GH_TRY(dnet_prep(D, cur, chw, nullptr, nullptr, 1, s));
    }
    for (int k = li; k >= lj; --k) {
        const LayerPlan& L = p->layers[k];
        p->cur_layer = k;
        int ks = 1;
        {
            ScopedTimer t(p, GLOWHIP_K_CONV_F2, 1, s);
            count_launch(p, "k_dn_gemm(f0,f2,f4)");
            GH_TRY(dnet_coupling_net(D, at<char>(packed, L.dn_w0), at<char>(packed, L.dn_w2), at<char>(packed, L.dn_w4), &ks, s));
        }
        {
            ScopedTimer t(p, GLOWHIP_K_CFINISH, 0, s);
            count_launch(p, "k_dn_fin");
            GH_TRY(dnet_finish(D, ks, L.d.f4_bias, at<float>(packed, L.f4_scale), tail_mode(L.d, 1), w.acc, nullptr, nullptr, 1, s));
        }
        ScopedTimer t(p, GLOWHIP_K_CHANMIX, 1, s);
        count_launch(p, "k_dn_gemm(mix)");
        GH_TRY(dnet_mix(D, at<char>(packed, L.dn_mixinv), at<float>(packed, L.an_inv_scale), L.d.an_bias, k > lj, s));
    }
    return GLOWHIP_OK;
}

// ---- the one-kernel coupling network (cnet_sh.hip).  A FlowStep is k_cnet (partial sums of h = f(z1)) + a finishing step
// (coupling, log-det, channel mixer); the finishing step of step k runs either as its own kernel or inside step k+1's k_cnet
// while that builds its window ("pending").
static CnetArgs cnet_base(const LayerPlan& L, const void* packed, int N, int reverse, float* scratch, const Workspace& w) {
    const glowhip_layer_desc& d = L.d;
    CnetArgs c{};
    c.w0 = at<char>(packed, L.cn_w0); c.w2 = at<char>(packed, L.cn_w2); c.w4 = at<char>(packed, L.cn_w4);
    c.N = N; c.Cin = d.C / 2; c.H = d.H; c.W = d.W; c.hidden = d.hidden; c.Cout = L.Cout;
    c.scratch = scratch;
    c.bias = d.f4_bias; c.scale = at<float>(packed, L.f4_scale);
    c.mode = d.coupling == GLOWHIP_COUPLING_AFFINE ? (reverse ? TAIL_AFFINE_REV : TAIL_AFFINE_FWD)
                                                   : (reverse ? TAIL_ADD_REV : TAIL_ADD_FWD);
    c.acc = w.acc;
    return c;
}
// which instance of k_cnet ran (run-time evidence for the tests): "variant:k_cnet<hidden,row split,pixel tile>"
static void count_cnet_variant(glowhip_plan* p, const CnetArgs& c, const CnetPending& pend) {
    char name[64];
    snprintf(name, sizeof name, "variant:k_cnet%s<%d,%d,%d>", pend.one_wave ? "1w" : "", c.hidden, pend.MS, 1 << pend.lpxt);
    count_launch(p, name);
}
static CnetMixer mixer_fwd(const LayerPlan& L, const void* packed) {      // ActNorm + permutation of step L, forward
    const glowhip_layer_desc& d = L.d;
    return CnetMixer{d.C, 0, d.an_bias, at<float>(packed, L.an_scale), d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr,
                     d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx : nullptr};
}
static CnetMixer mixer_rev(const LayerPlan& L, const void* packed) {      // permutation^-1 + ActNorm^-1 of step L
    const glowhip_layer_desc& d = L.d;
    return CnetMixer{d.C, 1, d.an_bias, at<float>(packed, L.an_inv_scale),
                     d.permutation == GLOWHIP_PERM_INVCONV ? at<float>(packed, L.winv) : nullptr,
                     d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx_inv : nullptr};
}
// may the finishing step of layer A run inside layer B's k_cnet?  (Off by default: the window-time finishing costs a workgroup
// ~25 k cycles of dependent global round trips with 8 waves, more than the ~10 us finishing kernel it saves; kept behind
// glowhip_debug_force_tail_tile(0x8000000) with its bitwise-equality test.)
static bool cnet_chain(const glowhip_plan* p, const LayerPlan& A, const LayerPlan& B) {
    const glowhip_layer_desc& a = A.d; const glowhip_layer_desc& b = B.d;
    return cnet_chain_enabled() && !g_sh_mix_disabled && a.kind == GLOWHIP_LAYER_FLOWSTEP && b.kind == GLOWHIP_LAYER_FLOWSTEP &&
           cnet_runs(p, A) && cnet_runs(p, B) && a.C == b.C && a.H == b.H && a.W == b.W && a.coupling == b.coupling &&
           cnet_pre_supported(b.C / 2, b.H, b.W, b.hidden, B.Cout, b.C);
}

// The coupling network of a FlowStep that neither k_cnet nor the deep-level kernels take (odd hidden widths, tiny images, the
// exact-fp32 family): one exact-fp32 kernel per layer.
static int run_coupling(glowhip_plan* P, const LayerPlan& L, const void* packed, const float* x1, long x1_bs, const float* z2_in,
                        long z2_in_bs, float* z2_out, long z2_out_bs, int N, int reverse, const Workspace& w,
                        hipStream_t s) {
    const glowhip_layer_desc& d = L.d;
    const int Ch = d.C / 2, HW = d.H * d.W, hid = d.hidden;
    {   // f.0: 3x3, Cin=C/2 -> hidden, ActNorm + ReLU epilogue
        ScopedTimer t0(P, GLOWHIP_K_CONV_F0, L.mfma_first || L.first_halo, s);
        if (L.first_halo) {
            const float* wf = at<float>(packed, L.f0_wt);
            count_launch(P, "k_conv_first_f32");
            GH_TRY(launch_conv_mfma_first(x1, x1_bs, wf, wf + (size_t)9 * Ch * hid, w.h1, N, Ch, d.H, d.W, hid, s, 1));
        } else if (L.mfma_first) {
            count_launch(P, "k_conv_wide_f32");
            GH_TRY(launch_conv_mfma_wide(x1, x1_bs, at<float>(packed, L.f0_wt), d.f0_an_bias, at<float>(packed, L.f0_scale),
                                         w.h1, N, Ch, d.H, d.W, hid, 3, s, 1, w.h2, (size_t)N * P->max_hidden));
        } else {
            ConvArgs c{x1, x1_bs, d.f0_w, nullptr, d.f0_an_bias, nullptr, at<float>(packed, L.f0_scale), 1, w.h1,
                       N, Ch, d.H, d.W, hid, 3};
            count_launch(P, "k_conv_direct");
            GH_TRY(launch_conv_direct(c, s));
        }
    }
    {   // f.2: 1x1, hidden -> hidden, ActNorm + ReLU epilogue
        ScopedTimer t2(P, GLOWHIP_K_CONV_F2, L.mfma_mid, s);
        if (L.mfma_mid) {
            count_launch(P, "k_gemm_f32");
            GH_TRY(launch_conv_mfma_wide(w.h1, (long)hid * HW, at<float>(packed, L.f2_wt), d.f2_an_bias,
                                         at<float>(packed, L.f2_scale), w.h2, N, hid, d.H, d.W, hid, 1, s));
        } else {
            ConvArgs c{w.h1, (long)hid * HW, d.f2_w, nullptr, d.f2_an_bias, nullptr, at<float>(packed, L.f2_scale), 1, w.h2,
                       N, hid, d.H, d.W, hid, 1};
            count_launch(P, "k_conv_direct");
            GH_TRY(launch_conv_direct(c, s));
        }
    }
    // f.4: 3x3 zeros conv (+bias, *exp(3 logs)) and the coupling itself
    ScopedTimer t4(P, GLOWHIP_K_CONV_F4, L.mfma_last, s);
    if (L.mfma_last) {
        TailConvArgs t{};
        t.x = w.h2; t.x_bs = (long)hid * HW; t.wp = at<float>(packed, L.f4_wp); t.bias = d.f4_bias;
        t.scale = at<float>(packed, L.f4_scale);
        t.N = N; t.Cin = hid; t.H = d.H; t.W = d.W; t.Cout = L.Cout;
        t.mode = d.coupling == GLOWHIP_COUPLING_AFFINE ? (reverse ? TAIL_AFFINE_REV : TAIL_AFFINE_FWD)
                                                       : (reverse ? TAIL_ADD_REV : TAIL_ADD_FWD);
        t.z2_in = z2_in; t.z2_in_bs = z2_in_bs; t.z2_out = z2_out; t.z2_out_bs = z2_out_bs; t.acc = w.acc;
        t.zeros = at<float>(packed, 64);   // zero block kept by glowhip_plan_pack
        count_launch(P, "k_conv_tail_f32");
        GH_TRY(launch_conv_mfma_tail(t, s));
    } else {
        if (L.wide_last) {     // (conv + bias) * exp(3 logs) on the fp32 MFMA implicit-GEMM kernel
            count_launch(P, "k_conv_wide_f32");
            // split-K partial sums go behind the output in the same (level-1 sized) buffer
            const size_t out_f = (size_t)N * L.Cout * HW, h1_f = (size_t)N * P->max_hidden;
            GH_TRY(launch_conv_mfma_wide(w.h2, (long)hid * HW, at<float>(packed, L.f4_wt), d.f4_bias, at<float>(packed, L.f4_scale),
                                         w.h1, N, hid, d.H, d.W, L.Cout, 3, s, 0, h1_f > out_f ? w.h1 + out_f : nullptr,
                                         h1_f > out_f ? h1_f - out_f : 0));
        } else {
            ConvArgs c{w.h2, (long)hid * HW, d.f4_w, d.f4_bias, nullptr, nullptr, at<float>(packed, L.f4_scale), 0, w.h1,
                       N, hid, d.H, d.W, L.Cout, 3};
            count_launch(P, "k_conv_direct");
            GH_TRY(launch_conv_direct(c, s));
        }
        CouplingTailArgs t{w.h1, z2_in, z2_in_bs, z2_out, z2_out_bs, N, Ch, HW,
                           d.coupling == GLOWHIP_COUPLING_AFFINE, reverse, w.acc};
        GH_TRY(launch_coupling_tail(t, s));
    }
    return GLOWHIP_OK;
}

// Split2d prior conv + tail (network/module.py:498-536).  d.C = channels of the un-split tensor.
static int run_split(const LayerPlan& L, const void* packed, const float* z1, long z1_bs, const float* z2, long z2_bs,
                     const float* eps, float* z2_out, long z2_out_bs, int N, int reverse, const Workspace& w,
                     hipStream_t s) {
    const glowhip_layer_desc& d = L.d;
    const int Ch = d.C / 2, HW = d.H * d.W;
    if (L.mfma_last) {
        TailConvArgs t{};
        t.x = z1; t.x_bs = z1_bs; t.wp = at<float>(packed, L.f4_wp); t.bias = d.f4_bias;
        t.scale = at<float>(packed, L.f4_scale);
        t.N = N; t.Cin = Ch; t.H = d.H; t.W = d.W; t.Cout = L.Cout;
        t.mode = reverse ? TAIL_SPLIT_REV : TAIL_SPLIT_FWD;
        t.z2_in = reverse ? eps : z2; t.z2_in_bs = reverse ? (long)Ch * HW : z2_bs;
        t.z2_out = z2_out; t.z2_out_bs = z2_out_bs; t.acc = w.acc;
        t.zeros = at<float>(packed, 64);
        GH_TRY(launch_conv_mfma_tail(t, s));
    } else {
        ConvArgs c{z1, z1_bs, d.f4_w, d.f4_bias, nullptr, nullptr, at<float>(packed, L.f4_scale), 0, w.h1,
                   N, Ch, d.H, d.W, L.Cout, 3};
        GH_TRY(launch_conv_direct(c, s));
        SplitTailArgs t{w.h1, z2, z2_bs, eps, z2_out, z2_out_bs, N, Ch, HW, reverse, w.acc};
        GH_TRY(launch_split_tail(t, s));
    }
    return GLOWHIP_OK;
}

static float* other_buf(const Workspace& w, const float* cur) { return cur == w.bufA ? w.bufB : w.bufA; }

// ---------------------------------------------------------------- encode (network/model.py:263-276)
// A Squeeze2d layer whose output only the NEXT layer's k_chanmix reads (the first FlowStep of a level on the k_cnet path or the
// kernel pairs) is not run at all: the mixer gathers the squeezed view itself, dequantisation noise and 8-bit scaling included
// ("fuse uint8 -> fp32 scaling + noise + first squeeze into the first kernel").
struct SqueezeFold { const void* src; int u8; float div; const float* noise; RngSpec rng; int W; };
// (testing: the mixer-fusion switch glowhip_debug_force_tail_tile(0x8000) also brings the squeeze kernels back -- the two must
// agree bit for bit)

static int run_forward(glowhip_plan* p, const void* packed, const float* x, const float* noise, float* z_out, int N,
                       const Workspace& w, hipStream_t s, int first_layer = 0, const RngSpec* rng = nullptr,
                       const uint8_t* x_u8 = nullptr, float u8_div = 1.f) {
    const float* cur = x;
    SqueezeFold fold{};      // set by a skipped squeeze layer, consumed by the next layer's k_chanmix
    const int nl = (int)p->layers.size();
    bool premixed = false;   // `cur` already holds this step's ActNorm + permutation output (applied by the previous tail)
    bool pending = false;    // the previous FlowStep's k_cnet has run, its finishing step has not (cnet_chain)
    CnetPending pend{};
    int scr_i = 0;
    for (int li = first_layer; li < nl; ++li) {
        const LayerPlan& L = p->layers[li];
        const glowhip_layer_desc& d = L.d;
        p->cur_layer = li;
        float* dst = (li == nl - 1) ? z_out : other_buf(w, cur);
        const int HW = d.H * d.W;
        const long chw = (long)d.C * HW;
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) {
            const bool u8 = x_u8 != nullptr && li == 0;
            bool foldable = false;
            if (!g_sh_mix_disabled && li + 1 < nl - 1 && d.W % 2 == 0) {
                const LayerPlan& Ln = p->layers[li + 1];
                foldable = Ln.d.kind == GLOWHIP_LAYER_FLOWSTEP && !dnet_runs(p, Ln) && chanmix_squeeze_foldable(Ln.d.C);
            }
            if (foldable) {
                fold = SqueezeFold{u8 ? (const void*)x_u8 : (const void*)cur, u8 ? 1 : 0, u8_div, noise,
                                   (li == 0 && rng) ? *rng : RngSpec{}, d.W};
                count_launch(p, "squeeze(folded)");
                noise = nullptr; rng = nullptr;
                continue;      // `cur` stays the un-squeezed tensor; the next layer reads it through `fold`
            }
            if (u8) GH_TRY(launch_squeeze_u8(x_u8, noise, dst, N, d.C, d.H, d.W, 2, u8_div, s, rng));
            else GH_TRY(launch_squeeze(cur, noise, dst, N, d.C, d.H, d.W, 2, 0, s, li == 0 ? rng : nullptr));
            noise = nullptr; rng = nullptr;
        } else {
            if (noise || (rng && li == 0)) {  // dequantisation noise with no leading squeeze (only possible at layer 0, cur == x):
                          // the identity "squeeze" (factor 1) adds it into a workspace buffer
                GH_TRY(launch_squeeze(cur, noise, w.bufA, N, d.C, d.H, d.W, 1, 0, s, rng));
                cur = w.bufA;
                noise = nullptr; rng = nullptr;
                if (li != nl - 1) dst = w.bufB;
            }
            const int Ch = d.C / 2;
            if (d.kind == GLOWHIP_LAYER_FLOWSTEP && dnet_runs(p, L) && !premixed && !pending) {
                // ---- deep level: the whole run of same-shape FlowSteps on the per-layer SH2 kernels (dnet_sh.hip)
                const int lj = dnet_run_end(p, li, +1);
                float* S = other_buf(w, cur);
                GH_TRY(run_dnet_forward(p, packed, li, lj, cur, S, N, w, s));
                if (lj == nl - 1) GH_TRY(launch_copy_strided(S, chw, z_out, chw, N, chw, s));
                cur = S;
                li = lj;
                continue;
            }
            if (d.kind == GLOWHIP_LAYER_FLOWSTEP && cnet_runs(p, L)) {
                // ---- k_cnet path.  `cur` holds the input of this step's mixer, or -- `premixed` -- its output, or -- `pending` --
                // the state the PREVIOUS step's k_cnet read, whose finishing (coupling + this step's mixer) this launch does itself
                float* scratch = (scr_i ^= 1) ? w.h1 : w.h2;
                CnetArgs c = cnet_base(L, packed, N, 0, scratch, w);
                if (pending) {
                    float* nxt = other_buf(w, cur);
                    c.pre_on = 1; c.pre = pend; c.pre_mix = mixer_fwd(L, packed); c.pre_z_new = nxt; c.pre_z_new_bs = chw;
                    cur = nxt;
                } else {
                    if (!premixed) {
                        ChanMixArgs m{};
                        m.in_a = cur; m.in_a_bs = chw; m.in_b = cur + (long)Ch * HW; m.in_b_bs = chw; m.Ca = Ch;
                        m.out = dst; m.out_bs = chw;
                        m.bias = d.an_bias; m.scale = at<float>(packed, L.an_scale);
                        m.matrix = d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr;
                        m.gather = d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx : nullptr;
                        m.reverse = 0; m.N = N; m.C = d.C; m.HW = HW;
                        if (fold.src) { m.sq_src = fold.src; m.sq_u8 = fold.u8; m.sq_div = fold.div; m.sq_noise = fold.noise; m.sq_rng = fold.rng; m.sq_W = fold.W; fold = SqueezeFold{}; }
                        ScopedTimer tm(p, GLOWHIP_K_CHANMIX, 0, s);
                        count_launch(p, "k_chanmix");
                        GH_TRY(launch_chanmix(m, s));
                        cur = dst;
                    }
                    c.x = cur; c.x_bs = chw; c.z_in = cur; c.z_in_bs = chw;
                }
                premixed = false;
                const bool chain = li + 1 < nl && cnet_chain(p, L, p->layers[li + 1]);
                float* out = (li == nl - 1) ? z_out : const_cast<float*>(cur);
                if (!chain) {   // the step is finished right behind its k_cnet: with the NEXT step's mixer when that is a same-shape FlowStep
                    c.z_out = out; c.z_out_bs = chw;
                    if (li + 1 < nl - 1 && !g_sh_mix_disabled && d.C <= 96) {
                        const LayerPlan& Ln = p->layers[li + 1];
                        const glowhip_layer_desc& dn = Ln.d;
                        if (dn.kind == GLOWHIP_LAYER_FLOWSTEP && dn.C == d.C && dn.H == d.H && dn.W == d.W) {
                            c.mix = mixer_fwd(Ln, packed);
                            premixed = true;
                        }
                    }
                    // ... by the launch itself where it can (fused finishing, sh.h CnetArgs::fin_cnt; per-launch event timing keeps the two apart)
                    if (!g_fuse_finish_off && !p->timing) c.fin_cnt = w.fin_cnt;
                }
                {
                    ScopedTimer t(p, GLOWHIP_K_CNET, 1, s);
                    count_launch(p, pending ? "k_cnet+prev_finish" : "k_cnet");
                    GH_TRY(launch_cnet_main(c, s, &pend));
                    count_cnet_variant(p, c, pend);
                }
                pending = chain;
                if (!chain) {
                    if (pend.finished) {
                        count_launch(p, "k_cnet(finishes the step)");
                    } else {
                        ScopedTimer t(p, GLOWHIP_K_CFINISH, 0, s);
                        count_launch(p, c.mix.C ? "k_cfinish+mixer" : "k_cfinish");
                        GH_TRY(launch_cnet_finish(c, pend, s));
                    }
                    cur = out;
                }
                continue;
            }
            if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
                if (premixed) {
                    dst = const_cast<float*>(cur);   // the step runs in place on the already mixed buffer
                } else {
                    ChanMixArgs m{};
                    m.in_a = cur; m.in_a_bs = chw; m.in_b = cur + (long)Ch * HW; m.in_b_bs = chw; m.Ca = Ch;
                    m.out = dst; m.out_bs = chw;
                    m.bias = d.an_bias; m.scale = at<float>(packed, L.an_scale);
                    m.matrix = d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr;
                    m.gather = d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx : nullptr;
                    m.reverse = 0; m.N = N; m.C = d.C; m.HW = HW;
                    if (fold.src) { m.sq_src = fold.src; m.sq_u8 = fold.u8; m.sq_div = fold.div; m.sq_noise = fold.noise; m.sq_rng = fold.rng; m.sq_W = fold.W; fold = SqueezeFold{}; }
                    ScopedTimer tm(p, GLOWHIP_K_CHANMIX, 0, s);
                    count_launch(p, "k_chanmix");
                    GH_TRY(launch_chanmix(m, s));
                }
                float* z2 = dst + (long)Ch * HW;
                GH_TRY(join_legacy(p, s));
                GH_TRY(run_coupling(p, L, packed, dst, chw, z2, chw, z2, chw, N, 0, w, s));
                premixed = false;
            } else {  // SPLIT2D: score z2 under the prior predicted from z1, keep z1
                GH_TRY(join_legacy(p, s));
                GH_TRY(run_split(L, packed, cur, chw, cur + (long)Ch * HW, chw, nullptr, nullptr, 0, N, 0, w, s));
                GH_TRY(launch_copy_strided(cur, chw, dst, (long)Ch * HW, N, (long)Ch * HW, s));
            }
        }
        cur = dst;
    }
    return GLOWHIP_OK;
}

// ---------------------------------------------------------------- decode (network/model.py:278-294)
static int run_reverse(glowhip_plan* p, const void* packed, const float* z, const float* const* eps, int n_eps,
                       float* x_out, int N, const Workspace& w, hipStream_t s) {
    const float* cur = z;
    const int nl = (int)p->layers.size();
    int ke = 0;
    bool pending = false;    // as in run_forward
    CnetPending pend{};
    int scr_i = 0;
    for (int li = nl - 1; li >= 0; --li) {
        const LayerPlan& L = p->layers[li];
        const glowhip_layer_desc& d = L.d;
        p->cur_layer = li;
        float* dst = (li == 0) ? x_out : other_buf(w, cur);
        const int HW = d.H * d.W;
        const long chw = (long)d.C * HW;
        const int Ch = d.C / 2;
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) {
            GH_TRY(launch_squeeze(cur, nullptr, dst, N, d.C * 4, d.H / 2, d.W / 2, 2, 1, s));
        } else if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
            float* z2 = dst + (long)Ch * HW;
            if (dnet_runs(p, L) && !pending) {
                const int lj = dnet_run_end(p, li, -1);
                float* S = other_buf(w, cur);
                GH_TRY(run_dnet_reverse(p, packed, li, lj, cur, S, N, w, s));
                if (lj == 0) GH_TRY(launch_copy_strided(S, chw, x_out, chw, N, chw, s));
                cur = S;
                li = lj;
                continue;
            }
            if (cnet_runs(p, L)) {
                // coupling^-1, permutation^-1 and ActNorm^-1 by the finishing step -- run by the next-executed step's k_cnet where
                // the two chain, by the finishing kernel otherwise
                float* scratch = (scr_i ^= 1) ? w.h1 : w.h2;
                CnetArgs c = cnet_base(L, packed, N, 1, scratch, w);
                if (pending) {
                    float* nxt = other_buf(w, cur);
                    c.pre_on = 1; c.pre = pend; c.pre_mix = mixer_rev(p->layers[li + 1], packed); c.pre_z_new = nxt; c.pre_z_new_bs = chw;
                    cur = nxt;
                } else {
                    c.x = cur; c.x_bs = chw; c.z_in = cur; c.z_in_bs = chw;
                }
                const bool chain = d.C <= 96 && li - 1 >= 0 && cnet_chain(p, L, p->layers[li - 1]) && p->layers[li - 1].d.C <= 96;
                float* fout = nullptr;
                if (!chain && d.C <= 96) {      // (as in run_forward: the finishing's arguments are there at launch time, the launch may do it)
                    fout = (li == 0) ? x_out : other_buf(w, cur);
                    c.z_out = fout; c.z_out_bs = chw; c.mix = mixer_rev(L, packed);
                    if (!g_fuse_finish_off && !p->timing) c.fin_cnt = w.fin_cnt;
                }
                {
                    ScopedTimer t(p, GLOWHIP_K_CNET, 1, s);
                    count_launch(p, pending ? "k_cnet+prev_finish" : "k_cnet");
                    GH_TRY(launch_cnet_main(c, s, &pend));
                    count_cnet_variant(p, c, pend);
                }
                pending = chain;
                if (!chain && d.C <= 96) {
                    float* out = fout;
                    if (pend.finished) {
                        count_launch(p, "k_cnet(finishes the step)");
                    } else {
                        ScopedTimer t(p, GLOWHIP_K_CFINISH, 0, s);
                        count_launch(p, "k_cfinish+mixer");
                        GH_TRY(launch_cnet_finish(c, pend, s));
                    }
                    cur = out;
                } else if (!chain) {
                    // wider than the finishing kernel's mixer (additive coupling, 96 < C <= 112: L.cnet holds, the fused mixer
                    // does not): coupling^-1 alone by the finishing kernel, then permutation^-1 + ActNorm^-1 by k_chanmix
                    float* mid = other_buf(w, cur);
                    c.z_out = mid; c.z_out_bs = chw;
                    {
                        ScopedTimer t(p, GLOWHIP_K_CFINISH, 0, s);
                        count_launch(p, "k_cfinish");
                        GH_TRY(launch_cnet_finish(c, pend, s));
                    }
                    float* out = (li == 0) ? x_out : other_buf(w, mid);
                    ChanMixArgs m{};
                    m.in_a = mid; m.in_a_bs = chw; m.in_b = mid + (long)Ch * HW; m.in_b_bs = chw; m.Ca = Ch;
                    m.out = out; m.out_bs = chw;
                    m.bias = d.an_bias; m.scale = at<float>(packed, L.an_inv_scale);
                    m.matrix = d.permutation == GLOWHIP_PERM_INVCONV ? at<float>(packed, L.winv) : nullptr;
                    m.gather = d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx_inv : nullptr;
                    m.reverse = 1; m.N = N; m.C = d.C; m.HW = HW;
                    ScopedTimer tm(p, GLOWHIP_K_CHANMIX, 0, s);
                    count_launch(p, "k_chanmix");
                    GH_TRY(launch_chanmix(m, s));
                    cur = out;
                }
                continue;
            }
            // z2 is updated IN PLACE when `cur` is a workspace buffer (every step but the first of a decode, whose input is the
            // caller's z): the mixer's inputs then do not alias its output, and the wide mixer may slice its output channels
            // over workgroups (config E's 4x4 level: 16 workgroups of 384 outputs each took 350 us per step with the alias)
            if (cur == w.bufA || cur == w.bufB) z2 = const_cast<float*>(cur) + (long)Ch * HW;
            GH_TRY(run_coupling(p, L, packed, cur, chw, cur + (long)Ch * HW, chw, z2, chw, N, 1, w, s));
            ChanMixArgs m{};
            m.in_a = cur; m.in_a_bs = chw; m.in_b = z2; m.in_b_bs = chw; m.Ca = Ch;
            m.out = dst; m.out_bs = chw;
            m.bias = d.an_bias; m.scale = at<float>(packed, L.an_inv_scale);
            m.matrix = d.permutation == GLOWHIP_PERM_INVCONV ? at<float>(packed, L.winv) : nullptr;
            m.gather = d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx_inv : nullptr;
            m.reverse = 1; m.N = N; m.C = d.C; m.HW = HW;
            ScopedTimer tm(p, GLOWHIP_K_CHANMIX, 0, s);
            count_launch(p, "k_chanmix");
            GH_TRY(launch_chanmix(m, s));
        } else {  // SPLIT2D reverse: z1 = cur (N, C/2, HW) -> cat(z1, mean + exp(logs)*eps)
            GH_REQUIRE(ke < n_eps && eps && eps[ke], "decode: missing eps draw for Split2d #%d", ke);
            GH_TRY(run_split(L, packed, cur, (long)Ch * HW, nullptr, 0, eps[ke], dst + (long)Ch * HW, chw, N, 1, w, s));
            GH_TRY(launch_copy_strided(cur, (long)Ch * HW, dst, chw, N, (long)Ch * HW, s));
            ++ke;
        }
        cur = dst;
    }
    return GLOWHIP_OK;
}

static int check_plan_args(const glowhip_plan* plan, const void* packed, int N) {
    GH_REQUIRE(plan != nullptr, "null plan");
    GH_REQUIRE(packed != nullptr, "null packed-parameter buffer (call glowhip_plan_pack first)");
    GH_REQUIRE(N >= 0 && N <= 65535, "batch size %d out of range", N);
    return GLOWHIP_OK;
}

}  // namespace glowhip

// ================================================================================================ C ABI
extern "C" {

glowhip_plan* glowhip_plan_create(const glowhip_layer_desc* layers, int n_layers) {
    if (!layers || n_layers <= 0) {
        set_error("plan_create: empty layer list");
        return nullptr;
    }
    glowhip_plan* p = new glowhip_plan();
    size_t off = 0;
    take(off, 256);  // offset 0: plan-wide data-independent log-det total (fp64); offset 64..127: zero block for LDS-DMA padding
    int C = layers[0].C, H = layers[0].H, W = layers[0].W;
    p->in_shape[0] = C; p->in_shape[1] = H; p->in_shape[2] = W;
    for (int i = 0; i < n_layers; ++i) {
        LayerPlan L;
        L.d = layers[i];
        const glowhip_layer_desc& d = L.d;
        auto fail = [&](const char* why) {
            set_error("plan_create: layer %d: %s (kind=%d C=%d H=%d W=%d)", i, why, d.kind, d.C, d.H, d.W);
            delete p;
            return (glowhip_plan*)nullptr;
        };
        if (d.C != C || d.H != H || d.W != W) return fail("input shape does not chain from the previous layer");
        if (d.C <= 0 || d.H <= 0 || d.W <= 0) return fail("empty shape");
        p->max_chw = std::max(p->max_chw, (long)C * H * W);
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) {
            if (H % 2 || W % 2) return fail("squeeze needs even H and W");
            C *= 4; H /= 2; W /= 2;
        } else if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
            if (C % 2) return fail("FlowStep needs an even channel count");  // network/model.py:169
            if (d.hidden <= 0) return fail("hidden_channels must be positive");
            if (!d.an_bias || !d.an_logs || !d.f0_w || !d.f0_an_bias || !d.f0_an_logs || !d.f2_w || !d.f2_an_bias ||
                !d.f2_an_logs || !d.f4_w || !d.f4_bias || !d.f4_logs)
                return fail("missing parameter pointer");
            if (d.permutation == GLOWHIP_PERM_INVCONV) {
                if (!d.invconv_w) return fail("missing invconv weight");
            } else if (d.permutation == GLOWHIP_PERM_GATHER) {
                if (!d.perm_idx || !d.perm_idx_inv) return fail("missing permutation tables");
            } else return fail("unknown permutation");
            if (d.coupling != GLOWHIP_COUPLING_ADDITIVE && d.coupling != GLOWHIP_COUPLING_AFFINE)
                return fail("unknown coupling");
            L.Cout = d.coupling == GLOWHIP_COUPLING_AFFINE ? C : C / 2;
            L.an_scale = take(off, (size_t)C * 4);
            L.an_inv_scale = take(off, (size_t)C * 4);
            L.winv = take(off, (size_t)C * C * 4);
            L.logabsdet = take(off, 4);
            L.konst = take(off, 8);
            L.lu_scratch = take(off, invconv_scratch_bytes(C));
            L.f0_scale = take(off, (size_t)d.hidden * 4);
            L.f2_scale = take(off, (size_t)d.hidden * 4);
            L.f4_scale = take(off, (size_t)L.Cout * 4);
            L.mfma_first = conv_mfma_wide_supported(C / 2, H, W, d.hidden, 3);
            L.mfma_mid = conv_mfma_wide_supported(d.hidden, H, W, d.hidden, 1);
            L.mfma_last = conv_mfma_tail_supported(d.hidden, H, W, L.Cout);
            L.first_halo = conv_mfma_first_supported(C / 2, H, W, d.hidden);
            if (L.first_halo) L.f0_wt = take(off, conv_mfma_first_packed_bytes(C / 2, d.hidden));
            else if (L.mfma_first) L.f0_wt = take(off, conv_mfma_wide_packed_bytes(C / 2, d.hidden, 3));
            if (L.mfma_mid) L.f2_wt = take(off, conv_mfma_wide_packed_bytes(d.hidden, d.hidden, 1));
            if (L.first_halo && L.mfma_first) L.f0_init = take(off, conv_mfma_wide_packed_bytes(C / 2, d.hidden, 3));
            else if (L.mfma_first) L.f0_init = L.f0_wt;      // already the plain K-major image
            L.cnet = cnet_supported(C / 2, H, W, d.hidden, L.Cout);
            if (L.cnet) {
                L.cn_w0 = take(off, sh2_image_bytes(cnet_g0(C / 2) * 8, d.hidden));
                L.cn_w2 = take(off, sh2_image_bytes(d.hidden, d.hidden));
                L.cn_w4 = take(off, cnet_w4_bytes(d.hidden, L.Cout));
                p->max_hidden = std::max(p->max_hidden, (long)cnet_scratch_floats_per_sample(H, W, L.Cout));
                // the input-gradient chain on the same kernel: f.4^T is its first layer (Cin = Cout), f.0^T its last (Cout = C / 2)
                L.cnet_bwd = d.hidden <= 512 && cnet_groups(C / 2) == 1 && cnet_supported(L.Cout, H, W, d.hidden, C / 2);
                if (L.cnet_bwd) {
                    L.cb_w0 = take(off, sh2_image_bytes(cnet_g0(L.Cout) * 8, d.hidden));
                    L.cb_w2 = take(off, sh2_image_bytes(d.hidden, d.hidden));
                    L.cb_w4 = take(off, cnet_w4_bytes(d.hidden, C / 2));
                    L.wt4 = take(off, (size_t)L.Cout * d.hidden * 9 * 4);
                    L.wt2 = take(off, (size_t)d.hidden * d.hidden * 4);
                    L.wt0 = take(off, (size_t)d.hidden * (C / 2) * 9 * 4);
                }
            }
            L.dnet = !L.cnet && d.permutation == GLOWHIP_PERM_INVCONV && dnet_supported(C, H, W, d.hidden, L.Cout);
            if (L.dnet) {
                L.dn_mix = take(off, sh2_image_bytes(C, C));
                L.dn_mixinv = take(off, sh2_image_bytes(C, C));
                L.dn_w0 = take(off, sh2_image_bytes(cnet_g0(C / 2) * 8, d.hidden));
                L.dn_w2 = take(off, sh2_image_bytes(d.hidden, d.hidden));
                L.dn_w4 = take(off, sh2_image_bytes(cnet_g0(d.hidden) * 8, L.Cout));
                p->max_hidden = std::max(p->max_hidden, (long)((dnet_scratch_bytes_per_sample(C, H, W, d.hidden, L.Cout) + 3) / 4));
            }
            if (L.mfma_last) L.f4_wp = take(off, conv_mfma_tail_packed_bytes(d.hidden, L.Cout));
            L.wide_last = !L.mfma_last && conv_mfma_wide_supported(d.hidden, H, W, L.Cout, 3);
            if (L.wide_last) L.f4_wt = take(off, conv_mfma_wide_packed_bytes(d.hidden, L.Cout, 3));
            L.dg4_first = conv_mfma_first_supported(L.Cout, H, W, d.hidden);
            if (L.dg4_first) L.f4T_wf = take(off, conv_mfma_first_packed_bytes(L.Cout, d.hidden));
            L.dg0_tail = conv_mfma_tail_supported(d.hidden, H, W, C / 2);
            if (L.dg0_tail) L.f0T_wp = take(off, conv_mfma_tail_packed_bytes(d.hidden, C / 2));
            p->max_hidden = std::max(p->max_hidden, (long)std::max(d.hidden, L.Cout) * H * W);
        } else if (d.kind == GLOWHIP_LAYER_SPLIT2D) {
            if (C % 2) return fail("Split2d needs an even channel count");
            if (!d.f4_w || !d.f4_bias || !d.f4_logs) return fail("missing conv2d_zeros parameter pointer");
            L.Cout = C;
            L.f4_scale = take(off, (size_t)C * 4);
            L.mfma_last = conv_mfma_tail_supported(C / 2, H, W, C);
            if (L.mfma_last) L.f4_wp = take(off, conv_mfma_tail_packed_bytes(C / 2, C));
            p->max_hidden = std::max(p->max_hidden, (long)C * H * W);
            p->n_split++;
            C /= 2;
        } else {
            return fail("unknown layer kind");
        }
        p->max_chw = std::max(p->max_chw, (long)C * H * W);
        p->layers.push_back(L);
    }
    p->out_shape[0] = C; p->out_shape[1] = H; p->out_shape[2] = W;
    // job tables of the batched pack
    for (const LayerPlan& L : p->layers) {
        const glowhip_layer_desc& d = L.d;
        if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
            StepPrepJob j{};
            j.w = d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr;
            j.an_logs = d.an_logs; j.C = d.C; j.HW = d.H * d.W;
            j.winv_off = L.winv; j.logabsdet_off = L.logabsdet; j.konst_off = L.konst; j.scratch_off = L.lu_scratch;
            p->prep_jobs.push_back(j);
            if (j.w && d.C <= 64) p->max_lds_c = std::max(p->max_lds_c, d.C);
            if (j.w) p->max_c = std::max(p->max_c, d.C);
            p->scale_jobs.push_back(ScaleJob{d.an_logs, L.an_scale, L.an_inv_scale, d.C, 1});
            p->scale_jobs.push_back(ScaleJob{d.f0_an_logs, L.f0_scale, 0, d.hidden, 0});
            p->scale_jobs.push_back(ScaleJob{d.f2_an_logs, L.f2_scale, 0, d.hidden, 0});
            p->scale_jobs.push_back(ScaleJob{d.f4_logs, L.f4_scale, 0, L.Cout, 0});
            // inference-use bit of the exact-fp32 images: layers that run k_cnet or the deep-level kernels read them only under the
            // family switches (bit 8)
            const int inf = (L.cnet || L.dnet) ? 8 : 1;
            if (L.first_halo) {
                RepackJob r{}; r.w = d.f0_w; r.out_off = L.f0_wt; r.kind = REPACK_FIRST; r.Cin = d.C / 2; r.Cout = d.hidden;
                r.fold_bias = d.f0_an_bias; r.fold_logs = d.f0_an_logs; r.use = 2 | inf;
                p->repack_jobs.push_back(r);
            } else if (L.mfma_first) {
                RepackJob r{}; r.w = d.f0_w; r.out_off = L.f0_wt; r.kind = REPACK_WIDE; r.Cin = d.C / 2; r.Cout = d.hidden;
                r.K = r.Cin * 9; r.Kpad = wide_kpad(r.Cin, 3); r.use = 2 | inf; p->repack_jobs.push_back(r);
            }
            if (L.first_halo && L.f0_init) {   // use bit 16 (internal): read by the data-dependent init pass only
                RepackJob r{}; r.w = d.f0_w; r.out_off = L.f0_init; r.kind = REPACK_WIDE; r.Cin = d.C / 2; r.Cout = d.hidden;
                r.K = r.Cin * 9; r.Kpad = wide_kpad(r.Cin, 3); r.use = 16; p->repack_jobs.push_back(r);
            }
            if (L.mfma_mid) {
                RepackJob r{}; r.w = d.f2_w; r.out_off = L.f2_wt; r.kind = REPACK_WIDE; r.Cin = d.hidden; r.Cout = d.hidden;
                r.K = r.Cin; r.Kpad = wide_kpad(r.Cin, 1); r.use = 2 | inf; p->repack_jobs.push_back(r);
            }
            if (L.cnet) {
                RepackJob r0{}; r0.w = d.f0_w; r0.out_off = L.cn_w0; r0.kind = REPACK_SH2_FIRST; r0.Cin = d.C / 2; r0.Cout = d.hidden;
                r0.K = cnet_g0(d.C / 2); r0.fold_bias = d.f0_an_bias; r0.fold_logs = d.f0_an_logs; r0.use = 3;
                p->repack_jobs.push_back(r0);
                RepackJob r2{}; r2.kperm = 1; r2.w = d.f2_w; r2.out_off = L.cn_w2; r2.kind = REPACK_SH2_GEMM; r2.Cin = d.hidden; r2.Cout = d.hidden;
                r2.K = d.hidden; r2.fold_bias = d.f2_an_bias; r2.fold_logs = d.f2_an_logs; r2.use = 3; p->repack_jobs.push_back(r2);
                const int ng = cnet_groups(L.Cout), cg = L.Cout / ng;       // one image per group of f.4 output channels
                for (int gi = 0; gi < ng; ++gi) {
                    RepackJob r4{}; r4.w = d.f4_w + (size_t)gi * cg * d.hidden * 9; r4.out_off = L.cn_w4 + gi * sh2_image_bytes(d.hidden, cnet_mpad4(cg));
                    r4.kind = REPACK_SH2_TAIL; r4.kperm = 1; r4.Cin = d.hidden; r4.Cout = cg;
                    r4.Kpad = cnet_mpad4(cg); r4.use = 3; p->repack_jobs.push_back(r4);
                }
            }
            if (L.dnet) {      // deep levels: the mixer as a GEMM image (W; W^-1 after the LU), f.0 / f.4 as (chunk, tap) images, f.2
                RepackJob rm{}; rm.w = d.invconv_w; rm.out_off = L.dn_mix; rm.kind = REPACK_SH2_GEMM; rm.Cin = d.C; rm.Cout = d.C; rm.K = d.C; rm.use = 1;
                p->repack_jobs.push_back(rm);
                RepackJob ri{}; ri.w = nullptr; ri.w_off = L.winv; ri.out_off = L.dn_mixinv; ri.kind = REPACK_SH2_GEMM; ri.Cin = d.C; ri.Cout = d.C;
                ri.K = d.C; ri.use = 4; ri.after_lu = 1; p->repack_jobs.push_back(ri);
                RepackJob r0{}; r0.w = d.f0_w; r0.out_off = L.dn_w0; r0.kind = REPACK_SH2_FIRST; r0.Cin = d.C / 2; r0.Cout = d.hidden;
                r0.K = cnet_g0(d.C / 2); r0.fold_bias = d.f0_an_bias; r0.fold_logs = d.f0_an_logs; r0.use = 1; p->repack_jobs.push_back(r0);
                RepackJob r2{}; r2.w = d.f2_w; r2.out_off = L.dn_w2; r2.kind = REPACK_SH2_GEMM; r2.Cin = d.hidden; r2.Cout = d.hidden;
                r2.K = d.hidden; r2.fold_bias = d.f2_an_bias; r2.fold_logs = d.f2_an_logs; r2.use = 1; p->repack_jobs.push_back(r2);
                RepackJob r4{}; r4.w = d.f4_w; r4.out_off = L.dn_w4; r4.kind = REPACK_SH2_FIRST; r4.Cin = d.hidden; r4.Cout = L.Cout;
                r4.K = cnet_g0(d.hidden); r4.use = 1; p->repack_jobs.push_back(r4);
            }
            if (L.cnet_bwd) {      // (training only) transposed copies, then the same three image kinds over them
                const int Ch = d.C / 2;
                p->flip_jobs.push_back(FlipJob{d.f4_w, L.wt4, L.Cout, d.hidden, 9});     // wt4[k][co][8 - tap] = W4[co][k][tap]
                p->flip_jobs.push_back(FlipJob{d.f2_w, L.wt2, d.hidden, d.hidden, 1});   // wt2[i][o] = W2[o][i]
                p->flip_jobs.push_back(FlipJob{d.f0_w, L.wt0, d.hidden, Ch, 9});         // wt0[ci][k][8 - tap] = W0[k][ci][tap]
                const int th = (d.hidden + 31) / 32;
                p->flip_tiles[0] = std::max(p->flip_tiles[0], th * ((L.Cout + 31) / 32));
                p->flip_tiles[1] = std::max(p->flip_tiles[1], th * th);
                p->flip_tiles[2] = std::max(p->flip_tiles[2], th * ((Ch + 31) / 32));
                RepackJob r0{}; r0.w = nullptr; r0.w_off = L.wt4; r0.out_off = L.cb_w0; r0.kind = REPACK_SH2_FIRST; r0.Cin = L.Cout; r0.Cout = d.hidden;
                r0.K = cnet_g0(L.Cout); r0.fold_bias = nullptr; r0.fold_logs = d.f2_an_logs; r0.use = 2;      // g_u2 = g_h2 (h2 > 0) exp(3 logs2)
                p->repack_jobs.push_back(r0);
                RepackJob r2{}; r2.kperm = 1; r2.w = nullptr; r2.w_off = L.wt2; r2.out_off = L.cb_w2; r2.kind = REPACK_SH2_GEMM; r2.Cin = d.hidden; r2.Cout = d.hidden;
                r2.K = d.hidden; r2.fold_bias = nullptr; r2.fold_logs = d.f0_an_logs; r2.use = 2; p->repack_jobs.push_back(r2);
                RepackJob r4{}; r4.kperm = 1; r4.w = nullptr; r4.w_off = L.wt0; r4.out_off = L.cb_w4; r4.kind = REPACK_SH2_TAIL; r4.Cin = d.hidden; r4.Cout = Ch;
                r4.Kpad = cnet_mpad4(Ch); r4.use = 2; p->repack_jobs.push_back(r4);
            }
            if (L.mfma_last) {
                RepackJob r{}; r.w = d.f4_w; r.out_off = L.f4_wp; r.kind = REPACK_TAIL; r.Cin = d.hidden; r.Cout = L.Cout;
                r.paired = d.coupling == GLOWHIP_COUPLING_AFFINE; r.MT = tail_mt(L.Cout, r.paired);
                r.use = 2 | inf;
                r.total = (long)tail_chunks(r.Cin) * (TAIL_CK / 4) * 9 * r.MT * 64; p->repack_jobs.push_back(r);
            }
            if (L.wide_last) {
                RepackJob r{}; r.w = d.f4_w; r.out_off = L.f4_wt; r.kind = REPACK_WIDE; r.Cin = d.hidden; r.Cout = L.Cout;
                r.K = r.Cin * 9; r.Kpad = wide_kpad(r.Cin, 3); r.use = 2 | inf; p->repack_jobs.push_back(r);
            }
            if (L.dg4_first) {   // input gradient of f.4 = 3x3 conv Cout -> hidden with w[ci][o][8-tap]
                RepackJob r{}; r.w = d.f4_w; r.out_off = L.f4T_wf; r.kind = REPACK_FIRST; r.Cin = L.Cout; r.Cout = d.hidden;
                r.transposed = 1; r.use = 2; p->repack_jobs.push_back(r);
            }
            if (L.dg0_tail) {    // input gradient of f.0 = 3x3 conv hidden -> C/2
                RepackJob r{}; r.w = d.f0_w; r.out_off = L.f0T_wp; r.kind = REPACK_TAIL; r.Cin = d.hidden; r.Cout = d.C / 2;
                r.paired = 0; r.MT = tail_mt(r.Cout, 0); r.transposed = 1; r.use = 2;
                r.total = (long)tail_chunks(r.Cin) * (TAIL_CK / 4) * 9 * r.MT * 64; p->repack_jobs.push_back(r);
            }
        } else if (d.kind == GLOWHIP_LAYER_SPLIT2D) {
            p->scale_jobs.push_back(ScaleJob{d.f4_logs, L.f4_scale, 0, L.Cout, 0});
            if (L.mfma_last) {
                RepackJob r{}; r.w = d.f4_w; r.out_off = L.f4_wp; r.kind = REPACK_TAIL; r.Cin = d.C / 2; r.Cout = L.Cout;
                r.paired = 1; r.MT = tail_mt(L.Cout, 1); r.use = 3;
                r.total = (long)tail_chunks(r.Cin) * (TAIL_CK / 4) * 9 * r.MT * 64; p->repack_jobs.push_back(r);
            }
        }
    }
    p->prep_off = take(off, p->prep_jobs.size() * sizeof(StepPrepJob));
    p->scale_off = take(off, p->scale_jobs.size() * sizeof(ScaleJob));
    // one slot of selected repack jobs per combination of the use bits that select images (1, 2, 8, 16): a pack of one mask never
    // rewrites the table a captured graph of another mask launches over
    p->repack_off = take(off, REPACK_SLOTS * p->repack_jobs.size() * sizeof(RepackJob));
    p->flip_off = take(off, p->flip_jobs.size() * sizeof(FlipJob));
    p->packed_bytes = align_up(off, 256);
    return p;
}

void glowhip_plan_destroy(glowhip_plan* plan) {
    if (!plan) return;
    if (plan->side) {      // (the side stream may still be writing into the caller's `packed` buffer)
        (void)hipStreamSynchronize(plan->side);
        (void)hipEventDestroy(plan->ev_fork); (void)hipEventDestroy(plan->ev_legacy); (void)hipEventDestroy(plan->ev_lu);
        (void)hipStreamDestroy(plan->side);
    }
    for (hipEvent_t e : plan->ev_pool) (void)hipEventDestroy(e);
    for (TimingSlot& t : plan->ev_used) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    delete plan;
}

int glowhip_plan_timing_enable(glowhip_plan* plan, int enable) {
    GH_REQUIRE(plan, "plan_timing_enable: null plan");
    plan->timing = enable != 0;
    if (!enable) {
        for (hipEvent_t e : plan->ev_pool) (void)hipEventDestroy(e);
        for (TimingSlot& t : plan->ev_used) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
        plan->ev_pool.clear();
        plan->ev_used.clear();
    }
    return GLOWHIP_OK;
}

int glowhip_plan_timing_read(glowhip_plan* plan, glowhip_timing_record* out, int max, int* n_out) {
    GH_REQUIRE(plan && n_out, "plan_timing_read: null argument");
    int n = 0;
    for (TimingSlot& t : plan->ev_used) {
        if (hipEventSynchronize(t.b) != hipSuccess) { set_error("plan_timing_read: hipEventSynchronize failed"); return GLOWHIP_ELAUNCH; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t.a, t.b);
        if (out && n < max) { out[n].kind = t.kind; out[n].layer = t.layer; out[n].mfma = t.mfma; out[n].ms = ms; ++n; }
        plan->ev_pool.push_back(t.a);
        plan->ev_pool.push_back(t.b);
    }
    plan->ev_used.clear();
    *n_out = n;
    return GLOWHIP_OK;
}

size_t glowhip_plan_packed_bytes(const glowhip_plan* plan) { return plan ? plan->packed_bytes : 0; }

size_t glowhip_plan_workspace_bytes(const glowhip_plan* plan, int N) {
    return (plan && N >= 0) ? workspace_bytes(plan, N) : 0;
}

int glowhip_plan_output_shape(const glowhip_plan* plan, int reverse, int32_t out[3]) {
    GH_REQUIRE(plan && out, "plan_output_shape: null argument");
    for (int i = 0; i < 3; ++i) out[i] = reverse ? plan->in_shape[i] : plan->out_shape[i];
    return GLOWHIP_OK;
}

int glowhip_plan_set_dequant_rng(glowhip_plan* plan, unsigned long long seed, int enable, unsigned long long* next_call) {
    GH_REQUIRE(plan, "plan_set_dequant_rng: null plan");
    if (enable > 0) {
        if (plan->rng_seed != seed) plan->rng_calls = 0;      // a new seed restarts the stream; switching off and on again does not
        plan->rng_on = true; plan->rng_seed = seed;
    } else if (enable == 0) {
        plan->rng_on = false;
    }
    if (next_call) *next_call = plan->rng_calls;
    return GLOWHIP_OK;
}

int glowhip_plan_set_dequant_stream(glowhip_plan* plan, unsigned long long seed, unsigned long long call) {
    GH_REQUIRE(plan, "plan_set_dequant_stream: null plan");
    plan->rng_on = true; plan->rng_seed = seed; plan->rng_calls = call;
    return GLOWHIP_OK;
}

int glowhip_dequant_noise(float* out, long n, unsigned long long seed, unsigned long long call, int n_bits, glowhip_stream_t stream) {
    GH_REQUIRE(out && n >= 0 && n_bits > 0 && n_bits <= 30, "dequant_noise: bad argument");
    return launch_dequant_noise(out, n, seed, call, (float)(1.0 / pow(2.0, n_bits)), (hipStream_t)stream);
}

int glowhip_plan_describe(const glowhip_plan* plan, char* buf, size_t buf_bytes) {
    return glowhip_plan_describe_for(plan, 0, buf, buf_bytes);
}

int glowhip_plan_launch_counts(glowhip_plan* plan, char* buf, size_t buf_bytes, int reset) {
    GH_REQUIRE(plan && buf && buf_bytes > 0, "plan_launch_counts: null argument");
    std::string out;
    for (const auto& kv : plan->launch_counts) out += kv.first + "=" + std::to_string(kv.second) + "\n";
    GH_REQUIRE(out.size() < buf_bytes, "plan_launch_counts: buffer too small");
    snprintf(buf, buf_bytes, "%s", out.c_str());
    if (reset) plan->launch_counts.clear();
    return GLOWHIP_OK;
}

int glowhip_plan_describe_for(const glowhip_plan* plan, int N, char* buf, size_t buf_bytes) {
    GH_REQUIRE(plan && buf && buf_bytes > 0, "plan_describe: null argument");
    std::string sdesc;
    char line[256];
    int li = 0;
    for (const LayerPlan& L : plan->layers) {
        const glowhip_layer_desc& d = L.d;
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) snprintf(line, sizeof line, "%d squeeze C=%d H=%d W=%d\n", li, d.C, d.H, d.W);
        else if (d.kind == GLOWHIP_LAYER_FLOWSTEP)
        {
            if (dnet_runs(plan, L))
                snprintf(line, sizeof line, "%d flowstep C=%d H=%d W=%d hidden=%d f=dnet-sh2 (mix, f.0, f.2, f.4, finish: one launch per layer)\n", li, d.C,
                         d.H, d.W, d.hidden);
            else if (cnet_runs(plan, L))
                snprintf(line, sizeof line, "%d flowstep C=%d H=%d W=%d hidden=%d f=cnet-sh2 (f.0+f.2+f.4 one kernel + finish)\n", li, d.C, d.H,
                         d.W, d.hidden);
            else      // the exact-fp32 kernels, layer by layer
                snprintf(line, sizeof line, "%d flowstep C=%d H=%d W=%d hidden=%d f0=%s f2=%s f4=%s\n", li, d.C, d.H, d.W, d.hidden,
                         L.first_halo ? "mfma-halo" : (L.mfma_first ? "mfma" : "direct"), L.mfma_mid ? "mfma" : "direct",
                         L.mfma_last ? "mfma" : (L.wide_last ? "mfma-wide" : "direct"));
        }
        else snprintf(line, sizeof line, "%d split2d C=%d H=%d W=%d prior=%s\n", li, d.C, d.H, d.W,
                      L.mfma_last ? "mfma" : "direct");
        sdesc += line;
        ++li;
    }
    snprintf(buf, buf_bytes, "%s", sdesc.c_str());
    return GLOWHIP_OK;
}

int glowhip_plan_pack(glowhip_plan* plan, void* packed, size_t packed_bytes, glowhip_stream_t stream) {
    return glowhip_plan_pack_for(plan, packed, packed_bytes, GLOWHIP_PACK_INFERENCE | GLOWHIP_PACK_TRAINING | GLOWHIP_PACK_INVERSE, stream);
}

int glowhip_plan_pack_for(glowhip_plan* plan, void* packed, size_t packed_bytes, int use, glowhip_stream_t stream) {
    GH_REQUIRE(plan && packed, "plan_pack: null argument");
    GH_REQUIRE(use & (GLOWHIP_PACK_INFERENCE | GLOWHIP_PACK_TRAINING), "plan_pack: empty use mask");
    // use bit 8 (internal): the exact-fp32 images of layers that normally run k_cnet / the deep-level kernels -- read by the inference
    // calls only with the split-half path switched off through the debug hook
    // (OR-ed in: the internal bits 16 = init pass' f.0 image and 32 = no LU of the caller's mask survive)
    if (g_sh_disabled) use |= GLOWHIP_PACK_INFERENCE | GLOWHIP_PACK_TRAINING | 8;
    // a plan on the exact-fp32 family reads the fp32 MFMA images, which are the training path's
    if (plan->family == GLOWHIP_FAMILY_EXACT_FP32) use |= GLOWHIP_PACK_TRAINING;
    plan->repack_sel.clear();
    int n_kind[5] = {0, 0, 0, 0, 0}, tail_blocks = 1, first_blocks = 2;
    // (group 4: SH2 GEMM images of W^-1 -- launched after the LU factorisations, on their stream)
    auto group = [](const RepackJob& r) { return r.after_lu ? 4 : (r.kind < REPACK_SH2_GEMM ? 0 : r.kind - REPACK_SH2_GEMM + 1); };
    for (int gk = 0; gk < 5; ++gk)            // sorted by kind group: each image kernel is launched over its own jobs only
        for (const RepackJob& r : plan->repack_jobs)
            if ((r.use & use) && group(r) == gk) {
                plan->repack_sel.push_back(r);
                ++n_kind[gk];
                if (gk == 3) tail_blocks = std::max(tail_blocks, (r.Cout + 7) / 8);
                if (gk == 2 && r.Cin >= 64) first_blocks = std::max(first_blocks, std::min(16, (r.Cout + 31) / 32));
            }
    GH_REQUIRE(packed_bytes >= plan->packed_bytes, "plan_pack: packed buffer too small (%zu < %zu)", packed_bytes,
               plan->packed_bytes);
    hipStream_t s = (hipStream_t)stream;
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));      // (a previous pack's side-stream part writes the same buffer)
    // job tables -> device (plan-constant contents; re-sent because `packed` is caller memory), then 4 launches
    if (hipMemsetAsync(packed, 0, 256, s) != hipSuccess) {
        set_error("plan_pack: hipMemsetAsync failed");
        return GLOWHIP_ELAUNCH;
    }
    auto upload = [&](size_t off, const void* src, size_t bytes) {
        return bytes == 0 || hipMemcpyAsync((char*)packed + off, src, bytes, hipMemcpyHostToDevice, s) == hipSuccess;
    };
    // The job tables are plan constants: they travel once per buffer -- the selected repack jobs once per (buffer, image bits of the
    // use mask), each mask into its OWN slot -- and stay in `packed`: a re-pack of the same buffer is then kernel launches only (no
    // host-to-device copy per step, and the sequence can be captured in a hipGraph whose table no later pack of another mask --
    // a training step, the exact-fp32 fall-back, the init pass -- overwrites).
    const int slot = repack_slot(use);
    const size_t slot_off = plan->repack_off + (size_t)slot * plan->repack_jobs.size() * sizeof(RepackJob);
    if (plan->tables_in != packed) {
        if (!upload(plan->prep_off, plan->prep_jobs.data(), plan->prep_jobs.size() * sizeof(StepPrepJob)) ||
            !upload(plan->scale_off, plan->scale_jobs.data(), plan->scale_jobs.size() * sizeof(ScaleJob)) ||
            !upload(plan->flip_off, plan->flip_jobs.data(), plan->flip_jobs.size() * sizeof(FlipJob))) {
            set_error("plan_pack: hipMemcpyAsync of the job tables failed");
            return GLOWHIP_ELAUNCH;
        }
        plan->tables_in = packed; plan->slots_in = 0;
    }
    if (!(plan->slots_in & (1u << slot))) {
        // (hipMemcpyAsync from pageable memory stages the source before it returns; the per-slot host copy is kept anyway)
        std::vector<RepackJob>& keep = plan->repack_slot_host[slot];
        keep = plan->repack_sel;
        if (!upload(slot_off, keep.data(), keep.size() * sizeof(RepackJob))) {
            set_error("plan_pack: hipMemcpyAsync of the repack job table failed");
            return GLOWHIP_ELAUNCH;
        }
        plan->slots_in |= 1u << slot;
    }
    // Fork: the legacy-kind images and the LU factorisations go to the plan's side stream (created on first use; a host resource
    // like the timing events) behind everything enqueued so far; whoever reads their results joins (join_legacy / join_lu).  What
    // the first kernels of a forward need -- scale tables, the product kernels' images -- stays on `stream`.
    // Only where it pays: plans with invertible 1x1 convolutions beyond 128 channels (blocked LU in global memory, 2.4 ms per pack
    // at config E: +5 % on its forward).  At config B the side stream's workgroups only get in the way of the k_cnet launch that
    // runs beside them (-0.5 %): everything stays on `stream` there.
    hipStream_t side = s;
    if (!g_pack_one_stream && plan->max_c > 128) {
        if (!plan->side) {
            if (hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&plan->ev_legacy, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&plan->ev_lu, hipEventDisableTiming) != hipSuccess) {
                set_error("plan_pack: could not create the side stream / events");
                return GLOWHIP_ELAUNCH;
            }
        }
        side = plan->side;
        if (hipEventRecord(plan->ev_fork, s) != hipSuccess || hipStreamWaitEvent(side, plan->ev_fork, 0) != hipSuccess) {
            set_error("plan_pack: fork onto the side stream failed");
            return GLOWHIP_ELAUNCH;
        }
    }
    if (use & GLOWHIP_PACK_TRAINING)      // transposed weight copies for the backward k_cnet images (read by the image kernels below)
        GH_TRY(launch_flipT_batched(at<FlipJob>(packed, plan->flip_off), (int)plan->flip_jobs.size(), plan->flip_tiles, packed, s));
    GH_TRY(launch_pack_batched(at<ScaleJob>(packed, plan->scale_off), (int)plan->scale_jobs.size(),
                               at<RepackJob>(packed, slot_off), n_kind, tail_blocks, packed, s, side, first_blocks));
    if (side != s) {
        if (hipEventRecord(plan->ev_legacy, side) != hipSuccess) { set_error("plan_pack: hipEventRecord failed"); return GLOWHIP_ELAUNCH; }
        plan->legacy_pending = true; plan->pending_captured = stream_capturing(s);
    }
    if (!(use & 32)) {      // (32, internal: weight images and scale tables only -- the init pass' first pack)
        bool all_small = !g_lu_small_off;
        for (const StepPrepJob& pj : plan->prep_jobs) all_small = all_small && step_prepare_small_takes(pj);
        const bool inv = (use & (GLOWHIP_PACK_INVERSE | GLOWHIP_PACK_TRAINING)) != 0;
        count_launch(plan, (all_small && !inv) ? "pack:k_step_prepare_small" : "pack:k_step_prepare_batched");
        GH_TRY(launch_step_prepare_batched(at<StepPrepJob>(packed, plan->prep_off), (int)plan->prep_jobs.size(),
                                           plan->max_lds_c, packed, side, inv, plan->max_c, all_small ? 1 : 0));
        if ((use & GLOWHIP_PACK_INVERSE) && n_kind[4] > 0)
            GH_TRY(launch_repack_sh2_gemm(at<RepackJob>(packed, slot_off) + (n_kind[0] + n_kind[1] + n_kind[2] + n_kind[3]), n_kind[4], packed, side));
    }
    if (side != s) {
        if (hipEventRecord(plan->ev_lu, side) != hipSuccess) { set_error("plan_pack: hipEventRecord failed"); return GLOWHIP_ELAUNCH; }
        plan->lu_pending = true; plan->pending_captured = stream_capturing(s);
    }
    return GLOWHIP_OK;
}

int glowhip_plan_set_family(glowhip_plan* plan, int family) {
    GH_REQUIRE(plan, "plan_set_family: null plan");
    GH_REQUIRE(family == GLOWHIP_FAMILY_AUTO || family == GLOWHIP_FAMILY_EXACT_FP32, "plan_set_family: unknown family %d", family);
    plan->family = family;
    return GLOWHIP_OK;
}

int glowhip_plan_get_family(const glowhip_plan* plan) { return plan ? plan->family : GLOWHIP_EINVAL; }

int glowhip_plan_status(const glowhip_plan* plan, const void* workspace, size_t workspace_bytes, int N, const float* result,
                        long elems_per_sample, int32_t* status_out, glowhip_stream_t stream) {
    GH_REQUIRE(plan && status_out, "plan_status: null argument");
    GH_REQUIRE(N >= 0 && N <= 65535, "batch size %d out of range", N);
    if (N == 0) return GLOWHIP_OK;
    Workspace w;
    GH_TRY(carve(plan, N, const_cast<void*>(workspace), workspace_bytes, w));
    return launch_status(w.acc, N, result, elems_per_sample, status_out, (hipStream_t)stream);
}

int glowhip_plan_pack_sync(glowhip_plan* plan) {
    GH_REQUIRE(plan, "plan_pack_sync: null plan");
    if (plan->side && (plan->legacy_pending || plan->lu_pending)) {
        if (hipStreamSynchronize(plan->side) != hipSuccess) { set_error("plan_pack_sync: hipStreamSynchronize failed"); return GLOWHIP_ELAUNCH; }
        plan->legacy_pending = plan->lu_pending = false;
    }
    return GLOWHIP_OK;
}

int glowhip_plan_forget_packed(glowhip_plan* plan) {
    GH_REQUIRE(plan, "plan_forget_packed: null plan");
    plan->tables_in = nullptr; plan->slots_in = 0;
    return GLOWHIP_OK;
}

int glowhip_plan_encode(glowhip_plan* plan, const void* packed, const float* x, const float* noise,
                        const float* logdet_in, float* z, float* logdet_out, int N, void* workspace,
                        size_t workspace_bytes, glowhip_stream_t stream) {
    GH_TRY(check_plan_args(plan, packed, N));
    GH_REQUIRE(x && z, "plan_encode: null tensor");
    if (N == 0) return GLOWHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    Workspace w;
    GH_TRY(carve(plan, N, workspace, workspace_bytes, w));
    GH_TRY(launch_zero_acc(w.acc, N, s, ACC_EXTRA, w.fin_cnt, fin_cnt_words(plan, N)));
    GH_TRY(run_forward(plan, packed, x, noise, z, N, w, s));
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));
    if (logdet_out)
        GH_TRY(launch_finalize(logdet_in, w.acc, at<double>(packed, 0), 1.0, 0.0, 1.0, logdet_out, nullptr, N, s, ACC_EXTRA));
    return GLOWHIP_OK;
}

int glowhip_plan_decode(glowhip_plan* plan, const void* packed, const float* z, const float* const* eps, int n_eps,
                        const float* logdet_in, float* x, float* logdet_out, int N, void* workspace,
                        size_t workspace_bytes, glowhip_stream_t stream) {
    GH_TRY(check_plan_args(plan, packed, N));
    GH_REQUIRE(x && z, "plan_decode: null tensor");
    GH_REQUIRE(n_eps >= plan->n_split, "plan_decode: %d eps draws given, plan has %d Split2d layers", n_eps,
               plan->n_split);
    if (N == 0) return GLOWHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    Workspace w;
    GH_TRY(carve(plan, N, workspace, workspace_bytes, w));
    GH_TRY(launch_zero_acc(w.acc, N, s, ACC_EXTRA, w.fin_cnt, fin_cnt_words(plan, N)));
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));      // decode reads W^-1 and the deep levels' images first
    GH_TRY(run_reverse(plan, packed, z, eps, n_eps, x, N, w, s));
    if (logdet_out)
        GH_TRY(launch_finalize(logdet_in, w.acc, at<double>(packed, 0), -1.0, 0.0, 1.0, logdet_out, nullptr, N, s, ACC_EXTRA));
    return GLOWHIP_OK;
}

int glowhip_glow_forward(glowhip_plan* plan, const void* packed, const float* x, const float* noise,
                         const float* prior_mean, const float* prior_logs, long prior_stride, int n_bits, float* z,
                         float* nll_out, float* objective_out, int N, void* workspace, size_t workspace_bytes,
                         glowhip_stream_t stream) {
    GH_TRY(check_plan_args(plan, packed, N));
    GH_REQUIRE(x && z && nll_out, "glow_forward: null tensor");
    GH_REQUIRE(n_bits > 0 && n_bits <= 30, "glow_forward: n_bits=%d", n_bits);
    if (N == 0) return GLOWHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    Workspace w;
    GH_TRY(carve(plan, N, workspace, workspace_bytes, w));
    GH_TRY(launch_zero_acc(w.acc, N, s, ACC_EXTRA, w.fin_cnt, fin_cnt_words(plan, N)));
    RngSpec rng{plan->rng_on && !noise, plan->rng_seed, plan->rng_calls, (float)(1.0 / pow(2.0, n_bits))};
    if (rng.on) ++plan->rng_calls;
    GH_TRY(run_forward(plan, packed, x, noise, z, N, w, s, 0, rng.on ? &rng : nullptr));
    const int* o = plan->out_shape;
    GH_TRY(launch_gaussian_logp(z, (long)o[0] * o[1] * o[2], prior_mean, prior_logs, prior_stride, N, o[0], o[1] * o[2],
                                w.acc, s));
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));      // the sum of the log|det W| terms enters here
    // objective = -ln(n_bins)*CHW + logdet + logp;  nll = -objective / (ln2 * CHW)   (network/model.py:425-450)
    const double chw = (double)plan->in_shape[0] * plan->in_shape[1] * plan->in_shape[2];
    const double offset = -log(pow(2.0, n_bits)) * chw;
    const double scale = -1.0 / (log(2.0) * chw);
    GH_TRY(launch_finalize(nullptr, w.acc, at<double>(packed, 0), 1.0, offset, scale, nll_out, objective_out, N, s, ACC_EXTRA));
    return GLOWHIP_OK;
}

// Glow.normal_flow from 8-bit pixels: the leading Squeeze2d reads the bytes itself
int glowhip_glow_forward_u8(glowhip_plan* plan, const void* packed, const uint8_t* x_u8, float divisor, const float* noise,
                            const float* prior_mean, const float* prior_logs, long prior_stride, int n_bits, float* z,
                            float* nll_out, float* objective_out, int N, void* workspace, size_t workspace_bytes,
                            glowhip_stream_t stream) {
    GH_TRY(check_plan_args(plan, packed, N));
    GH_REQUIRE(x_u8 && z && nll_out, "glow_forward_u8: null tensor");
    GH_REQUIRE(divisor > 0.f, "glow_forward_u8: divisor must be positive");
    GH_REQUIRE(n_bits > 0 && n_bits <= 30, "glow_forward_u8: n_bits=%d", n_bits);
    GH_REQUIRE(plan->layers.size() >= 2 && plan->layers[0].d.kind == GLOWHIP_LAYER_SQUEEZE,
               "glow_forward_u8: the plan must start with a Squeeze2d layer (and not end with it)");
    if (N == 0) return GLOWHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    Workspace w;
    GH_TRY(carve(plan, N, workspace, workspace_bytes, w));
    GH_TRY(launch_zero_acc(w.acc, N, s, ACC_EXTRA, w.fin_cnt, fin_cnt_words(plan, N)));
    const glowhip_layer_desc& d0 = plan->layers[0].d;
    plan->cur_layer = 0;
    RngSpec rng{plan->rng_on && !noise, plan->rng_seed, plan->rng_calls, (float)(1.0 / pow(2.0, n_bits))};
    if (rng.on) ++plan->rng_calls;
    (void)d0;
    GH_TRY(run_forward(plan, packed, w.bufA, noise, z, N, w, s, 0, rng.on ? &rng : nullptr, x_u8, divisor));
    const int* o = plan->out_shape;
    GH_TRY(launch_gaussian_logp(z, (long)o[0] * o[1] * o[2], prior_mean, prior_logs, prior_stride, N, o[0], o[1] * o[2],
                                w.acc, s));
    const double chw = (double)plan->in_shape[0] * plan->in_shape[1] * plan->in_shape[2];
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));
    return launch_finalize(nullptr, w.acc, at<double>(packed, 0), 1.0, -log(pow(2.0, n_bits)) * chw, -1.0 / (log(2.0) * chw),
                           nll_out, objective_out, N, s, ACC_EXTRA);
}

int glowhip_plan_actnorm_init(glowhip_plan* plan, void* packed, size_t packed_bytes, const float* x, const float* noise,
                              float actnorm_scale, int N, void* workspace, size_t workspace_bytes,
                              glowhip_stream_t stream) {
    GH_TRY(check_plan_args(plan, packed, N));
    GH_REQUIRE(x && N > 0, "plan_actnorm_init: empty batch");
    hipStream_t s = (hipStream_t)stream;
    Workspace w;
    GH_TRY(carve(plan, N, workspace, workspace_bytes, w));
    GH_TRY(launch_zero_acc(w.acc, N, s, ACC_EXTRA, w.fin_cnt, fin_cnt_words(plan, N)));
    // plain (ActNorm-free) fp32 MFMA weight images of every convolution: the training family's + the init pass's own f.0 image
    // (no LU here: the invertible 1x1 convolutions are applied with W itself, and the pack at the end factorises them)
    GH_TRY(glowhip_plan_pack_for(plan, packed, packed_bytes, GLOWHIP_PACK_TRAINING | 16 | 32, stream));
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));
    // Layer by layer: set the ActNorm statistics from the activations that reach it, refresh the packed
    // data of that layer, then run the layer forward with the fresh parameters (first training-mode
    // forward of the reference: network/module.py:45-46,66-67).
    const float* cur = x;
    const int nl = (int)plan->layers.size();
    for (int li = 0; li < nl; ++li) {
        LayerPlan& L = plan->layers[li];
        const glowhip_layer_desc& d = L.d;
        float* dst = other_buf(w, cur);
        const int HW = d.H * d.W;
        const long chw = (long)d.C * HW;
        const int Ch = d.C / 2;
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) {
            GH_TRY(launch_squeeze(cur, noise, dst, N, d.C, d.H, d.W, 2, 0, s));
            noise = nullptr;
        } else if (d.kind == GLOWHIP_LAYER_SPLIT2D) {
            GH_TRY(launch_copy_strided(cur, chw, dst, (long)Ch * HW, N, (long)Ch * HW, s));
        } else {
            if (noise) {
                GH_TRY(launch_squeeze(cur, noise, dst, N, d.C, d.H, d.W, 1, 0, s));
                cur = dst; dst = other_buf(w, cur); noise = nullptr;
            }
            const int hid = d.hidden;
            GH_TRY(launch_actnorm_init(cur, chw, N, d.C, HW, actnorm_scale, (float*)d.an_bias, (float*)d.an_logs, s));
            GH_TRY(pack_scales(d.an_logs, d.C, at<float>(packed, L.an_scale), at<float>(packed, L.an_inv_scale), s));
            ChanMixArgs m{};
            m.in_a = cur; m.in_a_bs = chw; m.in_b = cur + (long)Ch * HW; m.in_b_bs = chw; m.Ca = Ch;
            m.out = dst; m.out_bs = chw; m.bias = d.an_bias; m.scale = at<float>(packed, L.an_scale);
            m.matrix = d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr;
            m.gather = d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx : nullptr;
            m.reverse = 0; m.N = N; m.C = d.C; m.HW = HW;
            GH_TRY(launch_chanmix(m, s));
            // The statistics of a Conv2d's ActNorm are taken BETWEEN the convolution and the ActNorm (network/module.py:258-259,
            // 86-120), so the init pass needs the un-fused form -- not the slow one: raw convolution on the exact-fp32 MFMA kernels
            // (plain weight images packed above), statistics, then ActNorm + ReLU as one in-place pass.  The generic direct kernel
            // (two launches per convolution before) runs only for shapes no MFMA kernel takes.
            // f.0 (Conv2d's ActNorm uses scale 1, network/module.py:239)
            if (L.f0_init) {
                GH_TRY(launch_conv_mfma_wide(dst, chw, at<float>(packed, L.f0_init), nullptr, nullptr, w.h1, N, Ch, d.H, d.W, hid, 3, s, 0));
            } else {
                ConvArgs c0{dst, chw, d.f0_w, nullptr, nullptr, nullptr, nullptr, 0, w.h1, N, Ch, d.H, d.W, hid, 3};
                GH_TRY(launch_conv_direct(c0, s));
            }
            GH_TRY(launch_actnorm_init(w.h1, (long)hid * HW, N, hid, HW, 1.0f, (float*)d.f0_an_bias, (float*)d.f0_an_logs, s));
            GH_TRY(pack_scales(d.f0_an_logs, hid, at<float>(packed, L.f0_scale), nullptr, s));
            GH_TRY(launch_bias_scale_relu(w.h1, N, hid, HW, d.f0_an_bias, at<float>(packed, L.f0_scale), s));
            // f.2
            if (L.mfma_mid) {
                GH_TRY(launch_conv_mfma_wide(w.h1, (long)hid * HW, at<float>(packed, L.f2_wt), nullptr, nullptr, w.h2, N, hid, d.H, d.W,
                                             hid, 1, s, 0));
            } else {
                ConvArgs c2{w.h1, (long)hid * HW, d.f2_w, nullptr, nullptr, nullptr, nullptr, 0, w.h2, N, hid, d.H, d.W, hid, 1};
                GH_TRY(launch_conv_direct(c2, s));
            }
            GH_TRY(launch_actnorm_init(w.h2, (long)hid * HW, N, hid, HW, 1.0f, (float*)d.f2_an_bias, (float*)d.f2_an_logs, s));
            GH_TRY(pack_scales(d.f2_an_logs, hid, at<float>(packed, L.f2_scale), nullptr, s));
            GH_TRY(launch_bias_scale_relu(w.h2, N, hid, HW, d.f2_an_bias, at<float>(packed, L.f2_scale), s));
            // f.4 + coupling
            GH_TRY(pack_scales(d.f4_logs, L.Cout, at<float>(packed, L.f4_scale), nullptr, s));
            float* z2 = dst + (long)Ch * HW;
            if (L.mfma_last) {
                TailConvArgs t{};
                t.x = w.h2; t.x_bs = (long)hid * HW; t.wp = at<float>(packed, L.f4_wp); t.bias = d.f4_bias;
                t.scale = at<float>(packed, L.f4_scale);
                t.N = N; t.Cin = hid; t.H = d.H; t.W = d.W; t.Cout = L.Cout;
                t.mode = d.coupling == GLOWHIP_COUPLING_AFFINE ? TAIL_AFFINE_FWD : TAIL_ADD_FWD;
                t.z2_in = z2; t.z2_in_bs = chw; t.z2_out = z2; t.z2_out_bs = chw; t.acc = w.acc;     // (the log-det sums are not used)
                t.zeros = at<float>(packed, 64);
                GH_TRY(launch_conv_mfma_tail(t, s));
            } else {
                if (L.wide_last) {
                    const size_t out_f = (size_t)N * L.Cout * HW, h1_f = (size_t)N * plan->max_hidden;
                    GH_TRY(launch_conv_mfma_wide(w.h2, (long)hid * HW, at<float>(packed, L.f4_wt), d.f4_bias, at<float>(packed, L.f4_scale),
                                                 w.h1, N, hid, d.H, d.W, L.Cout, 3, s, 0, h1_f > out_f ? w.h1 + out_f : nullptr,
                                                 h1_f > out_f ? h1_f - out_f : 0));
                } else {
                    ConvArgs c4{w.h2, (long)hid * HW, d.f4_w, d.f4_bias, nullptr, nullptr, at<float>(packed, L.f4_scale), 0, w.h1,
                                N, hid, d.H, d.W, L.Cout, 3};
                    GH_TRY(launch_conv_direct(c4, s));
                }
                CouplingTailArgs t{w.h1, z2, chw, z2, chw, N, Ch, HW, d.coupling == GLOWHIP_COUPLING_AFFINE, 0, nullptr};
                GH_TRY(launch_coupling_tail(t, s));
            }
        }
        cur = dst;
    }
    // everything derived from the parameters is stale now: re-derive what the inference kernels read (log|det W| without W^-1:
    // the inverse and the training images are packed on demand by whoever decodes or trains next -- W^-1 of config E's 384 x 384
    // matrices alone costs more than the rest of the init pass)
    return glowhip_plan_pack_for(plan, packed, packed_bytes, GLOWHIP_PACK_INFERENCE, stream);
}

}  // extern "C"
