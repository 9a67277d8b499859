// plan_train.hip -- training step of a flow plan: forward that records a TAPE, and the reverse
// sweep that turns dL/dnll into parameter gradients.
//
// Tape (one contiguous caller-provided buffer): the output of every layer, and per FlowStep the coupling network's
// hidden activations h1, h2 (N,hidden,H,W) -- fp16 where the taping k_cnet wrote them (tape_has_masks; the slot keeps its fp32
// size), fp32 from the per-layer kernels -- and its post-scale output hout (N,Cout,H,W); per Split2d the prior
// conv output.  With 288 GB of HBM keeping them (11.6 GB at B=64 for the celeba64 model) is cheaper than the
// reversible-recompute alternative (a second forward through the MFMA-bound coupling nets).
//
// Backward of one FlowStep (reference forward: network/model.py:82-117), all gradients "g_": see backward.hip for the
// element-wise formulas.  Convolution input gradients are convolutions with flipped/transposed weights and reuse the
// forward kernels; weight gradients use k_wgrad_direct for now (correctness first -- the MFMA wgrad is the next
// step of this row).
#include "plan_internal.h"
#include "backward.h"

using namespace glowhip;

namespace glowhip {

struct TapeLayer { size_t out = 0, h1 = 0, h2 = 0, hout = 0, m1 = 0, m2 = 0; };   // m1 / m2: sign bits of h1 / h2 (k_cnet MODE 1)

static size_t tape_layout(const glowhip_plan* p, int N, std::vector<TapeLayer>* tl) {
    size_t off = 0;
    if (tl) tl->resize(p->layers.size());
    for (size_t i = 0; i < p->layers.size(); ++i) {
        const LayerPlan& L = p->layers[i];
        const glowhip_layer_desc& d = L.d;
        TapeLayer t;
        const size_t hw = (size_t)d.H * d.W;
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) {
            t.out = take(off, (size_t)N * d.C * hw * 4);
        } else if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
            t.out = take(off, (size_t)N * d.C * hw * 4);
            t.h1 = take(off, (size_t)N * d.hidden * hw * 4);
            t.h2 = take(off, (size_t)N * d.hidden * hw * 4);
            t.hout = take(off, (size_t)N * L.Cout * hw * 4);
            if (L.cnet) {
                t.m1 = take(off, (size_t)N * d.hidden * hw / 8);
                t.m2 = take(off, (size_t)N * d.hidden * hw / 8);
            }
        } else {
            t.out = take(off, (size_t)N * (d.C / 2) * hw * 4);
            t.hout = take(off, (size_t)N * d.C * hw * 4);
        }
        if (tl) (*tl)[i] = t;
    }
    return align_up(off, 256);
}

// backward workspace: acc u64 (N) | gld (N) | gsum | gA | gB | gh1 | gh2 | gpre | wT | fp64 accumulators
struct TrainWs {
    unsigned long long* acc; float* gld; double* gsum;
    float* gA; float* gB; float* gh1; float* gh2; float* gpre; float* wT; double* dacc;
    float* gsh;                   // partial-sum scratch of the taping / backward k_cnet launches
    float* col; float* partial;   // shift-expanded small operand / split-K partial tiles of the MFMA weight gradients
    size_t partial_floats;        // floats of ONE of the three split-K partial regions behind `partial`
    GradJob* jobs;                // device copy of the finalize job table (<= 9 per layer)
    LogsJob* ljobs;               // ... and of the log-scale job table (<= 2 per layer)
    size_t dacc_doubles;
};

static size_t max_weight_floats(const glowhip_plan* p) {
    size_t m = 0;
    for (const LayerPlan& L : p->layers) {
        const glowhip_layer_desc& d = L.d;
        if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
            m = std::max(m, (size_t)d.hidden * (d.C / 2) * 9);
            m = std::max(m, (size_t)d.hidden * d.hidden);
            m = std::max(m, (size_t)L.Cout * d.hidden * 9);
        } else if (d.kind == GLOWHIP_LAYER_SPLIT2D) {
            m = std::max(m, (size_t)d.C * (d.C / 2) * 9);
        }
    }
    return m;
}

// fp64 accumulators of every layer's reduction-type gradients live side by side: zeroed once, converted once
#ifndef GH_MIX_ACC_COPIES
#define GH_MIX_ACC_COPIES 16
#endif
constexpr int MIX_ACC_COPIES = GH_MIX_ACC_COPIES;     // copies of a FlowStep's mixer accumulators [W C*C][an_b C][an_l C] (backward.h ChanMixBwdArgs).
                                       // (64 copies, one per workgroup of the C = 48 launch and plain stores instead of its 154 k fp64
                                       // atomics, changed nothing in that launch -- it was not waiting for them -- and cost the
                                       // finalize kernel 43 us per step for the 4 x longer sums)
static size_t layer_acc_doubles(const LayerPlan& L) {
    const glowhip_layer_desc& d = L.d;
    if (d.kind == GLOWHIP_LAYER_FLOWSTEP) return MIX_ACC_COPIES * ((size_t)d.C * d.C + 2 * d.C + 2 * L.Cout) + 4 * d.hidden;
    if (d.kind == GLOWHIP_LAYER_SPLIT2D) return (size_t)2 * L.Cout;
    return 0;
}
static size_t max_acc_doubles(const glowhip_plan* p) {
    size_t m = 0;
    for (const LayerPlan& L : p->layers) m += layer_acc_doubles(L);
    return m + 64;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static bool g_train_sh = true;   // testing hook: 0 = exact-fp32 kernels for f.2 in the training step too
// (the plan's own family, glowhip_plan_set_family, decides first: GLOWHIP_FAMILY_EXACT_FP32 keeps the f16 pipe out of the training step)
static bool train_sh_enabled(const glowhip_plan* p) { return g_train_sh && !(p && p->family == GLOWHIP_FAMILY_EXACT_FP32); }
void plan_train_disable_sh(int off) { g_train_sh = off == 0; }
static bool g_train_cnet = true;   // testing hook: 0 = the training forward on the per-layer kernels (no taping k_cnet)
void plan_train_disable_cnet(int off) { g_train_cnet = off == 0; }

// Does FlowStep L run its training forward as the product path's two launches -- k_cnet storing h1 / h2 from its epilogues, the
// finishing kernel storing hout and the step output (cnet_sh.hip, TAPE)?  `scratch_floats`: room for the partial sums.
static bool tape_cnet(const glowhip_plan* p, const LayerPlan& L, int N, size_t scratch_floats);
static bool g_train_cnet_bwd = true;   // testing hook: 0 = the input-gradient chain on the per-layer kernels
void plan_train_disable_cnet_bwd(int off) { g_train_cnet_bwd = off == 0; }
// ... and its input-gradient chain as one backward k_cnet launch?  Needs the taping forward (the sign bits) and all of the
// coupling network's weight gradients requested (the log-scale gradients are derived from them).
static bool bwd_cnet(const glowhip_plan* p, const LayerPlan& L, int li, const glowhip_layer_grads& G, int N, size_t scratch_floats) {
    const glowhip_layer_desc& d = L.d;
    return g_train_cnet_bwd && L.cnet_bwd && tape_cnet(p, L, N, scratch_floats) && G.f0_w && G.f2_w && G.f4_w &&
           li < (int)p->tape_has_masks.size() && p->tape_has_masks[li] &&      // (the forward that filled this tape stored the sign bits)
           cnet_tape_supported(L.Cout, d.H, d.W, d.hidden, d.C / 2, N) &&
           cnet_scratch_floats(N, d.H, d.W, d.C / 2) <= scratch_floats;
}
static bool tape_cnet(const glowhip_plan* p, const LayerPlan& L, int N, size_t scratch_floats) {
    const glowhip_layer_desc& d = L.d;
    return g_train_cnet && train_sh_enabled(p) && d.kind == GLOWHIP_LAYER_FLOWSTEP && L.cnet && d.C <= 96 &&
           cnet_tape_supported(d.C / 2, d.H, d.W, d.hidden, L.Cout, N) && cnet_scratch_floats(N, d.H, d.W, L.Cout) <= scratch_floats;
}

static bool wgrad_fast(const LayerPlan& L) {
    const glowhip_layer_desc& d = L.d;
    return d.kind == GLOWHIP_LAYER_FLOWSTEP && (d.H * d.W) % 32 == 0 && d.hidden % 128 == 0;
}

static void wgrad_scratch_floats(const glowhip_plan* p, int N, size_t* col, size_t* partial) {
    *col = 0; *partial = 0;
    for (const LayerPlan& L : p->layers) {
        if (!wgrad_fast(L)) continue;
        const glowhip_layer_desc& d = L.d;
        const int HW = d.H * d.W, hid = d.hidden;
        const int m4 = round_up(L.Cout * 9, 128), n0 = round_up((d.C / 2) * 9, 64);
        *col = std::max(*col, (size_t)N * std::max(m4, n0) * HW);
        *partial = std::max(*partial, wgrad_mfma_partial_floats(hid, hid, N, HW));
        *partial = std::max(*partial, wgrad_mfma_partial_floats(m4, hid, N, HW));
        *partial = std::max(*partial, wgrad_mfma_partial_floats(hid, n0, N, HW));
    }
}

static size_t train_ws_layout(const glowhip_plan* p, int N, void* base, TrainWs* w) {
    size_t off = 0;
    const size_t o_acc = take(off, (size_t)N * 8 * (2 + ACC_EXTRA)), o_gld = take(off, (size_t)N * 4), o_gsum = take(off, 64);
    const size_t o_gA = take(off, (size_t)N * p->max_chw * 4), o_gB = take(off, (size_t)N * p->max_chw * 4);
    const size_t o_h1 = take(off, (size_t)N * p->max_hidden * 4), o_h2 = take(off, (size_t)N * p->max_hidden * 4);
    const size_t o_gpre = take(off, (size_t)N * p->max_chw * 4);
    const size_t o_gsh = take(off, (size_t)N * p->max_hidden * 4);
    const size_t o_wT = take(off, max_weight_floats(p) * 4);
    const size_t nd = max_acc_doubles(p);
    const size_t o_dacc = take(off, nd * 8);
    size_t colf, partf;
    wgrad_scratch_floats(p, N, &colf, &partf);
    const size_t o_col = take(off, colf * 4), o_part = take(off, 3 * partf * 4);      // three regions: a FlowStep's three GEMMs, reduced together
    const size_t o_jobs = take(off, p->layers.size() * 9 * sizeof(GradJob));
    const size_t o_ljobs = take(off, p->layers.size() * 2 * sizeof(LogsJob));
    if (w && base) {
        w->col = at<float>(base, o_col); w->partial = at<float>(base, o_part); w->partial_floats = partf; w->jobs = at<GradJob>(base, o_jobs); w->ljobs = at<LogsJob>(base, o_ljobs);
        w->acc = at<unsigned long long>(base, o_acc); w->gld = at<float>(base, o_gld); w->gsum = at<double>(base, o_gsum);
        w->gA = at<float>(base, o_gA); w->gB = at<float>(base, o_gB); w->gh1 = at<float>(base, o_h1);
        w->gh2 = at<float>(base, o_h2); w->gpre = at<float>(base, o_gpre); w->wT = at<float>(base, o_wT);
        w->dacc = at<double>(base, o_dacc); w->dacc_doubles = nd; w->gsh = at<float>(base, o_gsh);
    }
    return align_up(off, 256);
}

// ---------------------------------------------------------------- forward with tape
static int forward_train(glowhip_plan* p, const void* packed, const float* x, const float* noise, float* z_out, int N,
                         char* tape, const std::vector<TapeLayer>& tl, unsigned long long* acc, float* sh_scratch,
                         hipStream_t s) {
    const float* cur = x;
    const int nl = (int)p->layers.size();
    const size_t scratch_floats = (size_t)N * p->max_hidden;
    bool premixed = false;      // this step's ActNorm + permutation output is already in its tape slot (the previous step's finishing kernel)
    p->tape_has_masks.assign(nl, 0);
    for (int li = 0; li < nl; ++li) {
        const LayerPlan& L = p->layers[li];
        const glowhip_layer_desc& d = L.d;
        p->cur_layer = li;
        float* dst = at<float>(tape, tl[li].out);
        const int HW = d.H * d.W, Ch = d.C / 2, hid = d.hidden;
        const long chw = (long)d.C * HW;
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) {
            GH_TRY(launch_squeeze(cur, noise, dst, N, d.C, d.H, d.W, 2, 0, s));
            noise = nullptr;
        } else if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
            GH_REQUIRE(noise == nullptr, "forward_train: the dequantisation noise needs a leading Squeeze2d layer");
            float* h1 = at<float>(tape, tl[li].h1);
            float* h2 = at<float>(tape, tl[li].h2);
            float* hout = at<float>(tape, tl[li].hout);
            if (!premixed) {
                ChanMixArgs m{};
                m.in_a = cur; m.in_a_bs = chw; m.in_b = cur + (long)Ch * HW; m.in_b_bs = chw; m.Ca = Ch;
                m.out = dst; m.out_bs = chw; m.bias = d.an_bias; m.scale = at<float>(packed, L.an_scale);
                m.matrix = d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr;
                m.gather = d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx : nullptr;
                m.reverse = 0; m.N = N; m.C = d.C; m.HW = HW;
                GH_TRY(launch_chanmix(m, s));
            }
            premixed = false;
            if (tape_cnet(p, L, N, scratch_floats)) {
                // the product path's two launches, taping: k_cnet stores h1 / h2, the finishing kernel hout, the step output
                // (y1, z2') in place and -- when the next layer is a FlowStep of the same shape -- that step's mixer output into
                // ITS tape slot
                CnetArgs c{};
                c.w0 = at<char>(packed, L.cn_w0); c.w2 = at<char>(packed, L.cn_w2); c.w4 = at<char>(packed, L.cn_w4);
                c.N = N; c.Cin = Ch; c.H = d.H; c.W = d.W; c.hidden = hid; c.Cout = L.Cout;
                c.scratch = sh_scratch;
                c.bias = d.f4_bias; c.scale = at<float>(packed, L.f4_scale);
                c.mode = d.coupling == GLOWHIP_COUPLING_AFFINE ? TAIL_AFFINE_FWD : TAIL_ADD_FWD;
                c.acc = acc;
                c.x = dst; c.x_bs = chw; c.z_in = dst; c.z_in_bs = chw;
                c.z_out = dst; c.z_out_bs = chw;
                c.tape_h1 = h1; c.tape_h2 = h2; c.tape_hout = hout;
                c.mask1 = at<unsigned short>(tape, tl[li].m1); c.mask2 = at<unsigned short>(tape, tl[li].m2);
                c.in_scale = SH2_ACT_SCALE; c.out_scale = SH2_ACT_INV;
                if (li + 1 < nl) {
                    const LayerPlan& Ln = p->layers[li + 1];
                    const glowhip_layer_desc& dn = Ln.d;
                    if (dn.kind == GLOWHIP_LAYER_FLOWSTEP && dn.C == d.C && dn.H == d.H && dn.W == d.W) {
                        c.mix = CnetMixer{dn.C, 0, dn.an_bias, at<float>(packed, Ln.an_scale),
                                          dn.permutation == GLOWHIP_PERM_INVCONV ? dn.invconv_w : nullptr,
                                          dn.permutation == GLOWHIP_PERM_GATHER ? dn.perm_idx : nullptr};
                        c.z_out = at<float>(tape, tl[li + 1].out);
                        premixed = true;
                    }
                }
                p->tape_has_masks[li] = 1;
                count_launch(p, "k_cnet(tape)");
                {
                    CnetPending pend{};
                    {
                        ScopedTimer t(p, GLOWHIP_K_CNET_TAPE, 1, s);
                        GH_TRY(launch_cnet_main(c, s, &pend));
                    }
                    if (pend.one_wave) count_launch(p, "k_cnet1w(tape)");      // (run-time evidence for the tests: the taping instance of cnet1w_sh.hip took it)
                    ScopedTimer t(p, GLOWHIP_K_CFINISH, 0, s);
                    GH_TRY(launch_cnet_finish(c, pend, s));
                }
                cur = dst;
                continue;
            }
            // f.0 (shapes the taping k_cnet does not take: the exact-fp32 kernels, layer by layer)
            if (L.first_halo) {
                const float* wf = at<float>(packed, L.f0_wt);
                GH_TRY(launch_conv_mfma_first(dst, chw, wf, wf + (size_t)9 * Ch * hid, h1, N, Ch, d.H, d.W, hid, s, 1));
            } else if (L.mfma_first) {
                GH_TRY(launch_conv_mfma_wide(dst, chw, at<float>(packed, L.f0_wt), d.f0_an_bias,
                                             at<float>(packed, L.f0_scale), h1, N, Ch, d.H, d.W, hid, 3, s));
            } else {
                ConvArgs c{dst, chw, d.f0_w, nullptr, d.f0_an_bias, nullptr, at<float>(packed, L.f0_scale), 1, h1,
                           N, Ch, d.H, d.W, hid, 3};
                GH_TRY(launch_conv_direct(c, s));
            }
            // f.2
            if (L.mfma_mid) {
                GH_TRY(launch_conv_mfma_wide(h1, (long)hid * HW, at<float>(packed, L.f2_wt), d.f2_an_bias,
                                             at<float>(packed, L.f2_scale), h2, N, hid, d.H, d.W, hid, 1, s));
            } else {
                ConvArgs c{h1, (long)hid * HW, d.f2_w, nullptr, d.f2_an_bias, nullptr, at<float>(packed, L.f2_scale), 1, h2,
                           N, hid, d.H, d.W, hid, 1};
                GH_TRY(launch_conv_direct(c, s));
            }
            // f.4 + coupling, hout kept
            float* z2 = dst + (long)Ch * HW;
            const int affine = d.coupling == GLOWHIP_COUPLING_AFFINE;
            if (L.mfma_last) {
                TailConvArgs t{};
                t.x = h2; t.x_bs = (long)hid * HW; t.wp = at<float>(packed, L.f4_wp); t.bias = d.f4_bias;
                t.scale = at<float>(packed, L.f4_scale); t.N = N; t.Cin = hid; t.H = d.H; t.W = d.W; t.Cout = L.Cout;
                t.mode = affine ? TAIL_AFFINE_FWD : TAIL_ADD_FWD;
                t.z2_in = z2; t.z2_in_bs = chw; t.z2_out = z2; t.z2_out_bs = chw; t.acc = acc;
                t.zeros = at<float>(packed, 64); t.hout = hout;
                GH_TRY(launch_conv_mfma_tail(t, s));
            } else {
                ConvArgs c{h2, (long)hid * HW, d.f4_w, d.f4_bias, nullptr, nullptr, at<float>(packed, L.f4_scale), 0, hout,
                           N, hid, d.H, d.W, L.Cout, 3};
                GH_TRY(launch_conv_direct(c, s));
                CouplingTailArgs t{hout, z2, chw, z2, chw, N, Ch, HW, affine, 0, acc};
                GH_TRY(launch_coupling_tail(t, s));
            }
        } else {  // SPLIT2D
            float* hout = at<float>(tape, tl[li].hout);
            if (L.mfma_last) {
                TailConvArgs t{};
                t.x = cur; t.x_bs = chw; t.wp = at<float>(packed, L.f4_wp); t.bias = d.f4_bias;
                t.scale = at<float>(packed, L.f4_scale); t.N = N; t.Cin = Ch; t.H = d.H; t.W = d.W; t.Cout = L.Cout;
                t.mode = TAIL_SPLIT_FWD; t.z2_in = cur + (long)Ch * HW; t.z2_in_bs = chw; t.z2_out = nullptr;
                t.z2_out_bs = 0; t.acc = acc; t.zeros = at<float>(packed, 64); t.hout = hout;
                GH_TRY(launch_conv_mfma_tail(t, s));
            } else {
                ConvArgs c{cur, chw, d.f4_w, d.f4_bias, nullptr, nullptr, at<float>(packed, L.f4_scale), 0, hout,
                           N, Ch, d.H, d.W, L.Cout, 3};
                GH_TRY(launch_conv_direct(c, s));
                SplitTailArgs t{hout, cur + (long)Ch * HW, chw, nullptr, nullptr, 0, N, Ch, HW, 0, acc};
                GH_TRY(launch_split_tail(t, s));
            }
            GH_TRY(launch_copy_strided(cur, chw, dst, (long)Ch * HW, N, (long)Ch * HW, s));
        }
        cur = dst;
    }
    const int* o = p->out_shape;
    if (z_out)
        GH_TRY(launch_copy_strided(cur, (long)o[0] * o[1] * o[2], z_out, (long)o[0] * o[1] * o[2], N,
                                   (long)o[0] * o[1] * o[2], s));
    return GLOWHIP_OK;
}

// input gradient of a 'SAME' convolution y = conv(x, w): g_x (+)= conv(g_y, flipT(w)); generic path
static int dgrad_direct(const float* gy, const float* w, float* wT, float* gx, int N, int Cin, int H, int W, int Cout,
                        int ksize, hipStream_t s) {
    GH_TRY(launch_weight_flipT(w, wT, Cout, Cin, ksize, s));
    ConvArgs c{gy, (long)Cout * H * W, wT, nullptr, nullptr, nullptr, nullptr, 0, gx, N, Cout, H, W, Cin, ksize};
    return launch_conv_direct(c, s);
}

__global__ void __launch_bounds__(256) k_add_inplace(float* __restrict__ dst, long dst_bs, const float* __restrict__ src,
                                                     long src_bs, long per) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = blockIdx.y;
    if (i < per) dst[n * dst_bs + i] += src[n * src_bs + i];
}

static int zero_f64(double* p, size_t n, hipStream_t s) {
    if (hipMemsetAsync(p, 0, n * sizeof(double), s) != hipSuccess) {
        set_error("backward: hipMemsetAsync failed");
        return GLOWHIP_ELAUNCH;
    }
    return GLOWHIP_OK;
}

// ---------------------------------------------------------------- backward sweep
static int backward_sweep(glowhip_plan* p, const void* packed, const float* x_in, const char* tape,
                          const std::vector<TapeLayer>& tl, const glowhip_layer_grads* grads, float* grad_x, int N,
                          TrainWs& w, float* g_top, hipStream_t s) {
    // g: gradient w.r.t. the current layer's OUTPUT (contiguous (N, C_out, H, W)), held in gA/gB
    float* g = g_top;
    // power-of-two pre-scale of the gradients that travel as fp16 pairs (backward k_cnet, weight-gradient GEMMs): 2^round(log2(B ln2 CHW)), the inverse
    // of dL/d(objective) for loss = mean(nll) (network/model.py:448-450, 496-506)
    const float sh_grad_scale = exp2f(rintf(log2f((float)N * 0.6931472f * (float)p->in_shape[0] * p->in_shape[1] * p->in_shape[2])));
    const int nl = (int)p->layers.size();
    std::vector<GradJob>& jobs = p->grad_jobs;
    jobs.clear();
    p->logs_jobs.clear();
    std::vector<size_t> acc_base(nl, 0);
    {
        size_t o = 0;
        for (int i = 0; i < nl; ++i) { acc_base[i] = o; o += layer_acc_doubles(p->layers[i]); }
        GH_TRY(zero_f64(w.dacc, o, s));
    }
    auto fin = [&](const double* acc, float* out, int n, double add_mul, const float* winv = nullptr, int C = 0, int copies = 1,
                   long stride = 0) {
        if (out) jobs.push_back(GradJob{acc, out, n, add_mul, winv, C, copies, stride});
    };
    // log-scale gradients of the layers whose input-gradient chain ran as one k_cnet launch: d logs = 3 (<W, dW> + b db) READS the
    // weight gradients, which live in the per-level buckets a data-parallel run all-reduces (in place, on a side stream) as soon as
    // the level's mark has passed.  So the pending jobs are launched BEFORE a mark is recorded (computed after the sweep
    // they could see dW half reduced or summed but not yet averaged); without marks (one GPU) they run as one launch at the end.
    p->logs_jobs.reserve((size_t)nl * 2);      // (the host table is the source of asynchronous copies: it must not move)
    size_t logs_done = 0;
    auto flush_logs = [&]() {
        const size_t n = p->logs_jobs.size() - logs_done;
        if (n == 0) return GLOWHIP_OK;
        if (hipMemcpyAsync(w.ljobs + logs_done, p->logs_jobs.data() + logs_done, n * sizeof(LogsJob), hipMemcpyHostToDevice, s) != hipSuccess) {
            set_error("backward: hipMemcpyAsync of the log-scale job table failed");
            return GLOWHIP_ELAUNCH;
        }
        GH_TRY(launch_logs_from_dw_batched(w.ljobs + logs_done, (int)n, s));
        logs_done = p->logs_jobs.size();
        return GLOWHIP_OK;
    };
    size_t mark_i = 0;      // gradient-ready marks (glowhip_plan_backward_marks), in sweep order
    auto marks_down_to = [&](int li) {
        while (mark_i < p->bwd_marks.size() && p->bwd_marks[mark_i].first >= li) {
            GH_TRY(flush_logs());
            if (hipEventRecord(p->bwd_marks[mark_i].second, s) != hipSuccess) { set_error("backward: hipEventRecord of a mark failed"); return GLOWHIP_ELAUNCH; }
            ++mark_i;
        }
        return GLOWHIP_OK;
    };
    for (int li = nl - 1; li >= 0; --li) {
        GH_TRY(marks_down_to(li + 1));      // every layer above li has been swept
        p->cur_layer = li;
        const LayerPlan& L = p->layers[li];
        const glowhip_layer_desc& d = L.d;
        const glowhip_layer_grads& G = grads[li];
        const float* xin = li > 0 ? at<float>(tape, tl[li - 1].out) : x_in;   // this layer's input
        float* gnext = (g == w.gA) ? w.gB : w.gA;
        const int HW = d.H * d.W, Ch = d.C / 2, hid = d.hidden;
        const long chw = (long)d.C * HW;
        if (d.kind == GLOWHIP_LAYER_SQUEEZE) {
            if (li == 0 && grad_x == nullptr) break;      // nobody wants dL/dx
            float* dst = li == 0 ? grad_x : gnext;
            GH_TRY(launch_squeeze(g, nullptr, dst, N, d.C * 4, d.H / 2, d.W / 2, 2, 1, s));
            g = dst;
        } else if (d.kind == GLOWHIP_LAYER_FLOWSTEP) {
            const float* out = at<float>(tape, tl[li].out);
            const float* h1 = at<float>(tape, tl[li].h1);
            const float* h2 = at<float>(tape, tl[li].h2);
            const float* hout = at<float>(tape, tl[li].hout);
            const int affine = d.coupling == GLOWHIP_COUPLING_AFFINE;
            // accumulators: [W C*C][an_b C][an_l C][f0_b hid][f0_l hid][f2_b hid][f2_l hid][f4_b Cout][f4_l Cout]
            const long mstride = (long)d.C * d.C + 2 * d.C;       // one copy of the mixer accumulators
            double* aW = w.dacc + acc_base[li]; double* aAb = aW + (size_t)d.C * d.C; double* aAl = aAb + d.C;
            double* a0b = aW + MIX_ACC_COPIES * mstride; double* a0l = a0b + hid; double* a2b = a0l + hid; double* a2l = a2b + hid;
            double* a4b = a2l + hid; double* a4l = a4b + L.Cout;
            // (a) coupling tail: g (second half) becomes g_y2 in place; gpre = gradient of f.4's (conv + bias)
            CouplingBwdArgs cb{hout, out + (long)Ch * HW, chw, g + (long)Ch * HW, chw, g + (long)Ch * HW, w.gpre,
                               at<float>(packed, L.f4_scale), w.gld, a4b, a4l, N, Ch, L.Cout, HW, affine};
            cb.acc_copies = MIX_ACC_COPIES; cb.acc_stride = 2 * L.Cout;
            GH_TRY(launch_coupling_bwd(cb, s));
            // (b) f.4: weight gradient, then input gradient -> g_h2 (raw), then ReLU/ActNorm of f.2
            const bool fastw = wgrad_fast(L);
            if (fastw && bwd_cnet(p, L, li, G, N, (size_t)N * p->max_hidden)) {
                // The input-gradient chain g_pre -> g_u2 -> g_u0 -> d y1 as ONE k_cnet launch (MODE 2, cnet_sh.hip) on the transposed
                // weight images, the ReLU masks from the tape's sign bits; the bias gradients are row sums inside the weight-gradient
                // GEMMs that read g_u2 / g_u0 anyway, the log-scale gradients follow from dW and db (backward.h LogsJob).
                const int m4 = round_up(L.Cout * 9, 128), n0 = round_up(Ch * 9, 64);
                // (the 3x3 layers' shift-expanded operands are gathered by the GEMM's loader: WgradTaps)
                const bool vtaps = d.W >= 4 && (d.W & (d.W - 1)) == 0;
                const WgradTaps t4{0, L.Cout, d.H, d.W, -1}, t0{1, Ch, d.H, d.W, +1};
                WgradReduceJobs rj{};      // the three split-K reductions of this step run as one launch at its end
                rj.n = 3;
                // f.4's GEMM does not depend on the chain below, but it shares a launch with f.2's and f.0's, which do
                // (launch_wgrad_trio); shapes that launch does not take run here, alone
                const bool pair = vtaps && wgrad_pair_ok(HW, m4, hid, n0);
                // short pixel axes (<= 512 k-tiles of 32 pixels): f.2's GEMM joins the launch as well -- at 12 - 18 pixel slices
                // instead of 32 its workgroups' loops are two to three times as long and a third of the partial tiles is left
                const bool trio = pair && (long)N * HW / 32 <= 512;
                if (pair) {      // (launched below, behind the chain)
                } else if (vtaps) {
                    { ScopedTimer tw(p, GLOWHIP_K_WGRAD, 1, s); GH_TRY(launch_wgrad_mfma(w.gpre, (long)L.Cout * HW, h2, (long)hid * HW, w.partial, G.f4_w, N, HW, m4, hid,
                                             L.Cout * 9, hid, 1, s, sh_grad_scale, nullptr, &t4, &rj.job[0], 0, 1, 2)); }
                } else {
                    GH_TRY(launch_shift_expand(w.gpre, (long)L.Cout * HW, w.col, N, L.Cout, d.H, d.W, m4, -1, s));
                    { ScopedTimer tw(p, GLOWHIP_K_WGRAD, 1, s); GH_TRY(launch_wgrad_mfma(w.col, (long)m4 * HW, h2, (long)hid * HW, w.partial, G.f4_w, N, HW, m4, hid,
                                             L.Cout * 9, hid, 1, s, sh_grad_scale, nullptr, nullptr, &rj.job[0], 0, 1, 2)); }
                }
                CnetArgs c{};
                c.w0 = at<char>(packed, L.cb_w0); c.w2 = at<char>(packed, L.cb_w2); c.w4 = at<char>(packed, L.cb_w4);
                c.N = N; c.Cin = L.Cout; c.H = d.H; c.W = d.W; c.hidden = hid; c.Cout = Ch;
                c.scratch = w.gsh; c.mode = TAIL_ADD_FWD;
                c.x = w.gpre; c.x_bs = (long)L.Cout * HW; c.z_in = w.gpre; c.z_in_bs = (long)L.Cout * HW;
                c.tape_h1 = w.gh2; c.tape_h2 = w.gh1;
                c.mask1 = const_cast<unsigned short*>(at<unsigned short>(tape, tl[li].m1));
                c.mask2 = const_cast<unsigned short*>(at<unsigned short>(tape, tl[li].m2));
                c.in_scale = SH2_ACT_SCALE * sh_grad_scale; c.out_scale = SH2_ACT_INV * SH_LO_SCALE; c.bwd = 1;      // (g_u2 / g_u0 are stored times sh_grad_scale * 2^11: what the weight-gradient GEMMs split with two instructions per value)
                CnetPending pend{};
                count_launch(p, "k_cnet(bwd)");
                {
                    ScopedTimer t(p, GLOWHIP_K_CNET_BWD, 1, s);
                    GH_TRY(launch_cnet_main(c, s, &pend));
                }
                if (pend.one_wave) count_launch(p, "k_cnet1w(bwd)");      // (run-time evidence for the tests: the backward instance of cnet1w_sh.hip took it)
                // (its finishing step -- g_y1 += the partial sums -- rides in k_chanmix_bwd below)
                if (trio) {      // all three GEMMs side by side in one launch
                    count_launch(p, "k_wgrad(trio)");
                    ScopedTimer tw(p, GLOWHIP_K_WGRAD, 1, s);
                    GH_TRY(launch_wgrad_trio(w.gh2, h1, w.partial + w.partial_floats, G.f2_w, a2b, w.gpre, (long)L.Cout * HW, h2, w.partial, G.f4_w, m4, L.Cout * 9,
                                             w.gh1, out, chw, w.partial + 2 * w.partial_floats, G.f0_w, n0, Ch * 9, N, HW, hid, sh_grad_scale, a0b, t4, t0,
                                             &rj.job[1], &rj.job[0], &rj.job[2], s));
                } else {
                    ScopedTimer tw(p, GLOWHIP_K_WGRAD, 1, s); GH_TRY(launch_wgrad_mfma(w.gh2, (long)hid * HW, h1, (long)hid * HW, w.partial + w.partial_floats, G.f2_w, N, HW, hid, hid,
                                         hid, hid, 0, s, sh_grad_scale, a2b, nullptr, &rj.job[1], 0, 1, 7));
                }
                if (trio) {
                } else if (pair) {
                    count_launch(p, "k_wgrad(pair)");
                    ScopedTimer tw(p, GLOWHIP_K_WGRAD, 1, s);
                    GH_TRY(launch_wgrad_pair(w.gpre, (long)L.Cout * HW, h2, w.partial, G.f4_w, m4, L.Cout * 9, w.gh1, out, chw, w.partial + 2 * w.partial_floats,
                                             G.f0_w, n0, Ch * 9, N, HW, hid, sh_grad_scale, a0b, t4, t0, 2, 5, &rj.job[0], &rj.job[2], s));
                } else if (vtaps) {
                    { ScopedTimer tw(p, GLOWHIP_K_WGRAD, 1, s); GH_TRY(launch_wgrad_mfma(w.gh1, (long)hid * HW, out, chw, w.partial + 2 * w.partial_floats, G.f0_w, N, HW, hid, n0,
                                             hid, Ch * 9, 0, s, sh_grad_scale, a0b, &t0, &rj.job[2], 0, 0, 5)); }
                } else {
                    GH_TRY(launch_shift_expand(out, chw, w.col, N, Ch, d.H, d.W, n0, +1, s));
                    { ScopedTimer tw(p, GLOWHIP_K_WGRAD, 1, s); GH_TRY(launch_wgrad_mfma(w.gh1, (long)hid * HW, w.col, (long)n0 * HW, w.partial + 2 * w.partial_floats, G.f0_w, N, HW,
                                             hid, n0, hid, Ch * 9, 0, s, sh_grad_scale, a0b, nullptr, &rj.job[2], 0, 0, 5)); }
                }
                // (the three reductions ride in the mixer backward's launch below: k_chanmix_bwd_reduce)
                if (G.f2_an_logs) p->logs_jobs.push_back(LogsJob{d.f2_w, G.f2_w, d.f2_an_bias, a2b, G.f2_an_logs, hid, hid});
                if (G.f0_an_logs) p->logs_jobs.push_back(LogsJob{d.f0_w, G.f0_w, d.f0_an_bias, a0b, G.f0_an_logs, hid, Ch * 9});
                ChanMixBwdArgs mb{xin, chw, g, g, chw, d.an_bias, at<float>(packed, L.an_scale),
                                  d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr,
                                  d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx_inv : nullptr, aW, aAb, aAl, N, d.C, HW};
                mb.acc_copies = MIX_ACC_COPIES; mb.acc_stride = mstride;
                mb.add_part = pend.scratch; mb.add_scale = 1.0f / sh_grad_scale; mb.add_C = Ch; mb.add_MS = pend.MS;
                mb.add_tiles = pend.tiles; mb.add_R = pend.R; mb.add_NI = pend.NI; mb.add_lpxt = pend.lpxt; mb.add_H = d.H; mb.add_W = d.W;
                GH_TRY(launch_chanmix_bwd(mb, s, &rj));
                if (d.permutation == GLOWHIP_PERM_INVCONV) fin(aW, G.invconv_w, d.C * d.C, (double)HW, at<float>(packed, L.winv), d.C, MIX_ACC_COPIES, mstride);
                fin(aAb, G.an_bias, d.C, 0.0, nullptr, 0, MIX_ACC_COPIES, mstride);
                fin(aAl, G.an_logs, d.C, 3.0 * HW, nullptr, 0, MIX_ACC_COPIES, mstride);
                fin(a0b, G.f0_an_bias, hid, 0.0);
                fin(a2b, G.f2_an_bias, hid, 0.0);
                fin(a4b, G.f4_bias, L.Cout, 0.0, nullptr, 0, MIX_ACC_COPIES, 2 * L.Cout); fin(a4l, G.f4_logs, L.Cout, 0.0, nullptr, 0, MIX_ACC_COPIES, 2 * L.Cout);
                continue;
            }
            // The per-layer kernels below read fp32: a tape written by the taping k_cnet holds h1 / h2 as fp16 -- converted into the
            // partial-sum scratch (unused on this path), h2 first, h1 once h2 has been consumed.
            const bool half_tape = li < (int)p->tape_has_masks.size() && p->tape_has_masks[li];
            if (half_tape) {
                GH_TRY(launch_half_to_float(h2, w.gsh, N, hid, HW, s));
                h2 = w.gsh;
            }
            if (fastw) {   // dW4[o][i][tap] = sum_p g_pre[o][p - d(tap)] * h2[i][p]
                const int m4 = round_up(L.Cout * 9, 128);
                GH_TRY(launch_shift_expand(w.gpre, (long)L.Cout * HW, w.col, N, L.Cout, d.H, d.W, m4, -1, s));
                GH_TRY(launch_wgrad_mfma(w.col, (long)m4 * HW, h2, (long)hid * HW, w.partial, G.f4_w, N, HW, m4, hid,
                                         L.Cout * 9, hid, 1, s, train_sh_enabled(p) ? sh_grad_scale : 0.f));
            } else {
                GH_TRY(launch_wgrad_direct(w.gpre, h2, (long)hid * HW, G.f4_w, N, hid, d.H, d.W, L.Cout, 3, s));
            }
            if (L.dg4_first) {
                const float* wf = at<float>(packed, L.f4T_wf);
                GH_TRY(launch_conv_mfma_first(w.gpre, (long)L.Cout * HW, wf, wf + (size_t)9 * L.Cout * hid, w.gh2, N, L.Cout,
                                              d.H, d.W, hid, s, 0));
            } else {
                GH_TRY(dgrad_direct(w.gpre, d.f4_w, w.wT, w.gh2, N, hid, d.H, d.W, L.Cout, 3, s));
            }
            GH_TRY(launch_act_bwd(w.gh2, h2, at<float>(packed, L.f2_scale), N, hid, HW, a2b, a2l, s));
            if (half_tape) {
                GH_TRY(launch_half_to_float(h1, w.gsh, N, hid, HW, s));
                h1 = w.gsh;
            }
            // (c) f.2 (1x1)
            if (fastw) {
                GH_TRY(launch_wgrad_mfma(w.gh2, (long)hid * HW, h1, (long)hid * HW, w.partial, G.f2_w, N, HW, hid, hid, hid, hid,
                                         0, s, train_sh_enabled(p) ? sh_grad_scale : 0.f));
            } else {
                GH_TRY(launch_wgrad_direct(w.gh2, h1, (long)hid * HW, G.f2_w, N, hid, d.H, d.W, hid, 1, s));
            }
            if (L.mfma_mid) {   // W2 in its reference layout [o][i] is already the K-major image of the transposed GEMM
                GH_TRY(launch_conv_mfma_wide(w.gh2, (long)hid * HW, d.f2_w, nullptr, nullptr, w.gh1, N, hid, d.H, d.W, hid, 1,
                                             s, 0));
            } else {
                GH_TRY(dgrad_direct(w.gh2, d.f2_w, w.wT, w.gh1, N, hid, d.H, d.W, hid, 1, s));
            }
            GH_TRY(launch_act_bwd(w.gh1, h1, at<float>(packed, L.f0_scale), N, hid, HW, a0b, a0l, s));
            // (d) f.0: input is y1 = first half of the step output
            if (fastw) {   // dW0[o][i][tap] = sum_p g_u0[o][p] * y1[i][p + d(tap)]
                const int n0 = round_up(Ch * 9, 64);
                GH_TRY(launch_shift_expand(out, chw, w.col, N, Ch, d.H, d.W, n0, +1, s));
                GH_TRY(launch_wgrad_mfma(w.gh1, (long)hid * HW, w.col, (long)n0 * HW, w.partial, G.f0_w, N, HW, hid, n0, hid,
                                         Ch * 9, 0, s, train_sh_enabled(p) ? sh_grad_scale : 0.f));
            } else {
                GH_TRY(launch_wgrad_direct(w.gh1, out, chw, G.f0_w, N, Ch, d.H, d.W, hid, 3, s));
            }
            if (L.dg0_tail) {   // g_y1 += conv(g_u0, flipT(W0)) fused: the tail kernel's additive-coupling epilogue
                TailConvArgs t{};
                t.x = w.gh1; t.x_bs = (long)hid * HW; t.wp = at<float>(packed, L.f0T_wp); t.bias = nullptr; t.scale = nullptr;
                t.N = N; t.Cin = hid; t.H = d.H; t.W = d.W; t.Cout = Ch; t.mode = TAIL_ADD_FWD;
                t.z2_in = g; t.z2_in_bs = chw; t.z2_out = g; t.z2_out_bs = chw; t.acc = nullptr;
                t.zeros = at<float>(packed, 64); t.hout = nullptr;
                GH_TRY(launch_conv_mfma_tail(t, s));
            } else {
                GH_TRY(dgrad_direct(w.gh1, d.f0_w, w.wT, w.gpre, N, Ch, d.H, d.W, hid, 3, s));   // gpre reused: (N,Ch,HW)
                hipLaunchKernelGGL(k_add_inplace, dim3(cdiv((long)Ch * HW, 256), N), dim3(256), 0, s, g, chw, w.gpre,
                                   (long)Ch * HW, (long)Ch * HW);
                GH_LAUNCH_CHECK("k_add_inplace");
            }
            // (e) ActNorm + invconv / permutation: g (= g_y) -> g_x in place
            ChanMixBwdArgs mb{xin, chw, g, g, chw, d.an_bias, at<float>(packed, L.an_scale),
                              d.permutation == GLOWHIP_PERM_INVCONV ? d.invconv_w : nullptr,
                              d.permutation == GLOWHIP_PERM_GATHER ? d.perm_idx_inv : nullptr, aW, aAb, aAl, N, d.C, HW};
            mb.acc_copies = MIX_ACC_COPIES; mb.acc_stride = mstride;
                GH_TRY(launch_chanmix_bwd(mb, s));
            // (f) fp64 accumulators -> fp32 gradients (+ the log-det terms that do not depend on the data): queued,
            // converted by ONE launch after the sweep
            if (d.permutation == GLOWHIP_PERM_INVCONV) fin(aW, G.invconv_w, d.C * d.C, (double)HW, at<float>(packed, L.winv), d.C, MIX_ACC_COPIES, mstride);
            fin(aAb, G.an_bias, d.C, 0.0, nullptr, 0, MIX_ACC_COPIES, mstride);
            fin(aAl, G.an_logs, d.C, 3.0 * HW, nullptr, 0, MIX_ACC_COPIES, mstride);
            fin(a0b, G.f0_an_bias, hid, 0.0); fin(a0l, G.f0_an_logs, hid, 0.0);
            fin(a2b, G.f2_an_bias, hid, 0.0); fin(a2l, G.f2_an_logs, hid, 0.0);
            fin(a4b, G.f4_bias, L.Cout, 0.0, nullptr, 0, MIX_ACC_COPIES, 2 * L.Cout); fin(a4l, G.f4_logs, L.Cout, 0.0, nullptr, 0, MIX_ACC_COPIES, 2 * L.Cout);
        } else {  // SPLIT2D: output z1 (N,Ch,HW); input x = (z1, z2)
            const float* hout = at<float>(tape, tl[li].hout);
            double* a4b = w.dacc + acc_base[li]; double* a4l = a4b + L.Cout;
            // g_x first half <- g (gradient of z1), second half <- gradient of the log-density
            GH_TRY(launch_copy_strided(g, (long)Ch * HW, gnext, chw, N, (long)Ch * HW, s));
            SplitBwdArgs sb{hout, xin + (long)Ch * HW, chw, gnext + (long)Ch * HW, chw, w.gpre,
                            at<float>(packed, L.f4_scale), w.gld, a4b, a4l, N, Ch, HW};
            GH_TRY(launch_split_bwd(sb, s));
            {   // Conv2dZeros weight gradient: the split-K GEMM of the FlowSteps (A = gathered taps of g_pre, B = z1's Ch rows)
                const int m4 = round_up(L.Cout * 9, 128);
                const bool mf = train_sh_enabled(p) && G.f4_w && HW % 32 == 0 && Ch <= 64 && d.W >= 4 && (d.W & (d.W - 1)) == 0 &&
                                wgrad_mfma_partial_floats(m4, 64, N, HW) <= w.partial_floats;
                if (mf) {
                    const WgradTaps t4{0, L.Cout, d.H, d.W, -1};
                    GH_TRY(launch_wgrad_mfma(w.gpre, (long)L.Cout * HW, xin, chw, w.partial, G.f4_w, N, HW, m4, 64, L.Cout * 9, Ch, 1, s,
                                             sh_grad_scale, nullptr, &t4, nullptr, Ch));
                } else {
                    GH_TRY(launch_wgrad_direct(w.gpre, xin, chw, G.f4_w, N, Ch, d.H, d.W, L.Cout, 3, s));
                }
            }
            GH_TRY(dgrad_direct(w.gpre, d.f4_w, w.wT, w.gh1, N, Ch, d.H, d.W, L.Cout, 3, s));     // (N,Ch,HW)
            hipLaunchKernelGGL(k_add_inplace, dim3(cdiv((long)Ch * HW, 256), N), dim3(256), 0, s, gnext, chw, w.gh1,
                               (long)Ch * HW, (long)Ch * HW);
            GH_LAUNCH_CHECK("k_add_inplace");
            fin(a4b, G.f4_bias, L.Cout, 0.0); fin(a4l, G.f4_logs, L.Cout, 0.0);
            g = gnext;
        }
    }
    GH_TRY(marks_down_to(0));
    if (!jobs.empty()) {
        if (hipMemcpyAsync(w.jobs, jobs.data(), jobs.size() * sizeof(GradJob), hipMemcpyHostToDevice, s) != hipSuccess) {
            set_error("backward: hipMemcpyAsync of the finalize job table failed");
            return GLOWHIP_ELAUNCH;
        }
        GH_TRY(launch_grad_finalize_batched(w.jobs, (int)jobs.size(), w.gsum, s));
    }
    GH_TRY(flush_logs());
    return GLOWHIP_OK;
}

}  // namespace glowhip

// ================================================================================================ C ABI
extern "C" {

size_t glowhip_plan_tape_bytes(const glowhip_plan* plan, int N) {
    return (plan && N >= 0) ? tape_layout(plan, N, nullptr) : 0;
}

size_t glowhip_plan_train_workspace_bytes(const glowhip_plan* plan, int N) {
    return (plan && N >= 0) ? train_ws_layout(plan, N, nullptr, nullptr) : 0;
}

int glowhip_glow_forward_train(glowhip_plan* plan, const void* packed, const float* x, const float* noise,
                               const float* prior_mean, const float* prior_logs, long prior_stride, int n_bits,
                               float* z, float* nll_out, float* objective_out, int N, void* tape, size_t tape_bytes,
                               void* workspace, size_t workspace_bytes, glowhip_stream_t stream) {
    GH_REQUIRE(plan && packed && x && z && nll_out && tape && workspace, "glow_forward_train: null argument");
    GH_REQUIRE(N > 0 && N <= 65535, "glow_forward_train: batch size %d out of range", N);
    std::vector<TapeLayer> tl;
    GH_REQUIRE(tape_bytes >= tape_layout(plan, N, &tl), "glow_forward_train: tape too small");
    TrainWs w;
    GH_REQUIRE(workspace_bytes >= train_ws_layout(plan, N, workspace, &w), "glow_forward_train: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    GH_TRY(launch_zero_acc(w.acc, N, s, ACC_EXTRA));
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));      // the training kernels read the fp32 images from the first layer on
    GH_TRY(forward_train(plan, packed, x, noise, z, N, (char*)tape, tl, w.acc, w.gsh, s));
    const int* o = plan->out_shape;
    GH_TRY(launch_gaussian_logp(z, (long)o[0] * o[1] * o[2], prior_mean, prior_logs, prior_stride, N, o[0], o[1] * o[2],
                                w.acc, s));
    const double chw = (double)plan->in_shape[0] * plan->in_shape[1] * plan->in_shape[2];
    const double offset = -log(pow(2.0, n_bits)) * chw;
    const double scale = -1.0 / (log(2.0) * chw);
    return launch_finalize(nullptr, w.acc, at<double>(packed, 0), 1.0, offset, scale, nll_out, objective_out, N, s, ACC_EXTRA);
}

int glowhip_plan_backward_marks(glowhip_plan* plan, const int32_t* after_layer, void* const* events, int n) {
    GH_REQUIRE(plan && n >= 0 && (n == 0 || (after_layer && events)), "plan_backward_marks: bad argument");
    plan->bwd_marks.clear();
    for (int i = 0; i < n; ++i) {
        GH_REQUIRE(events[i] != nullptr, "plan_backward_marks: null event %d", i);
        GH_REQUIRE(after_layer[i] >= 0 && after_layer[i] < (int)plan->layers.size(), "plan_backward_marks: layer %d out of range", after_layer[i]);
        GH_REQUIRE(i == 0 || after_layer[i] < after_layer[i - 1], "plan_backward_marks: layer indices must decrease (sweep order)");
        plan->bwd_marks.emplace_back(after_layer[i], (hipEvent_t)events[i]);
    }
    return GLOWHIP_OK;
}

int glowhip_glow_backward(glowhip_plan* plan, const void* packed, const float* x, const void* tape, size_t tape_bytes,
                          const float* nll_grad, const float* z_grad, const float* prior_mean, const float* prior_logs,
                          long prior_stride, const glowhip_layer_grads* grads, float* grad_x, int N, void* workspace,
                          size_t workspace_bytes, glowhip_stream_t stream) {
    GH_REQUIRE(plan && packed && x && tape && nll_grad && grads && workspace, "glow_backward: null argument");
    GH_REQUIRE(N > 0 && N <= 65535, "glow_backward: batch size %d out of range", N);
    std::vector<TapeLayer> tl;
    GH_REQUIRE(tape_bytes >= tape_layout(plan, N, &tl), "glow_backward: tape too small");
    TrainWs w;
    GH_REQUIRE(workspace_bytes >= train_ws_layout(plan, N, workspace, &w), "glow_backward: workspace too small");
    for (const LayerPlan& L : plan->layers)
        GH_REQUIRE(L.d.kind != GLOWHIP_LAYER_FLOWSTEP || L.d.C <= 192, "glow_backward: C=%d not supported yet", L.d.C);
    hipStream_t s = (hipStream_t)stream;
    GH_TRY(join_legacy(plan, s)); GH_TRY(join_lu(plan, s));
    const double chw = (double)plan->in_shape[0] * plan->in_shape[1] * plan->in_shape[2];
    GH_TRY(launch_gld_from_nll(nll_grad, w.gld, N, 1.0 / (log(2.0) * chw), s));
    GH_TRY(launch_sum_gld(w.gld, N, w.gsum, s));
    // top: dL/dz = z_grad + gld * d logp/dz
    const int* o = plan->out_shape;
    const long per = (long)o[0] * o[1] * o[2];
    const float* zt = at<float>(tape, tl.back().out);
    GH_TRY(launch_prior_bwd(zt, prior_mean, prior_logs, prior_stride, w.gld, z_grad, w.gA, N, per, s));
    return backward_sweep(plan, packed, x, (const char*)tape, tl, grads, grad_x, N, w, w.gA, s);
}

}  // extern "C"
