// plan_internal.h -- plan data structures shared by plan.hip (inference executor) and plan_train.hip
// (training forward with a tape + backward executor).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "conv_mfma.h"
#include "sh.h"
#include "backward.h"

namespace glowhip {

struct LayerPlan {
    glowhip_layer_desc d;
    int Cout = 0;  // output channels of f.4 / of the Split2d prior conv
    // byte offsets into the packed buffer
    size_t an_scale = 0, an_inv_scale = 0, winv = 0, logabsdet = 0, konst = 0, lu_scratch = 0;
    size_t f0_scale = 0, f2_scale = 0, f4_scale = 0;
    size_t f0_wt = 0, f2_wt = 0, f4_wp = 0;
    bool mfma_first = false, mfma_mid = false, mfma_last = false;
    // training: input-gradient convolutions run on the forward kernels with flipped/transposed weight images
    size_t f4T_wf = 0, f0T_wp = 0;
    bool dg4_first = false, dg0_tail = false;
    bool cnet = false; size_t cn_w0 = 0, cn_w2 = 0, cn_w4 = 0;   // whole coupling network as one kernel (cnet_sh.hip), SH2 images
    // the input-gradient chain as one k_cnet launch (MODE 2): SH2 images of the transposed weights (cb_w0: f.4^T as the 3x3 first
    // layer, cb_w2: f.2^T, cb_w4: f.0^T as the 3x3 last layer) and the transposed fp32 copies they are built from (wt4 / wt2 / wt0)
    bool cnet_bwd = false; size_t cb_w0 = 0, cb_w2 = 0, cb_w4 = 0, wt4 = 0, wt2 = 0, wt0 = 0;
    // deep levels (dnet_sh.hip): one launch per layer on SH2 images -- the mixer W (and W^-1, built after the LU), f.0 and f.4 as
    // (chunk, tap) implicit-GEMM images, f.2 as a plain GEMM image
    bool dnet = false; size_t dn_mix = 0, dn_mixinv = 0, dn_w0 = 0, dn_w2 = 0, dn_w4 = 0;
    bool wide_last = false; size_t f4_wt = 0;   // f.4 on k_conv_wide<3> (+ separate coupling tail): levels no tail kernel takes (4x4 pixels)
    bool first_halo = false;  // f.0 on k_conv_first (stationary pixel window) instead of k_conv_wide<3>
    size_t f0_init = 0;       // data-dependent init pass: PLAIN K-major image of f.0 for k_conv_wide<3> (the k_conv_first image has
                              // the ActNorm being initialised folded in); 0 = none (the direct kernel runs)
};

}  // namespace glowhip

using namespace glowhip;

struct TimingSlot { int kind, layer, mfma; hipEvent_t a, b; };

struct glowhip_plan {
    std::vector<LayerPlan> layers;
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;     // unused events
    std::vector<TimingSlot> ev_used;     // recorded, not yet read
    int cur_layer = 0;
    // batched pack: host copies of the job tables + their byte offsets inside `packed`
    std::vector<StepPrepJob> prep_jobs;
    std::vector<ScaleJob> scale_jobs;
    std::vector<RepackJob> repack_jobs;
    std::vector<FlipJob> flip_jobs; int flip_tiles[3] = {1, 1, 1}; size_t flip_off = 0;      // (tiles: max per member of a (f.4, f.2, f.0) triple)
    std::vector<RepackJob> repack_sel;    // the subset selected by the last glowhip_plan_pack_for (kept alive for the async copy)
    size_t prep_off = 0, scale_off = 0, repack_off = 0;
    // glowhip_plan_pack forks onto a side stream what the first kernels of a forward do not wait for: the round-1 / fp32 weight images
    // (read by the deep levels and the Split2d priors) and the LU factorisations (log|det W| enters only the final sum; W^-1 is
    // read by decode / backward).  Consumers join through these events (join_legacy / join_lu).
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_legacy = nullptr, ev_lu = nullptr;
    bool legacy_pending = false, lu_pending = false;
    const void* tables_in = nullptr;      // the `packed` buffer that already holds the job tables (glowhip_plan_forget_packed resets)
    unsigned slots_in = 0;                // ... and which of its repack-table slots (repack_slot) have been filled
    std::vector<RepackJob> repack_slot_host[32];
    bool pending_captured = false;        // ev_legacy / ev_lu were last recorded inside a stream capture (see join_legacy)
    int max_lds_c = 0, max_c = 0;
    size_t packed_bytes = 0;
    int in_shape[3] = {0, 0, 0}, out_shape[3] = {0, 0, 0};
    long max_chw = 0;      // max over layer inputs/outputs of C*H*W
    long max_hidden = 0;   // max over steps of max(hidden, Cout) * H*W
    int n_split = 0;
    std::vector<char> tape_has_masks;          // per layer: the last glow_forward_train stored the ReLU sign bits (k_cnet MODE 1)
    std::vector<glowhip::LogsJob> logs_jobs;   // the same for the log-scale gradients derived from dW / db (backward.h)
    std::vector<glowhip::GradJob> grad_jobs;   // host copy of the last backward's finalize table (kept alive for the async copy)
    bool rng_on = false; unsigned long long rng_seed = 0, rng_calls = 0;   // in-kernel dequantisation noise (glowhip_plan_set_dequant_rng)
    std::vector<std::pair<int, hipEvent_t>> bwd_marks;   // (layer index, event): glowhip_plan_backward_marks
    int family = GLOWHIP_FAMILY_AUTO;          // kernel family of the coupling networks (glowhip_plan_set_family): a property of the plan
    std::map<std::string, long> launch_counts; // run-time record of which kernel families this plan launched (glowhip_plan_launch_counts)
};

namespace glowhip {

static inline void count_launch(glowhip_plan* p, const char* name) { if (p) ++p->launch_counts[name]; }

// ---------------------------------------------------------------- optional per-launch timing
struct ScopedTimer {
    glowhip_plan* p; hipStream_t s; TimingSlot slot; bool on;
    ScopedTimer(glowhip_plan* plan, int kind, int mfma, hipStream_t st) : p(plan), s(st), on(plan && plan->timing) {
        if (!on) return;
        auto get = [&]() { hipEvent_t e; if (!p->ev_pool.empty()) { e = p->ev_pool.back(); p->ev_pool.pop_back(); }
                           else (void)hipEventCreate(&e); return e; };
        slot.kind = kind; slot.layer = p->cur_layer; slot.mfma = mfma; slot.a = get(); slot.b = get();
        (void)hipEventRecord(slot.a, s);
    }
    ~ScopedTimer() { if (on) { (void)hipEventRecord(slot.b, s); p->ev_used.push_back(slot); } }
};


constexpr int REPACK_SLOTS = 32;
// slot of a use mask's selected repack jobs: the five bits that select images (1 inference, 2 training, 4 inverse: the W^-1 images
// of the deep levels, 8 round-1 images of cnet / dnet layers, 16 init pass' f.0 image); 32 (no LU) does not change the table
static inline int repack_slot(int use) { return use & 31; }

static inline bool stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}
// An event recorded inside a stream capture belongs to that capture: waiting for it from a stream that is not capturing fails with
// hipErrorCapturedEvent -- and there is nothing to wait for: the capture joined its own fork, and replays order themselves as graph
// nodes.  So pending flags set during a capture are dropped by the first join that comes from outside one.
static inline bool drop_captured_pending(glowhip_plan* p, hipStream_t s) {
    if (!p->pending_captured || stream_capturing(s)) return false;
    p->legacy_pending = p->lu_pending = p->pending_captured = false;
    return true;
}
// make stream s wait for the side-stream part of the last pack (no-ops when nothing is pending)
static inline int join_legacy(glowhip_plan* p, hipStream_t s) {
    if (p->legacy_pending && !drop_captured_pending(p, s) && hipStreamWaitEvent(s, p->ev_legacy, 0) != hipSuccess) { set_error("join_legacy: hipStreamWaitEvent failed"); return GLOWHIP_ELAUNCH; }
    return GLOWHIP_OK;
}
static inline int join_lu(glowhip_plan* p, hipStream_t s) {
    if (p->lu_pending && !drop_captured_pending(p, s) && hipStreamWaitEvent(s, p->ev_lu, 0) != hipSuccess) { set_error("join_lu: hipStreamWaitEvent failed"); return GLOWHIP_ELAUNCH; }
    return GLOWHIP_OK;
}

static inline size_t take(size_t& off, size_t bytes) {
    size_t o = align_up(off, 256);
    off = o + bytes;
    return o;
}

template <typename T>
static inline T* at(const void* base, size_t off) {
    return (T*)((char*)base + off);
}

}  // namespace glowhip
