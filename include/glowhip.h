/*
 * glowhip.h -- C ABI of the MI355X-native Glow flow engine (libglowhip.so, gfx950).
 *
 * The reference (corenel/pytorch-glow) has no FFI: its hot path is the Python module surface of
 * network/module.py, network/model.py and misc/ops.py.  Each entry point below names the reference
 * function (file:line) whose arithmetic it replaces; the Python shells in
 * pytorch-glow_amd/network/{module,model}.py bind them with ctypes.
 *
 * Conventions
 *   - plain C: device pointers, sizes and an opaque stream handle (a hipStream_t passed as void*);
 *     no torch / C++ types cross the boundary;
 *   - every tensor is fp32, NCHW, contiguous unless a stride argument says otherwise; every pointer is
 *     a DEVICE pointer on the device the stream belongs to (exceptions are marked HOST);
 *   - no allocation, no host synchronisation, no global mutable state inside: workspaces are passed in;
 *     calls are asynchronous on `stream` and re-entrant (one plan must not be executed concurrently on
 *     two streams with the same workspace);
 *   - return value: 0 on success, a negative GLOWHIP_E* code otherwise; glowhip_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 */
#ifndef GLOWHIP_H
#define GLOWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLOWHIP_VERSION 102 /* 0.1.2 */

#define GLOWHIP_OK 0
#define GLOWHIP_EINVAL (-1)    /* bad argument (shape, null pointer, unsupported value) */
#define GLOWHIP_ELAUNCH (-2)   /* HIP launch / runtime error */
#define GLOWHIP_EWORKSPACE (-3)/* workspace too small */

typedef void* glowhip_stream_t; /* hipStream_t */

int glowhip_version(void);
const char* glowhip_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Single-layer primitives (module surface of network/module.py)
 * ---------------------------------------------------------------------------------------------- */

/* Squeeze2d.squeeze / unsqueeze, network/module.py:551-592.
 * reverse=0: x (N,C,H,W) -> y (N,C*f*f,H/f,W/f), y[n, c*f*f+i*f+j, h, w] = x[n, c, h*f+i, w*f+j]
 * reverse=1: x (N,C,H,W) -> y (N,C/(f*f),H*f,W*f) (exact inverse).  C,H,W describe the INPUT. */
int glowhip_squeeze2d(const float* x, float* y, int N, int C, int H, int W, int factor, int reverse,
                      glowhip_stream_t stream);

/* ActNorm data-dependent initialisation, network/module.py:86-120 (batch_variance=False):
 * bias[c] = -mean_{n,h,w} x;  logs[c] = log(scale / (sqrt(mean (x+bias)^2) + 1e-6)) / 3.
 * x may be a channel slice of a wider tensor: element (n,c,p) is x[n*batch_stride + c*HW + p]. */
int glowhip_actnorm_init(const float* x, long batch_stride, int N, int C, int HW, float scale,
                         float* bias, float* logs, glowhip_stream_t stream);
/* The same with ActNorm(batch_variance=True), network/module.py:109-110: bias per channel as above, but ONE log-scale for all
 * channels from the second moment pooled over every dimension: logs[c] = log(scale / (sqrt(mean_{n,c,h,w} (x+bias)^2) + 1e-6)) / 3. */
int glowhip_actnorm_init_batch_variance(const float* x, long batch_stride, int N, int C, int HW, float scale,
                                        float* bias, float* logs, glowhip_stream_t stream);

/* ActNorm.forward, network/module.py:122-149.  reverse=0: y=(x+bias)*exp(3 logs); reverse=1:
 * y = x*exp(-3 logs) - bias.  If logdet_out != NULL: logdet_out[n] = (logdet_in ? logdet_in[n] : 0)
 * +/- 3*sum(logs)*HW. */
int glowhip_actnorm(const float* x, float* y, const float* bias, const float* logs, int N, int C, int HW,
                    int reverse, const float* logdet_in, float* logdet_out, glowhip_stream_t stream);

/* In-kernel LU (Gauss-Jordan, partial pivoting, fp64) of one C x C matrix: replaces torch.det /
 * Tensor.inverse at network/module.py:357,365.  winv (C*C, may be NULL) <- W^-1,
 * logabsdet (1 float) <- log|det W|.  scratch: >= glowhip_invconv_scratch_bytes(C) bytes. */
size_t glowhip_invconv_scratch_bytes(int C);
int glowhip_invconv_prepare(const float* w, int C, float* winv, float* logabsdet, void* scratch,
                            glowhip_stream_t stream);

/* Invertible1x1Conv.forward, network/module.py:344-369: y[n,o,p] = sum_i m[o,i] x[n,i,p] where m is the
 * matrix to APPLY (W forward, W^-1 reverse, from glowhip_invconv_prepare).  If logdet_out != NULL:
 * logdet_out[n] = logdet_in[n] + sign * logabsdet[0] * HW with sign=+1 (reverse=0) / -1 (reverse=1). */
int glowhip_invconv(const float* x, float* y, const float* m, const float* logabsdet, int N, int C, int HW,
                    int reverse, const float* logdet_in, float* logdet_out, glowhip_stream_t stream);

/* Permutation2d.forward, network/module.py:392-397: y[:, o] = x[:, idx[o]] (idx: C int32, device). */
int glowhip_permute_channels(const float* x, float* y, const int32_t* idx, int N, int C, int HW,
                             glowhip_stream_t stream);

/* Conv2d / Conv2dZeros forward, network/module.py:252-259 and :295-297, 'SAME' padding, stride 1,
 * ksize in {1,3}:  v = conv(x, w) + (bias ? bias[o] : 0);
 *                  v = (v + (post_bias ? post_bias[o] : 0)) * (post_logs ? exp(3*post_logs[o]) : 1);
 *                  y = relu ? max(v,0) : v.
 * Conv2d(+ActNorm): bias=NULL, post_bias/post_logs = actnorm.bias/logs.  Conv2dZeros: bias, post_logs=logs.
 * x element (n,ci,p) at x[n*x_batch_stride + ci*H*W + p] (lets z1 = first half of a wider tensor be read
 * in place); y contiguous (N,Cout,H,W); w (Cout,Cin,k,k). */
int glowhip_conv2d(const float* x, long x_batch_stride, const float* w, const float* bias, float* y,
                   int N, int Cin, int H, int W, int Cout, int ksize,
                   const float* post_bias, const float* post_logs, int relu, glowhip_stream_t stream);

/* GaussianDiag.logp, network/module.py:453-467: out[n] = (in ? in[n] : 0) +
 * sum_{c,p} -0.5*(log(2pi) + 2*logs + (x-mean)^2/exp(2*logs)).  mean/logs may be NULL (= 0).
 * Element (n,c,p) of x/mean/logs at base[n*stride + c*HW + p].  scratch16N: 16*N bytes (per-sample fixed-point accumulator +
 * sticky non-finite flag: a NaN / inf term makes out[n] NaN, it is never dropped). */
int glowhip_gaussian_logp(const float* x, long x_stride, const float* mean, const float* logs, long ml_stride,
                          int N, int C, int HW, const float* in, float* out, void* scratch16N,
                          glowhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Flow plans: FlowStep / Split2d / Squeeze2d stacks executed by one call
 * (FlowStep.normal_flow/reverse_flow network/model.py:82-154, Split2d.forward network/module.py:511-536,
 *  FlowModel.encode/decode network/model.py:263-294, Glow.normal_flow network/model.py:409-452)
 * ---------------------------------------------------------------------------------------------- */
enum { GLOWHIP_LAYER_SQUEEZE = 0, GLOWHIP_LAYER_FLOWSTEP = 1, GLOWHIP_LAYER_SPLIT2D = 2 };
enum { GLOWHIP_PERM_INVCONV = 0, GLOWHIP_PERM_GATHER = 1 };   /* 'invconv' | 'reverse'/'shuffle' */
enum { GLOWHIP_COUPLING_ADDITIVE = 0, GLOWHIP_COUPLING_AFFINE = 1 };

/* One layer of a FlowModel; C,H,W are the layer's INPUT shape.  Parameter pointers are device pointers
 * to the live parameters (state_dict layout of the reference).  They are read by
 * glowhip_plan_pack (derived data) AND by encode/decode (biases, weights of the direct kernels), so
 * they must stay valid -- same addresses -- for the life of the plan. */
typedef struct glowhip_layer_desc {
    int32_t kind;
    int32_t C, H, W;
    int32_t hidden;        /* FLOWSTEP: hidden_channels */
    int32_t permutation;   /* FLOWSTEP: GLOWHIP_PERM_* */
    int32_t coupling;      /* FLOWSTEP: GLOWHIP_COUPLING_* */
    int32_t reserved;
    const float* an_bias;  const float* an_logs;            /* actnorm.{bias,logs}        (C) */
    const float* invconv_w;                                 /* invconv.weight             (C,C) */
    const int32_t* perm_idx; const int32_t* perm_idx_inv;   /* Permutation2d tables       (C) */
    const float* f0_w; const float* f0_an_bias; const float* f0_an_logs; /* f.0: (hid,C/2,3,3),(hid),(hid) */
    const float* f2_w; const float* f2_an_bias; const float* f2_an_logs; /* f.2: (hid,hid,1,1),(hid),(hid) */
    const float* f4_w; const float* f4_bias;    const float* f4_logs;    /* f.4 / Split2d.conv2d_zeros:
                                                                            (Cout,Cin,3,3),(Cout),(Cout) */
} glowhip_layer_desc;

typedef struct glowhip_plan glowhip_plan;

/* Build a plan (HOST bookkeeping only; descs are copied).  Returns NULL on error. */
glowhip_plan* glowhip_plan_create(const glowhip_layer_desc* layers, int n_layers);
void glowhip_plan_destroy(glowhip_plan* plan);

/* Bytes of the persistent packed-parameter buffer / of the per-call workspace for batch N. */
size_t glowhip_plan_packed_bytes(const glowhip_plan* plan);
size_t glowhip_plan_workspace_bytes(const glowhip_plan* plan, int N);

/* Re-derive everything that depends only on the parameters (exp(3 logs), K-major re-layout of the
 * convolution weights for the MFMA kernels, in-kernel LU -> log|det W| and W^-1) into `packed`.
 * Call after the parameters changed (optimizer step, load_state_dict, ActNorm init). */
int glowhip_plan_pack(glowhip_plan* plan, void* packed, size_t packed_bytes, glowhip_stream_t stream);
/* The same, restricted to the weight images a caller is going to use: GLOWHIP_PACK_INFERENCE = what encode / decode /
 * glow_forward read (the split-half images where those kernels apply), GLOWHIP_PACK_TRAINING = what glow_forward_train /
 * glow_backward read (exact-fp32 images + the flipped/transposed input-gradient images).  glowhip_plan_pack = both.
 * Scale tables and log|det W| are always refreshed, W^-1 with GLOWHIP_PACK_INVERSE or GLOWHIP_PACK_TRAINING. */
#define GLOWHIP_PACK_INFERENCE 1
#define GLOWHIP_PACK_TRAINING 2
#define GLOWHIP_PACK_INVERSE 4   /* W^-1 of the invertible 1x1 convolutions (decode; implied by TRAINING): without it
                                    log|det W| comes from an LU of the matrix alone (C^3/3 instead of 2 C^3 updates) and W^-1 is stale */
int glowhip_plan_pack_for(glowhip_plan* plan, void* packed, size_t packed_bytes, int use, glowhip_stream_t stream);
/* Part of a pack -- the LU factorisations (their log|det W| enters only a forward's final sum) and the weight images only the
 * deep levels read -- runs on a side stream the plan owns, forked from `stream` and joined by event inside whichever plan call
 * consumes the results (on that call's stream): a forward's first kernels do not wait for it.  Nothing for the caller to do,
 * except before capturing plan calls WITHOUT the pack into a hipGraph: glowhip_plan_pack_sync (HOST, blocks) waits for the side
 * part of the last pack so that the captured calls have nothing outside the graph to join. */
int glowhip_plan_pack_sync(glowhip_plan* plan);
/* The pack keeps its (plan-constant) job tables inside `packed` and sends them only once per buffer ADDRESS.  A caller that frees
 * `packed` and passes a new allocation -- which may come back at the same address -- says so with this call (HOST bookkeeping);
 * the next pack uploads the tables again. */
int glowhip_plan_forget_packed(glowhip_plan* plan);

/* FlowModel.encode: x (N, C0,H0,W0 of layer 0) -> z (output shape of the last layer),
 * logdet_out[n] = (logdet_in ? logdet_in[n] : 0) + sum of all layers' log-determinant terms.
 * noise (shape of x, may be NULL) is added to x first (dequantisation, network/model.py:421). */
int glowhip_plan_encode(glowhip_plan* plan, const void* packed, const float* x, const float* noise,
                        const float* logdet_in, float* z, float* logdet_out, int N,
                        void* workspace, size_t workspace_bytes, glowhip_stream_t stream);

/* FlowModel.decode: z -> x, layers run in reverse.  eps[k] (device pointer, shape of the k-th Split2d's
 * z2 in DECODE order, already multiplied by eps_std) is the injected N(0,1) draw of
 * GaussianDiag.sample (network/module.py:470-483).  logdet_out may be NULL. */
int glowhip_plan_decode(glowhip_plan* plan, const void* packed, const float* z, const float* const* eps,
                        int n_eps, const float* logdet_in, float* x, float* logdet_out, int N,
                        void* workspace, size_t workspace_bytes, glowhip_stream_t stream);

/* Glow.normal_flow (network/model.py:409-452) in one call: z = x + noise; objective = -ln(2^n_bits)*CHW
 * + encode logdet + logp(z | prior_mean, prior_logs);  nll = -objective / (ln2*CHW).
 * prior_mean/prior_logs: (N,Cz,Hz,Wz) with batch stride prior_stride, NULL = zeros.
 * objective_out may be NULL. */
int glowhip_glow_forward(glowhip_plan* plan, const void* packed, const float* x, const float* noise,
                         const float* prior_mean, const float* prior_logs, long prior_stride, int n_bits,
                         float* z, float* nll_out, float* objective_out, int N,
                         void* workspace, size_t workspace_bytes, glowhip_stream_t stream);

/* The same from 8-bit pixels (N,C,H,W) as a data loader holds them: x = x_u8 / divisor (255 for torchvision's ToTensor,
 * dataset/celeba.py:74-86) + noise, converted inside the plan's leading Squeeze2d -- no fp32 copy of the batch is ever
 * made.  The plan must start with a Squeeze2d layer. */
int glowhip_glow_forward_u8(glowhip_plan* plan, const void* packed, const uint8_t* x_u8, float divisor, const float* noise,
                            const float* prior_mean, const float* prior_logs, long prior_stride, int n_bits, float* z,
                            float* nll_out, float* objective_out, int N, void* workspace, size_t workspace_bytes,
                            glowhip_stream_t stream);

/* Kernel family of a plan's coupling networks (network/module.py:300-319) -- a property of the PLAN, not of the process:
 *   GLOWHIP_FAMILY_AUTO        the split-half f16 matrix-pipe kernels wherever the shape allows (default; hidden activations
 *                              |v| < 4094, beyond that the result is non-finite and flagged, never a finite wrong value);
 *   GLOWHIP_FAMILY_EXACT_FP32  v_mfma_f32_32x32x2_f32 kernels only: the reference's full fp32 range at 1/3 of the speed.
 * Takes effect at the next glowhip_plan_pack_for (the family's weight images are packed on demand) + encode / decode /
 * glow_forward.  HOST bookkeeping; the caller serialises it with the plan's own calls as for any other plan call. */
#define GLOWHIP_FAMILY_AUTO 0
#define GLOWHIP_FAMILY_EXACT_FP32 1
int glowhip_plan_set_family(glowhip_plan* plan, int family);
int glowhip_plan_get_family(const glowhip_plan* plan);

/* Range / non-finite status of the encode / decode / glow_forward call that last ran with `workspace` for batch N (enqueue it
 * on the same stream right after that call): status_out[n] (DEVICE, N int32) = bit 0: a NaN term, bit 1: a +inf term, bit 2: a
 * -inf term entered sample n's log-det sum (the sticky flags the nll reports as NaN / inf) | bit 3: an element of
 * result[n*elems_per_sample ..) (the call's output tensor: x of a decode, z of an encode; may be NULL) is not finite.
 * Non-zero means: out of the product kernels' range (or genuinely diverged) -- re-run on GLOWHIP_FAMILY_EXACT_FP32 to get
 * the reference's answer (Glow.reverse_flow network/model.py:454-471 has no nll that would show it).  No host sync. */
int glowhip_plan_status(const glowhip_plan* plan, const void* workspace, size_t workspace_bytes, int N, const float* result,
                        long elems_per_sample, int32_t* status_out, glowhip_stream_t stream);

/* Dequantisation noise drawn INSIDE the leading squeeze (network/model.py:421): with enable != 0, every later
 * glowhip_glow_forward / _u8 call whose `noise` is NULL adds U(0, 2^-n_bits) from the counter-based generator
 * Philox4x32-10(key = seed; counter = element index, call number) -- no noise tensor, no RNG launch.  The call number starts at
 * 0 when the seed is (re)set and advances by one per such call; *next_call (HOST, may be NULL) receives its next value
 * (enable < 0: query only).  glowhip_dequant_noise writes the draw of a given call as a tensor (n elements in the order of x):
 * a forward with that tensor as `noise` is bitwise equal to the in-kernel draw. */
int glowhip_plan_set_dequant_rng(glowhip_plan* plan, unsigned long long seed, int enable, unsigned long long* next_call);
int glowhip_dequant_noise(float* out, long n, unsigned long long seed, unsigned long long call, int n_bits, glowhip_stream_t stream);
/* The same switch with the position given by the caller: the next glow_forward without a noise tensor draws with (seed, call),
 * the one after with call + 1 ...  Lets ONE stream serve every plan of a process (each batch shape has its own plan) and lets
 * the ranks of a data-parallel job key it differently (pytorch-glow_amd/network/model.py: torch's seed, the rank folded in). */
int glowhip_plan_set_dequant_stream(glowhip_plan* plan, unsigned long long seed, unsigned long long call);

/* Data-dependent ActNorm initialisation pass over a whole plan (first training-mode forward,
 * network/trainer.py:112-115 + network/module.py:45-46,66-67): runs encode on x and writes every
 * ActNorm's bias/logs THROUGH the parameter pointers of the layer descs (which must be writable).  Ends with
 * glowhip_plan_pack_for(GLOWHIP_PACK_INFERENCE): pack _TRAINING / _INVERSE data before training / decoding. */
int glowhip_plan_actnorm_init(glowhip_plan* plan, void* packed, size_t packed_bytes, const float* x,
                              const float* noise, float actnorm_scale, int N, void* workspace,
                              size_t workspace_bytes, glowhip_stream_t stream);

/* Output shape of the plan for a given direction (HOST). out[3] = {C,H,W}. */
int glowhip_plan_output_shape(const glowhip_plan* plan, int reverse, int32_t out[3]);

/* ------------------------------------------------------------------------------------------------
 * Training step (reference network/trainer.py:123-140: forward, loss = mean(nll), backward)
 * ---------------------------------------------------------------------------------------------- */
/* Writable fp32 gradient tensors of one layer, same shapes as the parameters named in glowhip_layer_desc.
 * Entries of parameters a layer does not have (and any gradient the caller does not want) are NULL. */
typedef struct glowhip_layer_grads {
    float* an_bias; float* an_logs; float* invconv_w;
    float* f0_w; float* f0_an_bias; float* f0_an_logs;
    float* f2_w; float* f2_an_bias; float* f2_an_logs;
    float* f4_w; float* f4_bias; float* f4_logs;
} glowhip_layer_grads;

/* Bytes of the activation tape / of the training workspace for batch N. */
size_t glowhip_plan_tape_bytes(const glowhip_plan* plan, int N);
size_t glowhip_plan_train_workspace_bytes(const glowhip_plan* plan, int N);

/* glowhip_glow_forward that also records the tape (every layer output + the coupling networks' hidden
 * activations) needed by glowhip_glow_backward.  Same results as glowhip_glow_forward. */
int glowhip_glow_forward_train(glowhip_plan* plan, const void* packed, const float* x, const float* noise,
                               const float* prior_mean, const float* prior_logs, long prior_stride, int n_bits,
                               float* z, float* nll_out, float* objective_out, int N, void* tape, size_t tape_bytes,
                               void* workspace, size_t workspace_bytes, glowhip_stream_t stream);

/* Gradients of a scalar loss L given nll_grad[n] = dL/dnll_n (1/B for Glow.generative_loss = mean(nll),
 * network/model.py:496-506) and optionally z_grad = dL/dz (NULL = 0): every non-NULL entry of grads[layer] is
 * OVERWRITTEN with dL/dparameter; grad_x (shape of x, may be NULL) receives dL/dx.  `x`, `tape`, `packed` and the
 * prior arguments must be the ones of the matching glowhip_glow_forward_train call. */
int glowhip_glow_backward(glowhip_plan* plan, const void* packed, const float* x, const void* tape, size_t tape_bytes,
                          const float* nll_grad, const float* z_grad, const float* prior_mean,
                          const float* prior_logs, long prior_stride, const glowhip_layer_grads* grads,
                          float* grad_x, int N, void* workspace, size_t workspace_bytes, glowhip_stream_t stream);

/* Gradient-ready marks: lets the caller overlap the gradient all-reduce of one process per GPU (the replacement of
 * nn.DataParallel's reduce-to-GPU-0, network/trainer.py:117-123) with the rest of the backward sweep.  The sweep runs from the
 * last layer to the first; glowhip_glow_backward records events[i] (hipEvent_t handles owned by the caller) on its stream as
 * soon as every kernel writing the convolution WEIGHT gradients (f0_w, f2_w, f4_w -- 99.8 % of the gradient bytes) of all layers
 * with index >= after_layer[i] has been enqueued.  after_layer must decrease.  The reduction-type gradients (biases, logs,
 * invconv matrices of ALL layers) are final only after the call's last kernel.  n = 0 clears the marks.  HOST bookkeeping. */
int glowhip_plan_backward_marks(glowhip_plan* plan, const int32_t* after_layer, void* const* events, int n);

/* ------------------------------------------------------------------------------------------------
 * Optimiser step (reference network/trainer.py:142-150: clip_grad_value_, clip_grad_norm_, optimizer.step() with the
 * torch.optim.Adam / Adamax of network/builder.py:10-13,108-113) over ALL parameters as two launches.
 * A chunk is at most 2^16 consecutive elements of one parameter with its gradient and optimiser state (m = exp_avg,
 * v = exp_avg_sq for Adam / exp_inf for Adamax).  partial_dev: n_chunks doubles of scratch.  kind: 0 = Adam (amsgrad off),
 * 1 = Adamax; step: 1-based count of this update (bias corrections); clip_value / max_norm <= 0: that clipping off.
 * grad_norm_out (1 float, may be NULL) <- total gradient norm after the value clipping, before the norm clipping, as
 * clip_grad_norm_ returns it.  Gradients are clipped IN PLACE as the reference's utilities do.
 * skip_if_nonfinite != 0: when that norm is NaN / inf the update launch leaves parameters, state and gradients untouched (the
 * device-side "found inf" of a loss scaler: no host sync) -- the caller reads grad_norm_out at its next natural sync and re-runs
 * the batch on GLOWHIP_FAMILY_EXACT_FP32 (training.TrainLoop).  0 = torch's semantics (NaN gradients give NaN parameters).
 * ---------------------------------------------------------------------------------------------- */
typedef struct glowhip_optim_chunk { float* param; float* grad; float* m; float* v; int32_t n; int32_t pad; } glowhip_optim_chunk;
int glowhip_optim_step(const glowhip_optim_chunk* chunks_dev, int n_chunks, int kind, float lr, double beta1, double beta2,
                       float eps, float weight_decay, int step, float clip_value, float max_norm, double* partial_dev,
                       float* grad_norm_out, int skip_if_nonfinite, glowhip_stream_t stream);
/* The same step with the values that change from step to step read from DEVICE memory -- hyper_dev[3] = {lr (a float's value),
 * 1 - beta1^step, 1 - beta2^step} as doubles, computed by the caller the way glowhip_optim_step does (python-double arithmetic, as
 * torch) -- so that the two launches can sit in a captured hipGraph whose kernel arguments are frozen (training.GraphedTrainStep:
 * the caller uploads the three values before each replay).  Same arithmetic, same bits as glowhip_optim_step. */
int glowhip_optim_step_dev(const glowhip_optim_chunk* chunks_dev, int n_chunks, int kind, const double* hyper_dev, double beta1,
                           double beta2, float eps, float weight_decay, float clip_value, float max_norm, double* partial_dev,
                           float* grad_norm_out, int skip_if_nonfinite, glowhip_stream_t stream);

/* Per-launch timing for benchmarks (HIP events recorded on the execution stream around every kernel of
 * the coupling path).  enable=1 creates an event pool (host resource), enable=0 destroys it; while enabled
 * every encode/decode appends records.  glowhip_plan_timing_read synchronises with the recorded events,
 * copies up to `max` records (launch order) and clears the list. */
enum { GLOWHIP_K_CHANMIX = 0, GLOWHIP_K_CONV_F0 = 1, GLOWHIP_K_CONV_F2 = 2, GLOWHIP_K_CONV_F4 = 3, GLOWHIP_K_OTHER = 4,
       GLOWHIP_K_CNET = 5 /* k_cnet: f.0 + f.2 + f.4 of a FlowStep */, GLOWHIP_K_CFINISH = 6 /* its finishing kernel */,
       GLOWHIP_K_CNET_TAPE = 7 /* training forward: k_cnet storing h1 / h2 */, GLOWHIP_K_CNET_BWD = 8 /* input-gradient chain on k_cnet */,
       GLOWHIP_K_WGRAD = 9 /* a FlowStep's three weight-gradient GEMMs + their split-K reduction */ };
typedef struct glowhip_timing_record {
    int32_t kind;   /* GLOWHIP_K_* */
    int32_t layer;  /* index into the plan's layer list */
    int32_t mfma;   /* 1 = MFMA kernel, 0 = direct kernel */
    float ms;       /* elapsed between the two events */
} glowhip_timing_record;
int glowhip_plan_timing_enable(glowhip_plan* plan, int enable);
int glowhip_plan_timing_read(glowhip_plan* plan, glowhip_timing_record* out, int max, int* n_out);

/* Testing hook for the fused tail convolution, so every variant can be exercised at any batch size: low byte =
 * pixels per workgroup (16/32/64/128; 0 = automatic, chosen by a cost model), | 0x100 = always split the
 * out-channel tiles over blockIdx.y, | 0x200 = never split, | 0x400 = no LDS-DMA tail kernels.
 * Kernel-family switches (the parity tests of the exact-fp32 kernels, A/B runs): | 0x800 = exact-fp32 MFMA kernels only
 * (split-half f16 path off, training included), | 0x8000 = no fusion around the channel mixer: k_squeeze + k_chanmix + a plain
 * finishing kernel as separate launches (bitwise equal to the fused forms); bits 22..24 = 1, 2 or 4: that many row splits of
 * k_cnet;
 * | 0x2000000 = cnet with 128-pixel tiles only, | 0x4000000 = 64-pixel tiles wherever supported,
 * | 0x10000 = no k_cnet1w (the one-wave-per-SIMD coupling-network kernel of csrc/cnet1w_sh.hip; k_cnet takes its launches: A/B),
 * | 0x80000 = f.2's weight-gradient GEMM on 128-column tiles at every level (no k_wgrad_gemm_ps512: A/B),
 * | 0x40000 = no backward instance of k_cnet1w (the level-1 input-gradient launch back on k_cnet's 64-pixel tiles: A/B),
 * | 0x20000 = the row-split instance of k_cnet1w where it applies (C = 24 levels with 112 .. tiles of 128 pixels; off by default:
 *   measured slower than k_cnet's 64-pixel tiles there; parity tests and A/B),
 * | 0x100000 = FUSED FINISHING on: a k_cnet1w launch finishes its FlowStep itself (arrival counters per tile, the last workgroup
 *   to arrive runs the finishing kernel's code; bit-identical, off by default: measured slower),
 * | 0x200000 = log|det W| of the 12 / 24 / 48-wide invconv matrices on the workgroup-wide LU instead of one wave per matrix (same bits: A/B),
 * | 0x8000000 = the finishing step of a FlowStep runs inside the next FlowStep's k_cnet (off by default: measured slower),
 * | 0x10000000 = the finishing kernel takes its pixel chunks in block order instead of the XCD-affine order (A/B),
 * | 0x20000000 = glowhip_plan_pack entirely on the caller's stream, no side-stream fork (A/B),
 * | 0x40000000 = glowhip_glow_forward_train on the per-layer kernels instead of the taping k_cnet (A/B, parity tests),
 * | 0x80000000 = glowhip_glow_backward's input-gradient chain on the per-layer kernels instead of the backward k_cnet.
 * 0 restores automatic selection.
 * Process-wide, not thread safe: a testing hook, not part of the operator surface. */
void glowhip_debug_force_tail_tile(int pixels_and_flags);

/* Introspection for tests / benchmarks: which kernels a plan will launch ("mfma" or "direct" per
 * convolution).  Writes a NUL-terminated description into buf. */
int glowhip_plan_describe(const glowhip_plan* plan, char* buf, size_t buf_bytes);
/* The same for a given batch size N: kernel choices that depend on how many workgroups a launch would have (the fused
 * f.0 + f.2 kernel runs only when N*H*W/64 workgroups cover the chip) are resolved as encode/decode would resolve them. */
int glowhip_plan_describe_for(const glowhip_plan* plan, int N, char* buf, size_t buf_bytes);
/* Run-time evidence of kernel selection: "kernel_family=launches\n" lines for every kernel family this plan has launched
 * (coupling path of encode / decode / glow_forward) since creation or the last reset.  HOST bookkeeping, counted at launch. */
int glowhip_plan_launch_counts(glowhip_plan* plan, char* buf, size_t buf_bytes, int reset);

#ifdef __cplusplus
}
#endif
#endif /* GLOWHIP_H */
