#!/usr/bin/env python3
"""bench.py -- images/sec of a full Glow forward + log-determinant (Glow.normal_flow) on MI355X.

Workload (default, BASELINE.json configs[1], per GPU): CelebA-shaped 64x64x3 batch of 64, L=3, K=32, hidden 512,
affine coupling, invertible 1x1 conv; fp32; synthetic uniform [0,1) images resident in HBM; random-init weights
of the reference's architecture + data-dependent ActNorm init on the first batch.  `--config D` / `--config E` run
BASELINE configs[3] / [4] (128x128 L=4 K=48, 32 images per GPU; 256x256 L=6 K=32, 16 images per GPU) as secondary lines.

One step = dequantisation noise (on-device RNG) -> squeeze/FlowStep/Split2d stack -> top prior -> nll (N,) ->
sum(nll) [-> RCCL all-reduce of the scalar when N > 1].  Every step also re-derives the parameter-dependent
data (glowhip_plan_pack: LU of the invconv weights -> log|det W|, exp(3 logs), MFMA weight images), i.e. the
weights are treated as freshly updated each step exactly as inside a training loop -- nothing is cached across
steps.  Multi-GPU: one process per GPU, batch sharded (weak scaling, fixed images per GPU), no data-path collective
other than the scalar loss all-reduce.

`python bench.py --gpus N` launches its N ranks ITSELF (one child process per GPU, spawned before anything touches a
GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the children's environment, RCCL rendezvous on
127.0.0.1).  Started under `python -m torch.distributed.run` (RANK and WORLD_SIZE already set) it is one rank of that job.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel timed live with HIP events
around each of its launches in an instrumented pass of the same step) and `cpu_baseline` (the CPU oracle timed on this
box's host cores on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, f32 in / f32 accumulate
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16/f16 dense MFMA (v_mfma_f32_32x32x16_f16)
# split-half path (csrc/sh.h): one fp32-accurate product = 3 f16 MFMA products => algorithmic-flop ceiling = 2500 / 3
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0

# BASELINE.json configs (per-GPU batches: SURVEY.md section 8 "B B=64; C 64/GPU; D 32/GPU; E 16/GPU")
CONFIGS = {
    "B": dict(image=64, L=3, K=32, hidden=512, batch=64, cpu_sample=64,
              label="CelebA 64x64x3 Glow L=3 K=32 hidden=512 affine+invconv", ref="BASELINE configs[1]"),
    "D": dict(image=128, L=4, K=48, hidden=512, batch=32, cpu_sample=4,
              label="CelebA 128x128x3 Glow L=4 K=48 hidden=512 affine+invconv", ref="BASELINE configs[3]"),
    "E": dict(image=256, L=6, K=32, hidden=512, batch=16, cpu_sample=1,
              label="256x256x3 Glow L=6 K=32 hidden=512 affine+invconv (openai/glow full config)", ref="BASELINE configs[4]"),
}
HEADLINE_METRIC = "images/sec full Glow fwd+logdet, 64x64x3 L=3 K=32, 1/2/4/8 GPU"


def build_model(G, util, device, cfg, batch, seed=2384):
    import numpy as np
    import torch
    hps = util.load_profile("celeba")  # built-in profile, reference schema
    hps.model.image_shape = [cfg["image"], cfg["image"], 3]
    hps.model.L, hps.model.K, hps.model.hidden_channels = cfg["L"], cfg["K"], cfg["hidden"]
    hps.optim.num_batch_train = batch
    hps.device.graph = ["cuda:0"]  # one process per GPU: one replica per process
    torch.manual_seed(seed)
    np.random.seed(seed)
    glow = G.Glow(hps)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():  # zero-init tails would make the coupling trivial: give them the survey's N(0, 0.002)
        for name, p in glow.named_parameters():
            if ".f.4." in name or "conv2d_zeros" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.002)
    return glow.to(device), hps


def flop_per_image(glow):
    """Contraction FLOPs of one forward (SURVEY.md section 8d formula: 2 FLOP per MAC, convolutions + invconv)."""
    hid = glow.hps.model.hidden_channels
    total = 0.0
    c, h, w = 3, glow.hps.model.image_shape[0], glow.hps.model.image_shape[1]
    for i in range(glow.flow.L):
        c, h, w = c * 4, h // 2, w // 2
        cout = c if glow.hps.ablation.flow_coupling == "affine" else c // 2
        total += glow.flow.K * 2.0 * h * w * (9 * (c // 2) * hid + hid * hid + 9 * hid * cout + c * c)
        if i < glow.flow.L - 1:
            total += 2.0 * h * w * 9 * (c // 2) * c
            c //= 2
    return total


def usable_cores(cap=32):
    """Host cores this process may really use: affinity mask AND cgroup CPU quota (os.cpu_count() reports the
    machine's 256 hardware threads even inside a small cgroup; oversubscribing OpenMP by 30x is pathological)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(glow, x_cpu, cfg, budget_s=90.0, fallback_batch=None):
    """Time the CPU oracle (a port of the reference's eager op sequence) on this box's host cores, as SURVEY.md section 8d asks: the
    metric's own batch (B = 64 at config B), one warm-up forward, then the MEDIAN OF 3 timed forwards (~12 s each on the box's 16
    usable cores: ~50 s in all, of a driver run that has half an hour).  `budget_s` only guards a much slower host: if the
    warm-up alone says three more forwards will not fit, the sample falls back to `fallback_batch` images (images are independent
    units, the rate per image is the same) -- and the line says which path was taken (`path`)."""
    import torch
    from oracle import glow_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in glow.state_dict().items()}

    def forward_time(xb):
        ocfg = O.default_cfg(image_shape=(cfg["image"], cfg["image"], 3), hidden_channels=cfg["hidden"], K=cfg["K"], L=cfg["L"],
                             batch=xb.shape[0])
        noise = torch.rand_like(xb) / 256
        t0 = time.perf_counter()
        O.glow_forward(xb, noise, sd, ocfg)
        return time.perf_counter() - t0

    path = "full batch"
    with torch.no_grad():
        warm = forward_time(x_cpu)            # warm-up (oneDNN primitive creation), excluded
        if fallback_batch and fallback_batch < x_cpu.shape[0] and 4 * warm > budget_s:
            x_cpu = x_cpu[:fallback_batch]
            path = f"fall-back to {fallback_batch} images (the warm-up forward of the full batch took {warm:.1f}s: 3 more do not fit {budget_s:.0f}s)"
            warm = forward_time(x_cpu)
        times = [forward_time(x_cpu) for _ in range(3)]
    med = sorted(times)[1]
    return {"value": round(x_cpu.shape[0] / med, 3), "unit": "images/sec", "cores": cores, "kind": "port", "path": path,
            "sample": f"median of 3 forwards of batch {x_cpu.shape[0]} ({', '.join(f'{t:.2f}' for t in times)} s) after one warm-up "
                      f"({warm:.1f}s, excluded); same model/weights as the GPU run, fp32, torch CPU threads={cores}; oracle/glow_oracle.py"}


KERNEL_KINDS = {0: "chanmix", 1: "conv_f0_3x3", 2: "conv_f2_1x1", 3: "conv_f4_3x3_tail", 5: "cnet_f0+f2+f4", 6: "cnet_finish",
                7: "cnet_tape_f0+f2+f4", 8: "cnet_bwd_dgrad_chain", 9: "wgrad_gemms"}
CNET_DESC = ("k_cnet1w / k_cnet: a FlowStep's coupling net f.0+f.2+f.4 in ONE launch, all levels (k_cnet1w, one wave per SIMD with h1 / h2 chained through "
             "the register file, where a level gives >= 224 128-pixel tiles; k_cnet, two waves per SIMD, elsewhere: f.0 3x3 conv C/2->512 + ActNorm + ReLU, f.2 1x1 conv 512->512 + "
             "ActNorm + ReLU, f.4 3x3 conv 512->C as taps-as-rows GEMM + tap sums; h1, h2 stay in LDS / registers; fp32-accurate products "
             "as 3 f16 MFMAs, peak = 2500/3 TFLOP/s algorithmic; all levels' launches)")


def traffic_record(cnet_path, sh_path):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (profiles/pmc_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this command, scripts/prof_pmc.sh) -- not re-measured by this run, so the record says which
    commit it was collected at and whether the kernel's source has changed since (sha256 of csrc/cnet_sh.hip + cnet1w_sh.hip at collection time)."""
    import hashlib
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None, None
    tj = json.load(open(tpath))
    key = "k_cnet_hbm_bytes_per_launch" if cnet_path else "k_gemm_glds_hbm_bytes_per_launch"
    src = {"file": "profiles/pmc_traffic.json", "collected_at_commit": tj.get("commit"),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (scripts/prof_pmc.sh), not re-measured by this run"}
    if cnet_path:
        try:
            now = hashlib.sha256(b"".join(open(os.path.join(ROOT, "pytorch-glow_amd", "csrc", f), "rb").read()
                                          for f in ("cnet_sh.hip", "cnet1w_sh.hip"))).hexdigest()[:16]      # k_cnet + k_cnet1w
        except OSError:
            now = None
        src["kernel_source_sha16_at_collection"], src["kernel_source_sha16_now"] = tj.get("cnet_sh_sha16"), now
        src["stale"] = tj.get("cnet_sh_sha16") != now
        if src["stale"]:
            print("bench.py: WARNING profiles/pmc_traffic.json was collected before the last change to csrc/cnet_sh.hip / cnet1w_sh.hip "
                  "(re-run scripts/prof_pmc.sh + scripts/pmc_summary.py)", file=sys.stderr, flush=True)
    return tj.get(key), src


def instrumented_pass(plan, hps, B, run_once, passes=3, with_traffic=False, train=False):
    """`passes` instrumented runs of a step (HIP events recorded by the C executor on the execution stream around every
    coupling-path launch, glowhip_plan_timing_*), OUTSIDE any timed loop: per-kernel-family time per step, launch counters, and
    the roofline of the dominant kernel = k_cnet over all its launches: achieved = sum of algorithmic FLOPs / sum of the launches'
    event durations.  Algorithmic FLOPs of a launch: 2 (9 (C/2) hid + hid^2 + 9 hid Cout) N HW; the
    input-gradient chain of the training step has the same shape with the 3x3 layers swapped, i.e. the same count."""
    plan.launch_counts(reset=True)
    plan.timing(True)
    for _ in range(passes):
        run_once()
    recs = plan.timing_read()
    plan.timing(False)
    counts = plan.launch_counts(reset=True)
    launches = {k: v // passes for k, v in counts.items() if not k.startswith("variant:")}
    variants = {k[len("variant:"):]: v // passes for k, v in counts.items() if k.startswith("variant:")}
    hid = hps.model.hidden_channels
    desc = plan.describe(B)
    cnet_path = "cnet-sh2" in desc
    sh_path = cnet_path
    bd = {}
    dom_ms, dom_flop, dom_bytes, dom_n = 0.0, 0.0, 0.0, 0
    per_level = {}
    for kind, layer, mfma, ms in recs:
        d = plan._descs[layer]
        key = f"{KERNEL_KINDS.get(kind, 'other')}_C{d.C}_{d.H}x{d.W}"
        bd.setdefault(key, [0.0, 0])
        bd[key][0] += ms
        bd[key][1] += 1
        total_px = B * d.H * d.W
        uses_128 = (hid // 128) * ((total_px + 127) // 128) >= 512
        cout = d.C if hps.ablation.flow_coupling == "affine" else d.C // 2
        dom = False
        if cnet_path:
            # dominant kernel = k_cnet: the whole coupling network of a FlowStep (f.0 3x3, f.2 1x1, f.4 3x3) in one launch;
            # h1 and h2 never reach HBM: algorithmic bytes = read z1 + write the f.4 partial sums (+ halo rows, ignored)
            if kind in (5, 7, 8):
                dom, fl = True, 2.0 * (9 * (d.C // 2) * hid + hid * hid + 9 * hid * cout) * total_px
                by = 4.0 * (d.C // 2) * total_px + 4.0 * cout * total_px
                if kind != 5:        # taping / backward launches also store the two hidden tensors (fp32) and read / write sign words
                    by += 2 * 4.0 * hid * total_px + 2 * hid * total_px / 8
        elif kind == 2 and mfma and uses_128:   # exact-fp32 path: k_gemm_glds (128x128 tiles)
            dom, fl, by = True, 2.0 * hid * hid * total_px, 2.0 * 4.0 * hid * total_px
        if dom:
            dom_ms += ms; dom_flop += fl; dom_bytes += by; dom_n += 1
            lv = per_level.setdefault(f"C{d.C}_{d.H}x{d.W}", [0.0, 0.0, 0])
            lv[0] += ms; lv[1] += fl; lv[2] += 1
    achieved = dom_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic, traffic_src = traffic_record(cnet_path, sh_path) if with_traffic else (None, None)
    peak = PEAK_SPLIT_TFLOPS if sh_path else PEAK_FP32_MFMA_TFLOPS
    name = (CNET_DESC + ("; here the taping forward (MODE 1: also stores h1 / h2 as fp16, pixel-tile-major, + sign words) and the input-gradient chain "
                         "(MODE 2) launches of the training step -- level 1 on the taping / backward instances of k_cnet1w, levels 2 / 3 on k_cnet" if train else "")) if cnet_path else \
           "k_gemm_glds (f.2: 1x1 conv 512->512 + ActNorm + ReLU, fp32-input MFMA, 128x128 tiles)"
    # (flat, short values first: the driver's record keeps scalars and the head of strings; the full records follow)
    tnote = None
    if traffic_src is not None:
        tnote = (f"{traffic_src['file']} @ {str(traffic_src.get('collected_at_commit'))[:10]}: rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, "
                 f"NOT re-measured in this run{'; STALE (kernel source changed since)' if traffic_src.get('stale') else ''}")
    roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": tnote, "traffic_source_detail": traffic_src,
            **{f"frac_{k}": round(v[1] / (v[0] * 1e-3) / 1e12 / peak, 4) for k, v in sorted(per_level.items()) if v[0] > 0},
            "launches": dom_n // passes, "avg_launch_us": round(1e3 * dom_ms / max(dom_n, 1), 2),
            "flop_per_launch_avg": dom_flop / max(dom_n, 1),
            "issued_mfma_tflops": round(achieved * (3 if sh_path else 1), 1),
            "algorithmic_hbm_bytes_per_launch_avg": dom_bytes / max(dom_n, 1),
            "hbm_GBps": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 1) if dom_ms > 0 else 0.0,
            "hbm_frac_of_8TBps": round(dom_bytes / (dom_ms * 1e-3) / 8e12, 4) if dom_ms > 0 else 0.0,
            "per_level": {k: {"avg_launch_us": round(1e3 * v[0] / v[2], 2), "frac": round(v[1] / (v[0] * 1e-3) / 1e12 / peak, 4)}
                          for k, v in sorted(per_level.items()) if v[0] > 0},
            "dominant_kernel_ms_per_step": round(dom_ms / passes, 3)}
    return {"roofline": roof, "breakdown": {k: round(v[0] / passes, 4) for k, v in sorted(bd.items())}, "launches": launches,
            "variants": variants, "sh_path": sh_path}



# ------------------------------------------------------------------------------------------------ workloads
def setup_workload(G, util, parallel, device, cfg_name, mode, B, rank, world, repack, graph=False):
    """Model + synthetic batch + one-step closure of a (config, mode) workload; sets torch's grad mode for it."""
    import torch
    cfg = CONFIGS[cfg_name]
    glow, hps = build_model(G, util, device, cfg, B)
    x = torch.rand(B, 3, cfg["image"], cfg["image"], generator=torch.Generator().manual_seed(2384 + rank)).to(device)
    # data-dependent ActNorm init on rank 0's first batch, then broadcast (reference trainer.py:112-115)
    glow.train()
    glow.flow.plan_for(x)            # (plan construction is host work: not part of the init pass's time)
    torch.cuda.synchronize()
    t_init = time.perf_counter()
    with torch.no_grad():       # (the init pass needs no activation tape)
        parallel.data_dependent_init(glow, x, rank=rank, world=world)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t_init
    glow.eval()
    plan = glow.flow.plan_for(x)
    wl = dict(glow=glow, hps=hps, plan=plan, x=x, cfg=cfg, init_ms=round(1e3 * t_init, 1), step0=dict(parallel.STEP0))
    if mode == "inverse":
        wl["z_top"] = z_top = torch.randn((B,) + tuple(plan.out_chw), device=device) * 0.7
    if mode == "train":
        from pytorch_glow_amd import training
        # Adam + noam warm-up + clip 5/100 (celeba profile); single rank: from its fourth step on the step is ONE hipGraph launch
        loop = training.TrainLoop(glow, hps, rank=rank, world=world, graph=graph and world == 1)
        wl["loop"] = loop
        torch.set_grad_enabled(True)
    else:
        torch.set_grad_enabled(False)   # forward+logdet metric: inference path (no activation tape)

    def step():
        if mode == "checked":   # the DROP-IN call: glow(x) in eval under no_grad, range check on (what infer.py / Trainer validation run)
            z, nll, _ = glow(x)
            return nll.sum()
        if mode == "train":     # secondary metric: the reference's training step (trainer.py:123-150)
            loss, _ = loop.step(x)
            return loss * (world * B)
        if mode == "inverse":   # secondary metric: sampling (eps drawn on device, W^-1 from the in-kernel LU)
            plan.ensure_packed(repack, use=plan.PACK_INFERENCE | plan.PACK_INVERSE)
            xs = glow.reverse_flow(z_top, None, eps_std=0.7)
            return xs.sum()
        if wl["graph"] is not None:   # the whole forward (noise draw + pack + launch list) as ONE hipGraph launch
            z, nll = wl["graph"]()
        else:
            z, nll, _ = glow.normal_flow(x, None, repack=repack)
        return parallel.reduce_loss(nll, world)
    wl["step"] = step
    wl["graph"], wl["launch"] = None, "eager (one C call per forward issuing the kernel list)"
    if graph and mode == "forward":
        try:
            wl["graph"] = glow.capture_forward(x, repack=repack)
            wl["launch"] = "hipGraph (forward captured once, one graph launch per step)"
        except Exception as e:     # capture is an optimisation of the host side only: the eager launch list is the same work
            wl["launch"] += f"; hipGraph capture failed: {type(e).__name__}: {str(e)[:200]}"
    return wl


SECONDARY = [("B", ["checked"]), ("D", ["forward"]), ("E", ["forward", "inverse"]), ("B", ["train"])]


def secondary_workloads(G, util, parallel, device, steps=3, warmup=3):
    """The other BASELINE configurations under the SAME process and clock as the headline line, AFTER its timed region (nothing
    of this is in `value` / `ms_per_step`): a few timed steps each of configs[3] (D) and configs[4] (E) forward, E sampling
    (reverse_flow, derived data kept across steps as when sampling from a trained model) and the config-B training step.  One
    GPU, per-GPU batches of SURVEY.md section 8; rates are images/sec of this one GPU."""
    import gc
    import torch
    out = {}
    for cfg_name, modes in SECONDARY:
        cfg = CONFIGS[cfg_name]
        B = cfg["batch"]
        wl = None
        for mode in modes:
            t_wall = time.perf_counter()
            if wl is None or mode == "train":
                wl = setup_workload(G, util, parallel, device, cfg_name, mode, B, 0, 1, repack=(mode == "forward"), graph=(mode == "train"))
                step = wl["step"]
            else:       # the same model and batch in another mode (config E: forward, then sampling)
                plan, glow = wl["plan"], wl["glow"]
                z_top = torch.randn((B,) + tuple(plan.out_chw), device=device) * 0.7

                def step():
                    plan.ensure_packed(False, use=plan.PACK_INFERENCE | plan.PACK_INVERSE)
                    return glow.reverse_flow(z_top, None, eps_std=0.7).sum()
            n_timed = steps + (2 if mode == "train" else 0) + (17 if mode == "checked" else 0)
            for _ in range(warmup + (7 if mode == "train" else 0)):      # (the training step's first iterations still allocate:
                step()                                                   #  12 GB of tape, gradient buckets, optimiser state)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            c0, m0 = time.process_time(), time.thread_time()
            e0.record()
            for _ in range(n_timed):
                last = step()
            e1.record()
            dt_host, dt_cpu, dt_main = time.perf_counter() - t0, time.process_time() - c0, time.thread_time() - m0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if "loop" in wl:
                wl["loop"].flush()
            launch = None
            if "loop" in wl:      # how the timed steps were launched; the instrumented pass below needs the eager launch list (per-launch events)
                lp = wl["loop"]
                launch = ("hipGraph (training step captured after the eager warm-up steps, one graph launch per step)" if lp._graphed is not None
                          else "eager" + (f"; hipGraph capture failed: {lp.graph_error}" if lp.graph_error else ""))
                lp.graph = False
            # roofline of this workload's dominant kernel: the same live-event pass as the headline's, after (outside) its timed steps
            inst = instrumented_pass(wl["plan"], wl["hps"], B, step, passes=3, train=(mode == "train"))
            torch.cuda.synchronize()
            if "loop" in wl:
                wl["loop"].flush()
            name = "B_forward_checked" if mode == "checked" else f"{cfg_name}_{mode}"
            out[name] = {"value": round(B * n_timed / dt, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dt / n_timed, 3),
                         "ms_per_step_gpu_events": round(e0.elapsed_time(e1) / n_timed, 3), "steps": n_timed,
                         # host side of a step: wall time of the enqueue loop (includes any wait for the device: the training loop
                         # lets the host run at most two steps ahead) and the CPU time the process burnt in it (all threads)
                         "host_enqueue_ms_per_step": round(1e3 * dt_host / n_timed, 3), "host_cpu_ms_per_step": round(1e3 * dt_cpu / n_timed, 3),
                         "host_cpu_main_thread_ms_per_step": round(1e3 * dt_main / n_timed, 3),
                         "warmup": warmup + (7 if mode == "train" else 0), "batch": B,
                         "workload": f"{cfg['label']}, {mode}, batch {B} ({cfg['ref']})",
                         "finite": bool(torch.isfinite(last).all()), "data_dependent_init_ms": wl["init_ms"],
                         **({"launch": launch} if launch else {}),
                         **({"call": "glow(x) in eval mode under torch.no_grad(), Glow.range_check on (network/inferer.py:55,81, trainer.py:163): "
                                     "the weights are not re-derived per call (eval: they do not change)",
                             "range_fallbacks": type(wl["glow"])._RANGE_FALLBACKS} if mode == "checked" else {}),
                         "kernel_families": sorted(inst["launches"]),
                         "roofline": inst["roofline"], "breakdown_ms_per_step": inst["breakdown"],
                         "wall_s_incl_setup": round(time.perf_counter() - t_wall, 1)}
            del last
        del wl, step
        gc.collect()
        torch.cuda.empty_cache()
    torch.set_grad_enabled(False)
    return out


# ------------------------------------------------------------------------------------------------ rank launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Start one child per GPU (this process never touches a GPU), wait for all, return the worst exit code."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   GLOWHIP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # N ranks share this host's cores: each child gets its share for OpenMP / torch's intra-op pool instead of every child
        # sizing its pool to the whole machine (8 x 256 threads in a 16-core cgroup would turn the enqueue loop host-bound)
        env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores(cap=256) // n)))
        env.setdefault("MKL_NUM_THREADS", env["OMP_NUM_THREADS"])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0:
                rc = rc or code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending.values():   # exact PIDs of our own children
                    q.terminate()
        time.sleep(0.05)
    return rc


def dry_run_cpu(args, rank, world):
    """The N-rank protocol of the real run -- rendezvous, warm-up, barrier, K timed steps, barrier, MAX over ranks, one JSON line
    on rank 0 -- with empty steps on a gloo group.  Measures nothing; proves the launcher and the collective sequence."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus
    def step():
        s = torch.ones(()) * (rank + 1)
        if world > 1:
            dist.all_reduce(s)
        return s
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    multi = None
    if world > 1:      # the diagnostics the real N-rank line carries (same helpers, same collective order; tiny buffer, gloo)
        from pytorch_glow_amd import parallel
        per_rank = parallel.gather_scalars(1e3 * dt / max(args.steps, 1) + rank, "cpu")
        multi = {"world": dist.get_world_size(), "backend": dist.get_backend(), "per_rank_device_ms_per_step": per_rank,
                 "ms_per_step_rank_min": min(per_rank), "ms_per_step_rank_max": max(per_rank),
                 "gradient_allreduce_flat": parallel.timed_allreduce(1 << 16, "cpu", reps=2, warmup=1)}
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "dry_run": True, "value": 0.0, "unit": "images/sec", "n_gpus": world, "multi_rank": multi,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / max(args.steps, 1), 4),
                          "rccl_world_size": dist.get_world_size() if world > 1 else 1,
                          "allreduce_check": float(loss), "scaling": "weak"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def multi_rank_diagnostics(torch, dist, parallel, device, args, wl, marks, world, rank, forced):
    """What makes the first N-rank RCCL run diagnostic rather than merely green.  Runs on EVERY rank, after the
    timed region and its MAX reduction (nothing of this is in `value`), in the same order on all of them -- the collectives below
    pair up by construction:
      * per-rank device time per step (events of each rank's own stream around its timed steps, no barrier inside): a slow rank
        or a slow GPU shows as the spread, which the MAX alone hides;
      * one flat fp32 all-reduce of the model's whole gradient (44.05 M elements = 176 MB at config B), timed on its own: what the
        training step's exchange costs when nothing overlaps it, as algorithmic / bus GB/s against the ~153 GB/s of one xGMI link;
      * train mode: three more steps with the bucket all-reduces stamped by events on the side stream (parallel.BUCKET_TIMING):
        time of the collectives per step, the part still running after the backward sweep has finished, the fraction hidden."""
    dev_ms = marks[0].elapsed_time(marks[-1]) / max(args.steps, 1)
    per_rank = parallel.gather_scalars(dev_ms, device)
    out = {"world": dist.get_world_size(), "backend": dist.get_backend(),
           "per_rank_device_ms_per_step": [round(v, 4) for v in per_rank],
           "ms_per_step_rank_min": round(min(per_rank), 4), "ms_per_step_rank_max": round(max(per_rank), 4)}
    numel = sum(p.numel() for p in wl["glow"].parameters() if p.requires_grad and p is not wl["glow"].h_top)
    out["gradient_allreduce_flat"] = parallel.timed_allreduce(numel, device)
    if args.mode == "train":
        loop = wl["loop"]
        loop.flush()
        parallel.BUCKET_TIMING = records = []
        try:
            for _ in range(3):
                loop.step(wl["x"])
            loop.flush()
            torch.cuda.synchronize()
        finally:
            parallel.BUCKET_TIMING = None
        out["gradient_bucket_overlap"] = parallel.bucket_overlap_report(records)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="B", help="BASELINE.json workload: B = configs[1] (headline), "
                    "D = configs[3] (128x128 L=4 K=48), E = configs[4] (256x256 L=6 K=32)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the config's: 64 / 32 / 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="forward mode: issue the kernel list eagerly every step instead of "
                    "replaying the captured hipGraph")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the same step on the exact-fp32 MFMA kernels that the headline "
                    "run times for the record (profiling runs: keeps the kernel table to the product path)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` workloads (configs D / E, sampling, training "
                    "step) that the default one-GPU headline run appends to its JSON line")
    ap.add_argument("--no-repack", action="store_true", help="inference mode: keep derived parameter data across steps")
    ap.add_argument("--repack", action="store_true", help="inverse mode: re-derive the parameter data (W^-1, weight images) every step "
                    "as well (default there: once -- sampling from a trained model)")
    ap.add_argument("--mode", choices=["forward", "inverse", "train"], default="forward",
                    help="forward = the headline metric (Glow.normal_flow); inverse = Glow.reverse_flow sampling throughput; "
                         "train = full training step (fwd with tape + HIP backward + RCCL gradient all-reduce + clip + Adam)")
    ap.add_argument("--debug-flags", default="0", help="glowhip_debug_force_tail_tile value for kernel-variant A/B runs; the JSON line "
                    "is then marked `debug_flags` and its metric `[debug run]` -- never part of a headline run")
    ap.add_argument("--dry-run-cpu", action="store_true", help="launcher / rendezvous / timing-protocol check without a GPU: the "
                    "ranks join a gloo group and time empty steps (tests/test_host.py); prints a line marked dry_run")
    args = ap.parse_args()

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and ("RANK" not in os.environ or env_world == 1):
        # not under a launcher: become the launcher.  Nothing in this process has touched a GPU (torch is not even imported).
        sys.exit(spawn_ranks(args.gpus))
    assert env_world == args.gpus, f"--gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}"

    import torch
    import torch.distributed as dist
    if env_world > 1 and os.environ.get("OMP_NUM_THREADS"):
        torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))

    rank = int(os.environ.get("RANK", "0"))
    world = env_world
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    assert torch.cuda.device_count() > local_rank, f"rank {rank}: no GPU {local_rank} (device_count={torch.cuda.device_count()})"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # GLOWHIP_BENCH_FORCE_DIST=1 with --gpus 1: a ONE-rank RCCL group, and the exchange code of the N-rank run executed over it (the
    # gradient buckets through the side-stream all-reduce, the timed flat all-reduce, the per-rank gather) -- so that a one-GPU box
    # runs every line of the multi-rank path; the line is marked `forced_one_rank_group` and its metric `[diagnostic run]`
    forced = world == 1 and os.environ.get("GLOWHIP_BENCH_FORCE_DIST") == "1"
    if forced:
        os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()))
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # (generous: ranks != 0 sit in the step-0 barrier while rank 0 runs the data-dependent init pass)
        dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=30))
        assert dist.get_world_size() == args.gpus, f"{dist.get_world_size()} ranks joined, --gpus {args.gpus}"

    import pytorch_glow_amd as G
    from pytorch_glow_amd.misc import util
    from pytorch_glow_amd import parallel

    # kernel-variant A/B runs only (scripts/ab_flags.sh): an explicit option, and the line says so -- a headline run takes no debug flags
    dbg = int(args.debug_flags, 0)
    if dbg:
        G.lib().glowhip_debug_force_tail_tile(dbg)
    cfg = CONFIGS[args.config]
    B = args.batch or cfg["batch"]
    repack = args.repack if args.mode == "inverse" else not args.no_repack
    parallel.FORCE_EXCHANGE = forced
    wl = setup_workload(G, util, parallel, device, args.config, args.mode, B, rank, world, repack,
                        graph=not args.no_graph and not dbg and not (forced and args.mode == "train"))
    glow, hps, plan, x, step = wl["glow"], wl["hps"], wl["plan"], wl["x"], wl["step"]
    z_top = wl.get("z_top")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # per-step stamps: events on torch's current stream = the stream every kernel of the step is launched on; no host sync
    # inside the timed region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    c0, m0 = time.process_time(), time.thread_time()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    dt_host = time.perf_counter() - t0      # host time to ENQUEUE the steps (how far the CPU runs ahead of the GPU)
    dt_cpu, dt_main = time.process_time() - c0, time.thread_time() - m0     # CPU time burnt doing so: the whole process (HIP runtime
                                                                            # threads included) | this thread (Python + launches)
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_images = world * B * args.steps
    value = total_images / dt
    multi = None
    if world > 1 or forced:
        multi = multi_rank_diagnostics(torch, dist, parallel, device, args, wl, marks, world, rank, forced)
    if "loop" in wl:      # how the timed training steps were launched
        lp = wl["loop"]
        wl["launch"] = ("hipGraph (training step captured after the eager warm-up steps, one graph launch per step)" if lp._graphed is not None
                        else "eager (two C calls per step issuing the kernel lists + the optimiser's two launches)"
                        + (f"; hipGraph capture failed: {lp.graph_error}" if lp.graph_error else ""))
        lp.graph = False

    if rank == 0:
        fpi = flop_per_image(glow)
        what = {"forward": "fwd+logdet", "inverse": "inverse (reverse_flow sampling)",
                "train": "training step (fwd+bwd+allreduce+clip+Adam)"}[args.mode]
        if args.mode == "forward" and args.config == "B":
            metric = HEADLINE_METRIC
        else:
            metric = f"images/sec Glow {what}, {cfg['image']}x{cfg['image']}x3 L={cfg['L']} K={cfg['K']} [secondary metric]"
        if dbg:
            metric += f" [debug run, flags {dbg:#x}]"
        if forced:
            metric += " [diagnostic run: one-rank RCCL group, exchange path forced]"
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (2xf16 split operands, fp32 accumulate)", "data": "synthetic",
            "ms_per_step_min": round(per_step[0], 4), "ms_per_step_median": round(per_step[len(per_step) // 2], 4),
            "ms_per_step_max": round(per_step[-1], 4), "host_enqueue_ms_per_step": round(1e3 * dt_host / args.steps, 4),
            "host_cpu_ms_per_step": round(1e3 * dt_cpu / args.steps, 4),     # what a rank costs its host: eight ranks share the node's cores
            "host_cpu_main_thread_ms_per_step": round(1e3 * dt_main / args.steps, 4),
            "rccl_world_size": dist.get_world_size() if world > 1 else 1, "launch": wl["launch"],
            "data_dependent_init_ms": wl["init_ms"],     # first training-mode forward (ActNorm statistics layer by layer + one forward), once
            "step0_exchange_ms": wl["step0"],            # rank 0: init pass | wait of the others in the barrier | flat parameter broadcast
            "arithmetic": "fp32 values carried as 2 x f16 (hi, lo), exact f16 products, fp32 accumulate (csrc/sh.h); "
                          "max-abs vs CPU reference 7e-6 (z), same as the exact-fp32 kernels",
            "config": {"workload": f"{cfg['label']}, {what}, batch {B}/GPU ({cfg['ref']})", "global_batch": world * B,
                       "parallelism": f"dp{world}", "repack_every_step": repack,
                       "loss_mean_nll_bits_per_dim": round(float(loss) / (world * B), 6) if args.mode != "inverse" else None},
            "model_tflops": round(value * fpi / 1e12, 2),
            "frac_of_fp32_mfma_peak_whole_model": round(value / world * fpi / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
            "frac_of_split_f16_peak_whole_model": round(value / world * fpi / (PEAK_SPLIT_TFLOPS * 1e12), 4),
        }
        if multi is not None:
            out["multi_rank"] = multi
        if forced:
            out["forced_one_rank_group"] = True
        if args.mode == "train":
            out.pop("model_tflops"), out.pop("frac_of_fp32_mfma_peak_whole_model"), out.pop("frac_of_split_f16_peak_whole_model")
            out["model_tflops_fwd_bwd"] = round(value * 3 * fpi / 1e12, 2)   # backward ~ 2x forward flops
            print(json.dumps(out), flush=True)
            if world > 1 or forced:
                dist.barrier()
                dist.destroy_process_group()
            return
        # ---- roofline of the dominant kernel: instrumented pass of the same step, HIP events per launch (rank-local: NO collective
        # here, the other ranks are already past the timed loop)
        if args.mode == "inverse":
            run_once = lambda: glow.reverse_flow(z_top, None, eps_std=0.7)
        else:
            run_once = lambda: glow.normal_flow(x, None, repack=repack)
        inst = instrumented_pass(plan, hps, B, run_once, with_traffic=(args.config == "B" and B == 64 and args.mode == "forward"))
        out["roofline"] = inst["roofline"]
        bd, launches, variants, sh_path = inst["breakdown"], inst["launches"], inst["variants"], inst["sh_path"]
        if args.mode == "forward" and sh_path and args.config == "B" and not args.no_exact_leg:
            # the same step on the exact-fp32 MFMA kernels (split-half path switched off), for the record; rank-local
            plan.set_family(plan.FAMILY_EXACT_FP32)       # a property of this plan (glowhip_plan_set_family), nothing process-wide
            try:
                for _ in range(2):
                    glow.normal_flow(x, None, repack=repack)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    glow.normal_flow(x, None, repack=repack)
                torch.cuda.synchronize()
                dt1 = (time.perf_counter() - t1) / 5
            finally:
                plan.set_family(plan.FAMILY_AUTO)
            out["exact_fp32_mfma_kernels"] = {"value": round(B / dt1, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dt1, 4),
                                              "note": "one GPU, same step with v_mfma_f32_32x32x2_f32 kernels only"}
        if args.mode == "forward":
            # how often the fp16-pair range guard fires: the timed steps are UNCHECKED calls (an overflow would surface as a non-finite
            # loss: `loss_finite`); one more, CHECKED forward of the same batch (Glow.forward under no_grad / eval: reads the status
            # back and re-runs an out-of-range batch on the exact-fp32 kernels) counts the fall-backs of this workload
            before = type(glow)._RANGE_FALLBACKS
            was_training = glow.training
            glow.eval()
            with torch.no_grad():
                glow(x)
            glow.train(was_training)
            torch.cuda.synchronize()
            out["range_fallbacks"] = {"checked_forward_of_the_timed_batch": type(glow)._RANGE_FALLBACKS - before,
                                      "loss_finite_in_timed_region": bool(torch.isfinite(torch.as_tensor(float(loss))))}
        out["breakdown_ms_per_step"] = bd
        out["breakdown_sum_ms"] = round(sum(bd.values()), 3)
        out["kernel_launches_per_step"] = launches
        out["k_cnet_instances_per_step"] = variants
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(glow, x[:min(B, cfg["cpu_sample"])].cpu(), cfg, fallback_batch=max(1, min(B, cfg["cpu_sample"]) // 2))
        if world == 1 and args.mode == "forward" and args.config == "B" and not args.no_secondary and not dbg:
            del glow, plan, wl, step
            torch.cuda.empty_cache()
            out["secondary"] = secondary_workloads(G, util, parallel, device)
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
