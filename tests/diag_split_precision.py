"""Diagnostic (not collected): numerical effect of evaluating the coupling-net convolutions with 2 x fp16 split operands
(hi = fp16(x), lo = fp16((x - hi) * 2048); y = hi*hi + (hi*lo + lo*hi) / 2048, fp32 accumulate) on the full celeba64 model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import glow_oracle as O
torch.set_grad_enabled(False)
B = int(os.environ.get("B", "2")); K = int(os.environ.get("K", "32")); MODE = os.environ.get("MODE", "f16x2")
cfg = O.default_cfg(K=K, batch=B)
sd = O.seeded_state_dict(cfg, invconv_perturb=float(os.environ.get("PERTURB", "0.0")))
g = torch.Generator().manual_seed(3)
x = torch.rand(B, 3, 64, 64, generator=g); noise = torch.rand(B, 3, 64, 64, generator=g) / 256
sd = O.glow_init_actnorm(x, noise, sd, cfg)
orig = F.conv2d
def split16(t):
    hi = t.half().float(); lo = ((t - hi) * 2048.0).half().float(); return hi, lo
def splitb(t):
    hi = t.bfloat16().float(); lo = (t - hi).bfloat16().float(); return hi, lo
def conv_split(inp, w, b=None, **kw):
    if w.shape[2] == 1 and w.shape[0] == w.shape[1] and w.shape[0] <= 64:   # invconv stays exact (k_chanmix)
        return orig(inp, w, b, **kw)
    if MODE == "f16x2":
        xh, xl = split16(inp); wh, wl = split16(w)
        y = orig(xh, wh, None, **kw) + (orig(xh, wl, None, **kw) + orig(xl, wh, None, **kw)) / 2048.0
    elif MODE == "bf16x2":
        xh, xl = splitb(inp); wh, wl = splitb(w)
        y = orig(xh, wh, None, **kw) + (orig(xh, wl, None, **kw) + orig(xl, wh, None, **kw))
    elif MODE == "f16x1":
        y = orig(inp.half().float(), w.half().float(), None, **kw)
    return y if b is None else y + b.view(1, -1, 1, 1)
t = time.time()
z32, nll32, _ = O.glow_forward(x, noise, sd, cfg)
sd64 = {k: v.double() for k, v in sd.items()}
z64, nll64, _ = O.glow_forward(x.double(), noise.double(), sd64, cfg)
O.F.conv2d = conv_split
zs, nlls, _ = O.glow_forward(x, noise, sd, cfg)
O.F.conv2d = orig
print(f"mode {MODE} B={B} K={K}  ({time.time()-t:.1f}s)")
print("fp32 oracle vs fp64 : z", float((z32 - z64).abs().max()), "nll", float((nll32 - nll64).abs().max()))
print("split       vs fp64 : z", float((zs - z64).abs().max()), "nll", float((nlls - nll64).abs().max()))
print("split       vs fp32 : z", float((zs - z32).abs().max()), "nll", float((nlls - nll32).abs().max()))
