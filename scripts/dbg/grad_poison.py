"""Debug: does the backward ACCUMULATE into its (torch.empty) gradient buffers?  Poison freed memory with 1e3 before the call."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import pytorch_glow_amd as G
from test_host import _g9_hps
from pytorch_glow_amd.misc import util
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
if which == "tiny":
    hps = _g9_hps(); hps.optim.num_batch_train = 4; img = 16; B = 4
else:
    hps = util.load_profile("celeba"); hps.model.K = 2; hps.optim.num_batch_train = 8; hps.device.graph = ["cuda:0"]; img = 64; B = 8
glow = G.Glow(hps).to("cuda:0")
x = torch.rand(B, 3, img, img, device="cuda:0")
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)
with torch.no_grad():
    for n, p in glow.named_parameters():
        if ".f.4." in n or "conv2d_zeros" in n:
            p.copy_(torch.randn_like(p) * 0.02)
def grads(poison):
    for p in glow.parameters(): p.grad = None
    glow.flow.pop_grad_buckets()
    if poison is not None:
        for sz in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
            t = [torch.full((sz,), poison, device="cuda") for _ in range(4)]
            del t
    with torch.enable_grad():
        _, nll_, _ = glow.normal_flow(x, None, noise=torch.zeros_like(x))
        glow.generative_loss(nll_).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in glow.named_parameters() if p.grad is not None}
a = grads(None); b = grads(1e3); c = grads(float("nan"))
bad = [(n, float((a[n] - b[n]).abs().max()), float(a[n].abs().max())) for n in a if not torch.equal(a[n], b[n])]
print(which, "tensors that differ after poisoning with 1e3:", len(bad), "of", len(a)); print(bad[:12])
print("tensors with NaN after NaN poison:", [n for n in c if not torch.isfinite(c[n]).all()][:12])
