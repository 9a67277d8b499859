"""Debug: cycle stamps of one workgroup (all four waves) of the LAST k_cnet1w launch of a forward (stamps build only):
make -C pytorch-glow_amd/csrc BUILD=build_stamps LIB=../libglowhip_stamps.so EXTRA=-DGLOWHIP_DEBUG_STAMPS
env: K, L (model depth), B (batch), FLAGS (glowhip_debug_force_tail_tile), TRAIN=1 (the taping instance), TRAIN=2 (the backward instance)."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["GLOWHIP_LIB_PATH"] = os.path.join(root, "pytorch-glow_amd", os.environ.get("STAMPLIB", "libglowhip_stamps.so"))
import torch
import pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
K = int(os.environ.get("K", "1")); L = int(os.environ.get("L", "1")); B = int(os.environ.get("B", "64"))
hps = util.load_profile("celeba"); hps.model.K = K; hps.model.L = L; hps.optim.num_batch_train = B; hps.device.graph = ["cuda:0"]
glow = G.Glow(hps).to("cuda:0")
x = torch.rand(B, 3, 64, 64, device="cuda")
fl = int(os.environ.get("FLAGS", "0"), 0)
if fl: G.lib().glowhip_debug_force_tail_tile(fl)
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)
if os.environ.get("TRAIN") == "2":   # the backward instance: the last k_cnet1w launch of a training step is its level-1 input-gradient launch
    with torch.enable_grad():
        for _ in range(3):
            z, nll, _ = glow.normal_flow(x, None)
            G.Glow.generative_loss(nll).backward()
elif os.environ.get("TRAIN"):        # the taping instance: training forwards (the tape is written when a graph is recorded)
    with torch.enable_grad():
        for _ in range(3): glow.normal_flow(x, None)
else:
    with torch.no_grad():
        glow.eval()
        for _ in range(3): glow.normal_flow(x, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 64))()
G.lib().glowhip_debug_read_stamps_all_cnet1w(buf)
names = {0: "start", 1: "window + tables built, first fills landed", 2: "f.0 of chunk 0", 3: "f.2 (+ f.0 inside) done", 4: "f.4 done", 5: "T staged", 6: "end"}
names.update({10 + c: f"  chunk {c} top" for c in range(16)})
for w in range(4):
    t = list(buf[w * 64:(w + 1) * 64])
    print(f"wave {w}")
    prev = t[0]
    for i in sorted(names, key=lambda i: (t[i] if t[i] >= t[0] else 0, i)):
        if t[i] >= t[0] and t[i] != 0:
            print(f"  {names[i]:44s} +{t[i] - prev:7d}   (t = {t[i] - t[0]:7d})")
            prev = t[i]
