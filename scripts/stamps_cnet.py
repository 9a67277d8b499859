"""Debug: s_memtime stamps of one workgroup (wave 0) of the LAST k_cnet launch of a forward (stamps build only):
make -C pytorch-glow_amd/csrc BUILD=build_stamps LIB=../libglowhip_stamps.so EXTRA=-DGLOWHIP_DEBUG_STAMPS
env: K, L (model depth), B (batch), FLAGS (glowhip_debug_force_tail_tile)."""
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
    glow.eval()
    for _ in range(3): glow.normal_flow(x, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
G.lib().glowhip_debug_read_stamps_cnet(buf)
t = list(buf)
names = {0: "start", 22: "pre: set up", 23: "pre: coupled", 24: "pre: barrier", 25: "pre: mixed", 26: "pre: log-det summed", 1: "window built", 2: "P1a done", 3: "barrier", 4: "P2a done", 5: "barrier", 6: "P1b done", 7: "barrier", 8: "P2b done",
         9: "barrier", 10: "h2 epilogue", 11: "h2 load0 written", 12: "barrier", 13: "P3 load0 done", 14: "barrier", 15: "h2 load1 written",
         16: "barrier", 17: "P3 load1 done", 18: "barrier", 20: "T staged", 21: "end"}
prev = t[0]
for i in sorted(names, key=lambda i: (t[i] if t[i] >= t[0] else 0, i)):
    if t[i] >= t[0] and t[i] != 0:
        print(f"{names[i]:20s} +{t[i] - prev:7d}   (t = {t[i] - t[0]:7d})")
        prev = t[i]
