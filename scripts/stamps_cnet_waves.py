"""Debug: s_memtime stamps of ALL EIGHT WAVES of one workgroup of the last k_cnet launch (stamps build, see stamps_cnet.py).
Prints, per stamp, each wave's time since the workgroup's first stamp, and each wave's SIMD / wave slot (HW_ID)."""
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
buf = (ctypes.c_ulonglong * 512)()
G.lib().glowhip_debug_read_stamps_all_cnet(buf)
t = [list(buf[w * 64:(w + 1) * 64]) for w in range(8)]
names = {0: "start", 26: "tables+first loads", 1: "window built", 2: "P1a done", 3: "barrier", 4: "P2a (+P1b first MFMAs) done", 5: "barrier", 6: "P1b done",
         7: "barrier", 8: "P2b done", 9: "barrier", 10: "h2 epilogue", 27: "T accumulators zeroed", 11: "h2 load0 written", 12: "barrier", 13: "P3 load0 done",
         14: "barrier", 15: "h2 load1 written", 16: "barrier", 17: "P3 load1 done", 18: "barrier", 28: "rs4 fetched", 29: "T stores issued", 20: "T staged", 21: "end"}
t0 = min(t[w][0] for w in range(8))
print("wave:      " + " ".join(f"{w:7d}" for w in range(8)))
print("SIMD/slot: " + " ".join(f"  {(t[w][63] >> 4) & 3}/{t[w][63] & 15:<3d}" for w in range(8)))
order = sorted((i for i in names if t[0][i] >= t0 and t[0][i] != 0), key=lambda i: t[0][i])
for i in order:
    print(f"{names[i]:28s}" + " ".join(f"{t[w][i] - t0:7d}" for w in range(8)))
