"""Debug: accumulated phase times of the four waves of workgroup 0 of the last f.2 weight-gradient launch (stamps build:
make BUILD=build_stamps LIB=../libglowhip_stamps.so EXTRA=-DGLOWHIP_DEBUG_STAMPS): cycles from step start to the last MFMA issued
(requests in between), through the split + LDS stores of the next tile, in the barrier, and between steps."""
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
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)          # ActNorm init
from pytorch_glow_amd import training
tl = training.TrainLoop(glow, hps)
for _ in range(3): tl.step(x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
G.lib().glowhip_debug_read_stamps_all_wgrad(buf)
t = [list(buf[w * 64:(w + 1) * 64]) for w in range(4)]
nk = t[0][4]
print(f"k-tiles per workgroup: {nk}")
print("wave:      " + " ".join(f"{w:9d}" for w in range(4)))
print("SIMD/slot: " + " ".join(f"    {(t[w][63] >> 4) & 3}/{t[w][63] & 15:<3d}" for w in range(4)))
for i, n in ((3, "between steps"), (0, "MFMAs + requests issued"), (5, "wait for the next tile"), (1, "split + LDS stores"), (2, "barrier")):
    print(f"{n:26s}" + " ".join(f"{t[w][i] / max(nk, 1):9.0f}" for w in range(4)) + "   cycles per k-tile")
print(f"{'sum':26s}" + " ".join(f"{(sum(t[w][:4]) + t[w][5]) / max(nk, 1):9.0f}" for w in range(4)))
