"""Debug: per-workgroup start / end times (s_memrealtime, 100 MHz) and placement of the LAST k_cnet launch of a forward (stamps
build, see stamps_cnet.py): how long the launch ramps up, what a CU does between its first and its second workgroup, how far apart
the workgroups finish.  env: K, L, B, FLAGS as stamps_cnet.py; NWG = workgroups of that launch (default 512: level 1 at B = 64)."""
import ctypes, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["GLOWHIP_LIB_PATH"] = os.path.join(root, "pytorch-glow_amd", os.environ.get("STAMPLIB", "libglowhip_stamps.so"))
import torch
import pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
K = int(os.environ.get("K", "1")); L = int(os.environ.get("L", "1")); B = int(os.environ.get("B", "64"))
NWG = int(os.environ.get("NWG", "512"))
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
buf = (ctypes.c_ulonglong * (4 * NWG))()
getattr(G.lib(), "glowhip_debug_read_wgtimes_" + os.environ.get("KERNEL", "cnet"))(buf, NWG)
t = np.array(list(buf), dtype=np.int64).reshape(NWG, 4)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) * 10.0, (t[:, 1] - t0) * 10.0        # ns
hw, xcc = t[:, 2], t[:, 3] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 16 + cu
print(f"workgroups {NWG}; launch span {end.max() / 1e3:.2f} us; distinct CUs {len(set(key.tolist()))}")
dur = end - start
print(f"workgroup duration us: min {dur.min() / 1e3:.2f} median {np.median(dur) / 1e3:.2f} max {dur.max() / 1e3:.2f}")
order = np.argsort(start)
print(f"starts us: first 256 within {np.sort(start)[min(255, NWG - 1)] / 1e3:.2f}; last start {start.max() / 1e3:.2f}")
print(f"ends us: first {end.min() / 1e3:.2f} median {np.median(end) / 1e3:.2f} last {end.max() / 1e3:.2f}")
gaps, firsts, seconds = [], [], []
for k in set(key.tolist()):
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(start[idx])]
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append(start[b] - end[a]); firsts.append(dur[a]); seconds.append(dur[b])
if gaps:
    gaps = np.array(gaps)
    print(f"same-CU gap end -> next start us: min {gaps.min() / 1e3:.2f} median {np.median(gaps) / 1e3:.2f} max {gaps.max() / 1e3:.2f}")
    print(f"first workgroup of a CU {np.median(firsts) / 1e3:.2f} us median, later ones {np.median(seconds) / 1e3:.2f} us")
per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
print("workgroups per CU:", dict(zip(*np.unique(per_cu, return_counts=True))))
for x_ in range(8):
    m = xcc == x_
    if m.any(): print(f"  XCC {x_}: {m.sum()} wgs, mean duration {dur[m].mean() / 1e3:.2f} us, last end {end[m].max() / 1e3:.2f} us")
