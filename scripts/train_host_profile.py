"""Debug: where the HOST time of one config-B training step goes (cProfile of TrainLoop.step, GPU work asynchronous).
Usage (GPU box): python scripts/train_host_profile.py [steps]"""
import cProfile, os, pstats, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
from pytorch_glow_amd import parallel, training
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
wl = bench.setup_workload(G, util, parallel, dev, "B", "train", 64, 0, 1, repack=False)
step = wl["step"]
for _ in range(8):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {1e3 * t_host / steps:.2f} ms/step, wall {1e3 * t_all / steps:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
