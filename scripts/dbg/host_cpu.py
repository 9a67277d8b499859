"""Debug: CPU time (main thread | whole process) per config-B training step, with and without the deferred range check."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
if os.environ.get("BLOCKSYNC"):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(4)))      # hipDeviceScheduleBlockingSync
import pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
from pytorch_glow_amd import parallel, training
import bench
dev = torch.device("cuda:0")
wl = bench.setup_workload(G, util, parallel, dev, "B", "train", 64, 0, 1, repack=False)
loop, x = wl["loop"], wl["x"] if "x" in wl else None
step = wl["step"]
def measure(tag, n=12):
    for _ in range(6): step()
    torch.cuda.synchronize()
    t0, c0, m0 = time.perf_counter(), time.process_time(), time.thread_time()
    for _ in range(n): step()
    t1, c1, m1 = time.perf_counter(), time.process_time(), time.thread_time()
    torch.cuda.synchronize()
    print(f"{tag}: host wall {1e3*(t1-t0)/n:.2f} ms/step, process cpu {1e3*(c1-c0)/n:.2f}, main thread {1e3*(m1-m0)/n:.2f}")
measure("range check on")
loop.flush(); loop.range_check = False
measure("range check off")
