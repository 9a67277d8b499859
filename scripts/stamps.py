"""Debug: s_memtime stamps of workgroup 0 of the LAST k_tail_sh / k_gemm_sh launch of one forward (stamps build only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GLOWHIP_LIB_PATH"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-glow_amd", "" + os.environ.get("STAMPLIB", "libglowhip_stamps.so") + "")
import torch
import bench, pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
K = int(os.environ.get("K", "1")); L = int(os.environ.get("L", "1"))
hps = util.load_profile("celeba"); hps.model.K = K; hps.model.L = L; hps.optim.num_batch_train = 64; hps.device.graph = ["cuda:0"]
glow = G.Glow(hps).to("cuda:0")
x = torch.rand(64, 3, 64, 64, device="cuda")
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)
    glow.eval()
    for _ in range(3): glow.normal_flow(x, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
G.lib().glowhip_debug_read_stamps_tail(buf)
t = list(buf)
G.lib().glowhip_debug_read_stamps_gemm(buf)
t[16:32] = list(buf)[16:32]
def rel(idx, base): return [(t[i] - t[base]) for i in idx]
print("tail:  prologue issued %d | first 8 stage starts %s | loop end %d | after sync %d | T written %d | tap sums done %d" % (
    t[1]-t[0], rel(range(8,16),0), t[2]-t[0], t[3]-t[0], t[4]-t[0], t[5]-t[0]))
G.lib().glowhip_debug_read_stamps_f02(buf)
u = list(buf)
print("f02:   window built %d | phase 1 done %d | after barrier %d | phase 2 loop done %d | end %d" % (u[33]-u[32], u[34]-u[32], u[35]-u[32], u[36]-u[32], u[37]-u[32]))
print("gemm:  prologue issued %d | first 8 stage starts %s | loop end %d | epilogue end %d" % (
    t[17]-t[16], rel(range(24,32),16), t[18]-t[16], t[19]-t[16]))
