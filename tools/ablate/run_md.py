import sys, os, gc
sys.path.insert(0, os.getcwd())
import torch
from runia_core_amd import _hip
gc.disable()
torch.manual_seed(0)
N, n = 10000, 256
y = torch.randn(N, n, dtype=torch.float64, device="cuda")
mean = torch.randn(n, dtype=torch.float64, device="cuda")
a = torch.randn(n, n, dtype=torch.float64, device="cuda")
pp = _hip.pack_weights((a @ a.T).contiguous())
for _ in range(30): s = _hip.md_score(y, mean, pp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): s = _hip.md_score(y, mean, pp)
e1.record(); torch.cuda.synchronize()
print("md_score %.4f ms" % (e0.elapsed_time(e1) / 100))
y32 = y.float(); m32 = mean.float()
for _ in range(30): s = _hip.md_score(y32, m32, pp)
torch.cuda.synchronize()
