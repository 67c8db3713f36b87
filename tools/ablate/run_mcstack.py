import sys, os, gc
sys.path.insert(0, os.getcwd())
import torch
from runia_core_amd import _hip
gc.disable()
torch.manual_seed(0)
N, C, H, W, NMC = 10000, 512, 4, 4, 16
x = torch.randn(N, C, H, W, device="cuda")
rand = torch.rand(N, NMC, H, W, device="cuda")
for _ in range(20): z = _hip.mc_stack(x, rand, NMC, 0.5, 2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): z = _hip.mc_stack(x, rand, NMC, 0.5, 2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("mc_stack %.4f ms  %.1f GB/s" % (ms, N * (C * H * W * 4 + NMC * H * W * 4 + NMC * C * 4) / ms * 1e-6))
for _ in range(20): h = _hip.kl_entropy_per_dim(z, NMC, 5)
torch.cuda.synchronize()
