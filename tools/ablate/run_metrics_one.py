import os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ind = torch.randn(n, dtype=torch.float64, device=dev, generator=g) + 0.4
ood = torch.randn(n, dtype=torch.float64, device=dev, generator=g) - 0.4
for _ in range(20): _hip.ood_metrics(ind, ood)
torch.cuda.synchronize()
