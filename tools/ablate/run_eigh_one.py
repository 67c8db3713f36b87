import os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
n = int(sys.argv[1])
torch.manual_seed(0)
g = torch.randn(n, n, dtype=torch.float64, device="cuda")
a = (g @ g.T / n + torch.eye(n, dtype=torch.float64, device="cuda")).contiguous()
for _ in range(3): w, v = _hip.eigh(a)
torch.cuda.synchronize()
