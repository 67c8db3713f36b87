import gc, os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
gc.disable(); torch.manual_seed(0)
a = torch.randn(1_000_000, dtype=torch.float64, device="cuda") + 1.0
b = torch.randn(1_000_000, dtype=torch.float64, device="cuda")
for _ in range(10): r = _hip.ood_metrics(a, b)
torch.cuda.synchronize()
