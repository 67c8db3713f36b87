import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
g = torch.Generator(device="cuda").manual_seed(3)
ind = (torch.randn(n, dtype=torch.float64, device="cuda", generator=g) + 0.4).to(dt)
ood = (torch.randn(n, dtype=torch.float64, device="cuda", generator=g) - 0.4).to(dt)
for _ in range(30): out = _hip.ood_metrics(ind, ood)
torch.cuda.synchronize()
print(out.cpu().numpy())
