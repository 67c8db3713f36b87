import torch, sys, os
sys.path.insert(0, os.getcwd())
import runia_core_amd._hip as _hip
if os.environ.get('RUNIA_LIB'):
    import ctypes
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
import numpy as np
torch.manual_seed(0)
N, D, r = 10000, 512, 256
h = torch.randn(N, D, dtype=torch.float64, device="cuda")
M = torch.randn(D, r, dtype=torch.float64, device="cuda") * 0.05
c = torch.randn(r, dtype=torch.float64, device="cuda")
pm = _hip.pack_weights(M)
out = torch.empty(N, dtype=torch.float64, device="cuda")
ref = -((h @ M + c) ** 2).sum(1)
for _ in range(3): _hip.proj_sq_score(h, pm, c, r, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): _hip.proj_sq_score(h, pm, c, r, out=out)
e1.record(); torch.cuda.synchronize()
print("RT", os.environ.get("RUNIA_PROJ_RT"), "%.1f us" % (e0.elapsed_time(e1) * 20), "err", float(((out - ref).abs() / ref.abs()).max()))
