import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable()
N, D, C = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (262144, 2048, 10)
g = torch.Generator(device="cuda").manual_seed(3)
centres = torch.randn(C, D, device="cuda", generator=g) * 0.5
lab = torch.randint(0, C, (N,), device="cuda", generator=g)
f = torch.relu(centres[lab] + torch.randn(N, D, device="cuda", generator=g))
a = torch.randn(D, D, dtype=torch.float64, device="cuda", generator=g)
prec = (a @ a.T / D + torch.eye(D, dtype=torch.float64, device="cuda")).contiguous()
cm = centres.contiguous()
packed = _hip.pack_weights(prec)
mu_p = (cm.double() @ prec).contiguous()
for _ in range(2): s = _hip.mahalanobis_score(f, cm, packed, mu_p)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): s = _hip.mahalanobis_score(f, cm, packed, mu_p)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"N {N} D {D} C {C}: " + "Mahalanobis %.2f ms  %.1f TFLOP/s (2 D^2 per row)  checksum %.6e" % (ms, 2.0 * N * D * D / ms * 1e-9, float(s.sum())))
