"""GMM / DDU scoring: the triangular f32 kernel (round 6, runia_gmm_log_prob_f32) against the dense f64 per-class form of rounds 4-5.
    python tools/ablate/run_gmm.py [N D C]      (default 262144 2048 10: the f4.gmm_ddu leg of bench.py)"""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
from runia_core_amd.inference.funcs import GmmState
gc.disable()
N, D, C = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (262144, 2048, 10)
DENSE = os.environ.get("GMM_DENSE", "1") == "1"
g = torch.Generator(device="cuda").manual_seed(3)
loc = torch.randn(C, D, device="cuda", generator=g) * 0.5
tril = []
for c in range(C):
    a = torch.randn(D, 2 * D, dtype=torch.float64, device="cuda", generator=g)
    tril.append(torch.linalg.cholesky(a @ a.T / (2 * D) + 0.05 * torch.eye(D, dtype=torch.float64, device="cuda")).float().cpu())
    del a
gmm = torch.distributions.MultivariateNormal(loc=loc.cpu(), scale_tril=torch.stack(tril))
lab = torch.randint(0, C, (N,), device="cuda", generator=g)
x = (loc[lab] + torch.randn(N, D, device="cuda", generator=g)).contiguous()


def timed(fn, reps=3):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


st = GmmState(gmm)
ms, lse = timed(lambda: st.energy_device(x))
print(f"N {N} D {D} C {C}: triangular f32  {ms:9.3f} ms   {1e-9 * N * C * D * D / ms:7.1f} TFLOP/s of D^2 per (row, class)   "
      f"checksum {float(lse.double().sum()):.9e}")
m = min(N, 64)
want = torch.logsumexp(gmm.log_prob(x[:m].cpu()[:, None, :]), dim=1).numpy()
want64 = torch.logsumexp(torch.distributions.MultivariateNormal(loc=gmm.loc.double(), scale_tril=gmm.scale_tril.double()
                                                                ).log_prob(x[:m].cpu().double()[:, None, :]), dim=1).numpy()
rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
print(f"   first {m} rows: vs torch f32 {rel(lse[:m].cpu().numpy(), want):.2e}   vs torch f64 {rel(lse[:m].cpu().numpy(), want64):.2e}   "
      f"(torch f32 vs f64 {rel(want, want64):.2e})")
if DENSE:
    sd = GmmState(gmm, dense=True)
    ms_d, lse_d = timed(lambda: sd.energy_device(x), reps=2)
    print(f"   dense f64 per class {ms_d:9.3f} ms   {1e-9 * N * C * 2 * D * D / ms_d:7.1f} TFLOP/s of 2 D^2   x{ms_d / ms:.2f}   "
          f"vs torch f32 {rel(lse_d[:m].cpu().numpy(), want):.2e}  vs f64 {rel(lse_d[:m].cpu().numpy(), want64):.2e}")
