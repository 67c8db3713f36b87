"""PCA transform / MD / K2' with few components (the reference default is nro_components=16)."""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N, D = 10000, 512
h = torch.randn(N, D, dtype=torch.float64, device="cuda")
for n in (16, 32, 64, 128, 256):
    comp = torch.linalg.qr(torch.randn(D, n, dtype=torch.float64, device="cuda"))[0].contiguous()
    pct = _hip.pack_weights(comp)
    bias = torch.randn(n, dtype=torch.float64, device="cuda"); scale = torch.rand(n, dtype=torch.float64, device="cuda") + 0.5
    a = torch.randn(n, n, dtype=torch.float64, device="cuda"); prec = (a @ a.T / n + torch.eye(n, dtype=torch.float64, device="cuda")).contiguous()
    pp = _hip.pack_weights(prec); mean = torch.randn(n, dtype=torch.float64, device="cuda") * 0.1
    y = _hip.pca_transform(h, pct, bias, scale, n)
    us_p = t(lambda: _hip.pca_transform(h, pct, bias, scale, n))
    us_m = t(lambda: _hip.md_score(y, mean, pp))
    us_k2 = t(lambda: _hip.pca_md_score(h, pct, bias, scale, mean, pp, n))
    m = torch.randn(D, n, dtype=torch.float64, device="cuda").contiguous(); c = torch.randn(n, dtype=torch.float64, device="cuda")
    pm = _hip.pack_weights(m); out = torch.zeros(N, dtype=torch.float64, device="cuda")
    us_k2p = t(lambda: _hip.proj_sq_accumulate(h, pm, c, n, out))
    print(f"n={n:4d}: pca_transform {us_p:7.1f} us  md {us_m:7.1f} us  K2 {us_k2:7.1f} us  K2' {us_k2p:7.1f} us", flush=True)
