"""Sampler + entropy on map shapes without a fused kernel (generic LDS sampler + per-dim entropy)."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
for (n, c, h, w, n_mc, bs) in ((2000, 256, 14, 14, 16, 5), (500, 128, 28, 28, 16, 7), (2000, 512, 7, 7, 16, 3), (2000, 1024, 5, 5, 16, 2), (4000, 512, 3, 3, 16, 2), (2000, 64, 16, 16, 16, 4)):
    x = torch.relu(torch.randn(n, c, h, w, device="cuda")); r = torch.rand(n, n_mc, h, w, device="cuda")
    def f():
        z = _hip.mc_stack(x, r, n_mc, 0.5, bs)
        return _hip.kl_entropy_per_dim(z, n_mc, 5)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): z = _hip.mc_stack(x, r, n_mc, 0.5, bs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = (x.numel() * 4 + r.numel() * 4 + n * n_mc * c * 4) / 1e9
    print(f"{n}x{c}x{h}x{w} bs {bs}: sampler {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s  {n / ms * 1e-3:.3f} M images/s   fused-supported {bool(_hip.mc_entropy_supported(h, w, n_mc, 5))}", flush=True)
