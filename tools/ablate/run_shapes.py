"""Fused sampler + entropy (K0 + K1) on other map shapes, GPU at working clocks."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
gc.disable(); torch.manual_seed(0)
for (n, c, h, w, n_mc, bs) in ((10000, 512, 4, 4, 16, 2), (4000, 512, 8, 8, 16, 3), (4000, 512, 7, 7, 16, 3), (10000, 2048, 2, 2, 16, 1), (10000, 512, 4, 4, 32, 2), (10000, 512, 4, 4, 12, 2)):
    x = torch.relu(torch.randn(n, c, h, w, device="cuda")); r = torch.rand(n, n_mc, h, w, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        _hip.mc_entropy(x, r, n_mc, 0.5, bs, 5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _hip.mc_entropy(x, r, n_mc, 0.5, bs, 5)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = (x.numel() * 4 + r.numel() * 4 + n * c * 8) / 1e9
    print(f"{n}x{c}x{h}x{w} n_mc {n_mc}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s  {n / ms * 1e-3:.2f} M images/s")
