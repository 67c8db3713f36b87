"""MD / LaREM on un-reduced features: the triangular-factor kernel (round 6, runia_md_score_tril_*) against the dense P form."""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from runia_core_amd import _hip
gc.disable()
def t(fn, reps=5):
    for _ in range(2): out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
for N, n in ((262144, 512), (262144, 1024), (262144, 2048), (8, 2048), (512, 2048)):
    g = torch.Generator(device="cuda").manual_seed(n)
    a = torch.randn(n, 2 * n, dtype=torch.float64, device="cuda", generator=g)
    prec = torch.linalg.inv(a @ a.T / (2 * n) + 0.05 * torch.eye(n, dtype=torch.float64, device="cuda"))
    prec = (0.5 * (prec + prec.T)).contiguous()
    x = torch.randn(N, n, device="cuda", generator=g)
    mean = torch.randn(n, device="cuda", generator=g)
    G, info = _hip.cholesky(torch.flip(prec, dims=(0, 1)).contiguous())
    pw = _hip.pack_weights(torch.flip(G, dims=(0, 1)).contiguous())
    pp = _hip.pack_weights(prec)
    ms_t, s_t = t(lambda: _hip.md_score_tril(x, mean, pw))
    ms_d, s_d = t(lambda: _hip.md_score(x, mean, pp))
    rel = float(((s_t - s_d).abs() / s_d.abs().clamp_min(1.0)).max())
    print(f"N {N} n {n}: triangular {ms_t:8.3f} ms ({N * n * n * 1e-9 / ms_t:6.1f} TFLOP/s of n^2)   dense {ms_d:8.3f} ms ({2.0 * N * n * n * 1e-9 / ms_d:6.1f} TFLOP/s of 2 n^2)"
          f"   x{ms_d / ms_t:.2f}   max rel diff {rel:.1e}")
