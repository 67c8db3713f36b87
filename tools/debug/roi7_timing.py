"""per-ROI 7x7 sampler + entropy at cfg4 size: K0 (table) and K1 separately."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from runia_core_amd import _hip
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (K, C, H, W, n_mc, bs) in ((60000, 256, 7, 7, 16, 3), (60000, 256, 8, 8, 16, 3), (60000, 256, 4, 4, 16, 2), (30000, 256, 7, 7, 32, 3), (30000, 256, 8, 8, 32, 3)):
    if not _hip.mc_entropy_supported(H, W, n_mc, 5):
        print((K, C, H, W, n_mc), "unsupported"); continue
    x = torch.relu(torch.randn(K, C, H, W, device=dev, generator=g)).contiguous()
    cd = _hip.CounterDraws(3, 0)
    tab = _hip.mc_mask_table(cd, K, H, W, n_mc, 0.4, bs)
    t0 = t(lambda: _hip.mc_mask_table(cd, K, H, W, n_mc, 0.4, bs, out=tab))
    h = torch.empty(K, C, dtype=torch.float64, device=dev)
    t1 = t(lambda: _hip.mc_entropy(x, None, n_mc, 0.4, bs, 5, out=h, table=tab))
    items = K * C
    by = (H * W * 4 + 8) * items
    print(f"K={K} C={C} {H}x{W} n_mc={n_mc}: K0 {t0*1e3:.1f} us, K1 {t1*1e3:.1f} us = {items/t1/1e6:.2f} G items/s = {by/t1/1e9:.2f} TB/s ({by/t1/1e9/8:.1%} of HBM)", flush=True)
