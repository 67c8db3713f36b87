import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n, dt in ((1_000_000, torch.float64), (1_000_000, torch.float32), (10_000, torch.float64), (100_000, torch.float64)):
    ind = (torch.randn(n, dtype=torch.float64, device=dev, generator=g) + 0.4).to(dt)
    ood = (torch.randn(n, dtype=torch.float64, device=dev, generator=g) - 0.4).to(dt)
    ms = t(lambda: _hip.ood_metrics(ind, ood))
    print(f"metrics {n}+{n} {dt}: {ms*1e3:.1f} us = {2*n/ms/1e6:.2f} G scores/s", flush=True)
