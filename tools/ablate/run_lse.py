import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from runia_core_amd import _hip
gc.disable()
for n, c in ((1_000_000, 10), (10_000, 10), (1_000_000, 16), (1_000_000, 7)):
    x = torch.randn(n, c, device="cuda") * 3
    for _ in range(50): _hip.row_lse_msp(x, True, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): _hip.row_lse_msp(x, True, True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 5
    print(f"{n}x{c}: {us:.1f} us  {n * (4 * c + 8) / us / 1e3:.0f} GB/s")
