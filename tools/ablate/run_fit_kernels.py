#!/usr/bin/env python3
"""The setup-time kernels of round 6's second half on cfg3-sized inputs, for a kernel trace (tools/prof_cmd.sh):
percentile of 102.4 M activations (radix select), Cholesky + triangular inverse of ten 2048 x 2048 matrices (f32 / f64), pinvh."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip  # noqa: E402
from runia_core_amd.device_fit import pinvh_device  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(3)
x = torch.relu(torch.randn(50_000, 2048, device="cuda", generator=g))
n = x.numel()
for _ in range(3):
    lo, hi = _hip.kth_smallest_flat(x, [int(0.9 * (n - 1)), int(0.9 * (n - 1)) + 1])
torch.cuda.synchronize()
t = time.perf_counter()
lo, hi = _hip.kth_smallest_flat(x, [int(0.9 * (n - 1)), int(0.9 * (n - 1)) + 1])
torch.cuda.synchronize()
print(f"two order statistics of {n} f32: {1e3 * (time.perf_counter() - t):.2f} ms ({lo:.6f}, {hi:.6f}); one pass reads {4 * n / 1e6:.0f} MB")
a = torch.randn(10, 2048, 6000, device="cuda", generator=g, dtype=torch.float64)
cov = (a @ a.transpose(1, 2) / 6000 + 0.01 * torch.eye(2048, device="cuda", dtype=torch.float64)).contiguous()
del a
for dt in (torch.float32, torch.float64):
    c = cov.to(dt)
    for _ in range(2):
        L, info = _hip.cholesky(c, 0.0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    L, info = _hip.cholesky(c, 0.0)
    torch.cuda.synchronize()
    print(f"cholesky {dt}: {1e3 * (time.perf_counter() - t):.2f} ms, info {int(info.abs().max())}; D^3/3 x 10 = {10 * 2048 ** 3 / 3 / 1e9:.1f} GFLOP of multiply-adds x 2")
L64 = L.to(torch.float64)
for _ in range(2):
    w = _hip.tril_inverse(L64)
torch.cuda.synchronize()
t = time.perf_counter()
w = _hip.tril_inverse(L64)
torch.cuda.synchronize()
print(f"tril_inverse f64: {1e3 * (time.perf_counter() - t):.2f} ms")
for _ in range(2):
    p = pinvh_device(cov[0])
torch.cuda.synchronize()
t = time.perf_counter()
p = pinvh_device(cov[0])
torch.cuda.synchronize()
print(f"pinvh 2048 (Cholesky route): {1e3 * (time.perf_counter() - t):.2f} ms, |P A - I| {float((p @ cov[0] - torch.eye(2048, device='cuda', dtype=torch.float64)).abs().max()):.1e}")
