#!/usr/bin/env python3
"""ood_metrics on score sets shaped like the postprocessors' real outputs (not only N(0,1)): LaREM scores (-chi2(256)), LaRED
log-densities (-300 ... -2000), kNN distances in [-2, 0], energies ~ N(8, 2), probabilities in [0, 1]; 20 000 and 2 M scores.
python tools/debug/metrics_timing.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker)
from runia_core_amd import _hip  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(3)


def chi2(n, k, scale=1.0):
    return -(torch.randn(n, k, device="cuda", generator=g, dtype=torch.float64) ** 2).sum(1) * scale


sets = {
    "normal": lambda n: (torch.randn(n, device="cuda", generator=g, dtype=torch.float64) + 0.6, torch.randn(n, device="cuda", generator=g, dtype=torch.float64) * 1.3 - 0.4),
    "larem_chi2_256": lambda n: (chi2(n, 256), chi2(n, 256, 1.15)),
    "lared_logdens": lambda n: (chi2(n, 64, 4.0) - 300, chi2(n, 64, 5.0) - 320),
    "knn_f32": lambda n: (-(torch.rand(n, device="cuda", generator=g) * 0.8).float(), -(torch.rand(n, device="cuda", generator=g) * 1.0 + 0.1).float()),
    "energy_f32": lambda n: ((torch.randn(n, device="cuda", generator=g) * 2 + 9).float(), (torch.randn(n, device="cuda", generator=g) * 2 + 7).float()),
    "msp_f32": lambda n: (torch.rand(n, device="cuda", generator=g).float() ** 0.3, torch.rand(n, device="cuda", generator=g).float() ** 0.6),
}
for n in (10_000, 1_000_000):
    for name, make in sets.items():
        a, b = make(n)
        for _ in range(5):
            out = _hip.ood_metrics(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = _hip.ood_metrics(a, b)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        line = f"{2 * n:8d} scores  {name:16s} {ms:8.4f} ms"
        if n <= 10_000:
            exp = oracle.auroc_fpr95_aupr(a.cpu().numpy(), b.cpu().numpy())
            line += f"   max |d| vs oracle {max(abs(x - y) for x, y in zip(out.cpu().tolist(), exp)):.2e}"
        print(line, flush=True)
