#!/usr/bin/env python3
"""Profiling target: 20 x ood_metrics on 1 M + 1 M LaREM-like f64 scores (-chi2(256)), then 20 x on 10 000 + 10 000.
  rocprofv3 --kernel-trace --stats -- python3 tools/debug/metrics_profile_target.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
a = -(torch.randn(n, 256, device="cuda", generator=g, dtype=torch.float64) ** 2).sum(1)
b = -(torch.randn(n, 256, device="cuda", generator=g, dtype=torch.float64) ** 2).sum(1) * 1.15
for _ in range(25):
    out = _hip.ood_metrics(a, b)
torch.cuda.synchronize()
print(out.cpu().tolist())
