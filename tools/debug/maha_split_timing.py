#!/usr/bin/env python3
"""Mahalanobis at cfg3 size (1 M x 2048 f32, 10 classes): the block-major column-split launches (round 5) against the
one-workgroup-per-tile launch - ms, TFLOP/s, same bits.  python tools/debug/maha_split_timing.py [rows]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from runia_core_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, c = 2048, 10
g = torch.Generator(device="cuda").manual_seed(1)
f = torch.relu(torch.randn(n, d, device="cuda", generator=g) + 0.3)
a = torch.randn(d, d, dtype=torch.float64, device="cuda", generator=g)
prec = a @ a.T / d + torch.eye(d, dtype=torch.float64, device="cuda")
cm = torch.randn(c, d, device="cuda", generator=g)
packed = _hip.pack_weights(prec)
mu_p = cm.double() @ prec
flop = n * (2.0 * d * d + 2.0 * d * c + 3.0 * d)
res = {}
for split in (True, False, True, False):
    for _ in range(2):
        s = _hip.mahalanobis_score(f, cm, packed, mu_p, split=split)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s = _hip.mahalanobis_score(f, cm, packed, mu_p, split=split)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"split={split}: {ms:.2f} ms  {flop / ms / 1e9:.1f} TFLOP/s  ({flop / ms / 1e9 / 78.6:.3f} of 78.6)", flush=True)
    res[split] = s
print("same bits:", bool(torch.equal(res[True], res[False])))
