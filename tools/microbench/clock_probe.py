#!/usr/bin/env python3
"""Calibration of runia_clock_probe (csrc/lib.hip): what do the two counters and the dependent-FMA chain read when the GPU
is idle, right after a vector-issue-bound load (K1), right after a matrix-core load (f64 GEMM), and for several chain lengths?
  python tools/microbench/clock_probe.py > gpurun_out/r5_clock_probe.txt
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip  # noqa: E402

_hip.require_gpu()
props = torch.cuda.get_device_properties(0)
print("device", props.name, "clock_rate attr (kHz):", getattr(props, "clock_rate", None))


def probes(n, chain=8192, gap=0.0):
    out = []
    for _ in range(n):
        p = _hip.clock_probe(chain)
        torch.cuda.synchronize()
        out.append(_hip.clock_ghz(p))
        if gap:
            time.sleep(gap)
    return out


def show(tag, recs):
    print(tag)
    for r in recs:
        print("   ghz %.4f  probe %.2f us  cycles/fma %.3f  ns/fma %.4f" % (r["ghz"], r["probe_us"], r["cycles_per_dependent_fma"],
                                                                            r["cycles_per_dependent_fma"] / r["ghz"]))


time.sleep(1.0)
show("idle for 1 s, then 8 probes back to back", probes(8))
for chain in (1024, 4096, 16384, 65536, 262144, 1 << 20):
    show(f"chain {chain}", probes(2, chain))

# vector-issue-bound load: the fused sampler + entropy kernel on 10 000 images
x = torch.relu(torch.randn(10000, 512, 4, 4, device="cuda"))
rand = torch.rand(10000, 16, 4, 4, device="cuda")
for secs in (0.05, 0.3, 1.0, 3.0):
    t = time.perf_counter()
    while time.perf_counter() - t < secs:
        for _ in range(20):
            _hip.mc_entropy(x, rand, 16, 0.5, 2, 5)
        torch.cuda.synchronize()
    show(f"after {secs} s of mc_entropy launches: 4 probes back to back", probes(4))
# probe queued directly behind the load (no host synchronisation between)
for _ in range(200):
    _hip.mc_entropy(x, rand, 16, 0.5, 2, 5)
p = [_hip.clock_probe() for _ in range(3)]
torch.cuda.synchronize()
show("3 probes queued directly behind 200 launches (no sync between)", [_hip.clock_ghz(q) for q in p])

# matrix-core load: f64 products
a = torch.randn(8192, 2048, device="cuda", dtype=torch.float64)
b = torch.randn(2048, 2048, device="cuda", dtype=torch.float64)
t = time.perf_counter()
while time.perf_counter() - t < 1.0:
    for _ in range(5):
        _hip.matmul_f64(a, b)
    torch.cuda.synchronize()
p = [_hip.clock_probe() for _ in range(3)]
torch.cuda.synchronize()
show("after 1 s of f64 matrix products", [_hip.clock_ghz(q) for q in p])
time.sleep(0.5)
show("0.5 s idle later", probes(4))
