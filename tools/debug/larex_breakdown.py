#!/usr/bin/env python3
"""Where the wall time of the harness-shaped workload goes: cProfile of one device-resident log_evaluate_larex sweep with
blocking launches (HIP_LAUNCH_BLOCKING=1 python tools/debug/larex_breakdown.py), cumulative times of the harness-level calls."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_workloads as bw  # noqa: E402
from runia_core_amd.evaluation import log_evaluate_larex  # noqa: E402

device = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else [50000, 10000, 10000]
tr, tr_lab = bw.larex_entropies(device, sizes[0], 100, "ind")
va, va_lab = bw.larex_entropies(device, sizes[1], 200, "ind")
ind = {"train latent_space_means": tr, "valid latent_space_means": va, "train labels": tr_lab, "valid labels": va_lab}
ood = {}
names = ["ood_corr", "ood_shift"]
for i, name in enumerate(names):
    ood[f"{name} latent_space_means"], ood[f"{name} labels"] = bw.larex_entropies(device, sizes[2], 300 + 100 * i, name)


class Cfg:
    ind_dataset, ood_datasets, n_pca_components, num_classes, k_neighbors = "synth", names, list(bw.LAREX_SWEEP), 10, 50


def run(dev):
    np.random.seed(2024)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = log_evaluate_larex(Cfg(), [], {}, dict(ind), dict(ood), postprocessors=list(bw.LAREX_POSTPROCESSORS), device_resident=dev)
    torch.cuda.synchronize()
    return out, time.perf_counter() - t


run(True)
for dev in (True, False):
    pr = cProfile.Profile()
    pr.enable()
    _, sec = run(dev)
    pr.disable()
    print(f"==== device_resident={dev}: {sec:.3f} s")
    st = pstats.Stats(pr)
    st.sort_stats("cumulative")
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        if any(k in fn for k in ("runia_core_amd", "sklearn/decomposition")) or name in ("cholesky_ex", "to_host", "to_device"):
            rows.append((ct, nc, tt, os.path.basename(fn), line, name))
    for ct, nc, tt, fn, line, name in sorted(rows, reverse=True)[:45]:
        print(f"{ct:8.3f} s cum {tt:8.3f} s own {nc:6d} calls  {fn}:{line} {name}")
