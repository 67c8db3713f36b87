#!/usr/bin/env python3
"""cProfile of one baseline's setup + scoring at cfg3 width (where does a `calculate_all_baselines` entry spend its time).
Usage: python tools/ablate/run_baseline_setup_profile.py ddu|react|mdist|knn|vim"""
import cProfile
import io
import os
import pstats
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_workloads as bw  # noqa: E402
from runia_core_amd.evaluation.baselines import calculate_all_baselines  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ddu"
    dev = torch.device("cuda:0")
    centres = bw.class_centres(dev)
    w_all, b_all = bw.linear_head(dev)
    w, b = w_all[:10].contiguous(), b_all[:10].contiguous()

    def split(n, which):
        f, _ = bw.feature_rows(0, n, which, dev, centres)
        return f.cpu().numpy(), bw.logits_of(f, w, b).cpu().numpy()

    trf, trl = split(50_000, 1)
    vaf, val = split(10_000, 0)
    ind = {"train features": trf, "train logits": trl, "valid features": vaf, "valid logits": val}
    ood = {"o features": vaf.copy(), "o logits": val.copy()}
    fc = {"weight": w.cpu().numpy(), "bias": b.cpu().numpy()}
    cfg = {"ood_datasets": ["o"], "ash_percentile": 90, "react_percentile": 90, "dice_percentile": 90, "gen_gamma": 0.1, "k_neighbors": 50}
    warnings.simplefilter("ignore")
    calculate_all_baselines([name], dict(ind), dict(ood), fc, cfg, 10, device_resident=True)  # warm-up
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    calculate_all_baselines([name], dict(ind), dict(ood), fc, cfg, 10, device_resident=True)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
