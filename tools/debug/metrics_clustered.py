"""Clustered score sets through the metrics step (debug aid): device vs oracle, for the library named by RUNIA_LIB."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from runia_core_amd import _hip
if os.environ.get("RUNIA_LIB"): _hip._LIB_PATH = os.environ["RUNIA_LIB"]
for seed in range(6):
    rng = np.random.default_rng(seed)
    nc = 3000
    si = np.concatenate([3.0 + 1e-12 * rng.random(nc), rng.standard_normal(5) * 50])
    so = np.concatenate([3.0 + 1e-12 * (rng.random(nc // 2) - 0.3), rng.standard_normal(3) * 50])
    g = _hip.ood_metrics(torch.from_numpy(si).cuda(), torch.from_numpy(so).cuda()).cpu().numpy()
    e = np.array(oracle.auroc_fpr95_aupr(si, so))
    # the same with the sigmoid taken on the HOST and handed over as scores inside [0, 1] (no exp on the device)
    hi, ho = 1 / (1 + np.exp(-si)), 1 / (1 + np.exp(-so))
    g2 = _hip.ood_metrics(torch.from_numpy(hi).cuda(), torch.from_numpy(ho).cuda()).cpu().numpy()
    print(seed, "device sigmoid", np.abs(g - e).max(), " host sigmoid", np.abs(g2 - e).max())
