"""Rows of the GEN test with peaked logits where the kernel and the oracle differ (which probabilities are involved)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
import oracle
c, m = 16, 5
rng = np.random.default_rng(c * 31 + m)
lg = (rng.standard_normal((257, c)) * 3).astype(np.float32)
lg[5] = 0.0
lg[7:40] *= np.linspace(2, 40, 33, dtype=np.float32)[:, None]
got = _hip.gen_score(torch.from_numpy(lg).cuda(), 0.1, m).cpu().numpy()
exp = oracle.gen_score(lg, 0.1, m)
bad = np.flatnonzero(np.abs(got - exp) > 1e-5 * np.maximum(1, np.abs(exp)))
for r in bad[:6]:
    x = lg[r].astype(np.float64); p = np.exp(x - x.max()); p /= p.sum()
    print(r, got[r], exp[r], "top probabilities (f64):", np.sort(p)[::-1][:7])
