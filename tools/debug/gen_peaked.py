"""GEN on confident rows (a winner at p -> 1): the kernel and the float32 oracle against the float64 value.
(1 - p) ** gamma in float32 moves by gamma * ulp(p) / (1 - p) per ulp of p: both carry that error; which is closer is luck."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
import oracle
rng = np.random.default_rng(5)
for c, m, gam, spread in ((16, 1, 0.1, 6.0), (16, 3, 0.5, 6.0), (1000, 3, 0.1, 6.0), (16, 3, 0.05, 6.0), (100, 10, 0.1, 3.0)):
    lg = (rng.standard_normal((4096, c)) * spread).astype(np.float32)
    g = _hip.gen_score(torch.from_numpy(lg).cuda(), gam, m).cpu().numpy().astype(np.float64)
    o32 = oracle.gen_score(lg, gam, m).astype(np.float64)
    o64 = oracle.gen_score(lg.astype(np.float64), gam, m)
    eg, eo = np.abs(g - o64), np.abs(o32 - o64)
    print(f"C {c:4d} M {m:2d} gamma {gam:4.2f} spread {spread}: kernel vs f64 max {eg.max():.2e} mean {eg.mean():.2e} | f32 oracle vs f64 max {eo.max():.2e} "
          f"mean {eo.mean():.2e} | rows where the kernel is further than the oracle: {(eg > eo).mean():.3f}, further than 2x (and > 1e-5): {((eg > 2 * eo) & (eg > 1e-5)).mean():.4f}")
