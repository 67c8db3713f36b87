#!/usr/bin/env python3
"""Where the time of a GMM / DDU fit goes at a given width (device gmm_fit + GmmState): class covariances, float32 Cholesky,
triangular inverses, the copies.  Usage: python tools/ablate/run_gmm_fit.py [D] [classes] [rows]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip  # noqa: E402

if os.environ.get("RUNIA_LIB"):  # a library variant (tools/ablate/build_lib_variant.sh)
    _hip._LIB_PATH = os.environ["RUNIA_LIB"]
from runia_core_amd.inference.funcs import GmmState, gmm_fit  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t)
    return r, float(np.median(out))


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    c = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
    g = torch.Generator(device="cuda").manual_seed(1)
    lab = torch.randint(0, c, (n,), device="cuda", generator=g)
    centres = torch.randn(c, d, device="cuda", generator=g) * 0.5
    x = torch.relu(centres[lab] + torch.randn(n, d, device="cuda", generator=g))
    xs = x[torch.argsort(lab)]
    per = n // c
    (_, cov), t_cov = timed(lambda: _hip.covariance(xs[:per]))
    covs = torch.stack([_hip.covariance(xs[i * per:(i + 1) * per])[1].to(torch.float32) for i in range(c)])
    (tril, info), t_chol = timed(lambda: _hip.cholesky(covs, 0.0))
    assert int(info.abs().max()) == 0
    ref = torch.linalg.cholesky(covs.cpu().double())
    err = float((tril.cpu().double() - ref).abs().max() / ref.abs().max())
    w, t_inv = timed(lambda: _hip.tril_inverse(tril.to(torch.float64)))
    eye = float((w.cpu()[0] @ tril.cpu().double()[0] - torch.eye(d, dtype=torch.float64)).abs().max())
    (_, t_d2h) = timed(lambda: tril.cpu())
    xh, labh = x.cpu(), lab.cpu()
    (gj, t_fit) = timed(lambda: gmm_fit(xh, labh, c), reps=1)
    (_, t_state) = timed(lambda: GmmState(gj[0]), reps=1)
    c64, t_chol64 = timed(lambda: _hip.cholesky(covs.to(torch.float64), 0.0), reps=1)
    print(f"D {d} classes {c} rows {n}: covariance of one class {1e3 * t_cov:.2f} ms | cholesky f32 ({c} matrices) {1e3 * t_chol:.1f} ms "
          f"(max err vs f64 LAPACK {err:.2e}) | cholesky f64 {1e3 * t_chol64:.1f} ms | tril_inverse f64 {1e3 * t_inv:.1f} ms (|W L - I| {eye:.1e}) | "
          f"factor D2H {1e3 * t_d2h:.1f} ms | gmm_fit (host rows in) {1e3 * t_fit:.0f} ms | GmmState {1e3 * t_state:.0f} ms")


if __name__ == "__main__":
    main()
