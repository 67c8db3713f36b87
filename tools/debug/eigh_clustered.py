"""Termination of the blocked Jacobi solver on clustered spectra (identity-like precision matrices of whitened data)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
rng = np.random.default_rng(0)
for n, kind in ((256, "whitened"), (256, "identity+1e-16"), (512, "whitened"), (64, "two clusters"), (300, "indefinite"), (256, "rank-deficient"), (100, "zeros")):
    if kind == "whitened":
        x = rng.standard_normal((4096, n)); x -= x.mean(0)
        c = x.T @ x / 4096; w, v = np.linalg.eigh(c); xw = x @ v / np.sqrt(w)
        a = np.linalg.pinv(np.cov(xw.T, bias=True))
    elif kind == "identity+1e-16":
        e = rng.standard_normal((n, n)) * 1e-16; a = np.eye(n) + e + e.T
    elif kind == "two clusters":
        q, _ = np.linalg.qr(rng.standard_normal((n, n))); a = (q * np.r_[np.ones(n // 2), 3 * np.ones(n - n // 2)]) @ q.T
    elif kind == "indefinite":
        g = rng.standard_normal((n, n)); a = g + g.T; np.fill_diagonal(a, 0.0)
    elif kind == "rank-deficient":
        g = rng.standard_normal((n, 10)); a = g @ g.T
    else:
        a = np.zeros((n, n))
    a = (a + a.T) * 0.5
    try:
        w, v = _hip.eigh(torch.from_numpy(a).cuda(), max_sweeps=30)
        w, v = w.cpu().numpy(), v.cpu().numpy()
        nrm = max(1e-300, np.abs(a).max())
        print(f"{kind:16s} n={n}: ok  |w-lapack|/|A| {np.abs(w - np.linalg.eigvalsh(a)).max()/nrm:.1e}  |Av-vw|/|A| {np.abs(a@v - v*w).max()/nrm:.1e}  |V'V-I| {np.abs(v.T@v-np.eye(n)).max():.1e}", flush=True)
    except Exception as e:
        print(f"{kind:16s} n={n}: FAILED {e}", flush=True)
