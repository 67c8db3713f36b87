"""Randomised differential check of the fused sampler + entropy (K0 + K1) and of the table-path sampler against the
CPU oracle over map shapes, sample counts, block sizes, drop probabilities and input scales (one MI355X)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import oracle
from runia_core_amd import _hip
rng = np.random.default_rng(123)
shapes = [(4, 4, range(6, 33)), (2, 2, range(9, 33)), (8, 8, range(9, 33)), (7, 7, range(9, 33))]
bad = 0
for t in range(60):
    h, w, nmcs = shapes[rng.integers(len(shapes))]
    n_mc = int(rng.choice(list(nmcs)))
    c = int(rng.integers(1, 300)); n = int(rng.integers(1, 9)); bs = int(rng.integers(1, min(h, w) + 1))
    p = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.7]))
    x = np.maximum(rng.standard_normal((n, c, h, w)), 0).astype(np.float32) * float(rng.choice([1e-3, 1.0, 50.0]))
    rand = rng.random((n, n_mc, h, w)).astype(np.float32)
    xd, rd = torch.from_numpy(x).cuda(), (torch.from_numpy(rand).cuda() if p > 0 else None)
    hf = _hip.mc_entropy(xd, rd, n_mc, p, bs, 5).cpu().numpy()
    z = np.concatenate([oracle.mc_stack(x[i:i + 1], rand[i], p, bs) for i in range(n)])
    zs = _hip.mc_stack(xd, rd, n_mc, p, bs).cpu().numpy()
    # maps of 2x2 / 4x4: the kernel's summation order is the oracle's (torch's) -> same bits; wider rows are summed
    # by torch in vector lanes, the kernel goes left to right -> a few ulp
    exact = (h, w) in ((4, 4), (2, 2))
    ok_z = np.array_equal(z, zs, equal_nan=True) if exact else bool(np.allclose(z, zs, rtol=3e-6, atol=1e-30, equal_nan=True))
    fin = np.isfinite(zs.reshape(n, n_mc, c)).all(axis=1)
    exp = oracle.kl_entropy_per_dim_vectorized(np.where(np.isfinite(zs), zs, 0.0), n_mc, 5)  # from the kernel's samples
    err = float(np.abs(hf[fin] - exp[fin]).max()) if fin.any() else 0.0
    nan_ok = bool(np.isnan(hf[~fin]).all())
    if not (ok_z and err < 1e-10 and nan_ok):
        bad += 1
        print("MISMATCH", (h, w, n_mc, c, n, bs, p), ok_z, err, nan_ok)
print("fuzz done, mismatches:", bad)

# ---- round 3: counter draws with the redraw of fully dropped maps (K0) against the oracle's equivalent explicit draws ----
for t in range(40):
    h, w, nmcs = shapes[rng.integers(len(shapes))]
    n_mc = int(rng.choice(list(nmcs)))
    c = int(rng.integers(1, 80)); n = int(rng.integers(1, 40)); bs = int(rng.integers(1, min(h, w) + 1))
    p = float(rng.choice([0.3, 0.6, 0.9, 0.99]))
    seed, first = int(rng.integers(0, 2**40)), int(rng.integers(0, 2**33))
    x = torch.relu(torch.randn(n, c, h, w)).cuda()
    red = oracle.counter_draws_redrawn(n, n_mc, h, w, seed, first, p, bs)
    a = _hip.mc_entropy(x, torch.from_numpy(red).cuda(), n_mc, p, bs, 5)
    b = _hip.mc_entropy(x, _hip.CounterDraws(seed, first, True), n_mc, p, bs, 5)
    if not torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)):
        bad += 1
        print("MISMATCH redraw", (h, w, n_mc, c, n, bs, p, seed, first), flush=True)
# ---- round 3: blocked Jacobi eigen-solver against LAPACK (eigenvalues, residual, orthogonality) ----
for t in range(25):
    n = int(rng.integers(1, 300))
    g = rng.standard_normal((n, max(1, int(rng.integers(1, n + 40)))))
    a = g @ g.T / g.shape[1] + np.diag(rng.random(n) * float(rng.choice([0.0, 1e-6, 1.0])))
    w, v = _hip.eigh(torch.from_numpy(a).cuda())
    w, v = w.cpu().numpy(), v.cpu().numpy()
    nrm = max(1.0, float(np.abs(a).max()))
    e1 = float(np.abs(w - np.linalg.eigvalsh(a)).max()) / nrm
    e2 = float(np.abs(a @ v - v * w).max()) / nrm
    e3 = float(np.abs(v.T @ v - np.eye(n)).max())
    if not (e1 < 1e-11 and e2 < 1e-11 and e3 < 1e-11):
        bad += 1
        print("MISMATCH eigh", n, e1, e2, e3, flush=True)
print("fuzz_k1 round-3 families done, mismatches so far:", bad, flush=True)
