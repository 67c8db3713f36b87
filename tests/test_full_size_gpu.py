"""BASELINE.json configs at FULL size on the GPU, checked through size-independent properties
(the oracle cannot finish these sizes in seconds; bounded slices are compared against it)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from runia_core_amd import _hip

    _hip.require_gpu()
    return _hip


def test_cfg3_energy_msp_1m_rows(hip):
    """1M x 1000 logits (4 GB): shift invariance, permutation invariance, slice vs oracle."""
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(1_000_000, 1000, device="cuda", generator=g) * 2
    lse, msp = hip.row_lse_msp(x, True, True)
    assert lse.shape == (1_000_000,) and bool(torch.isfinite(lse).all()) and bool(((msp > 0) & (msp <= 1)).all())
    # logsumexp(x + c) = logsumexp(x) + c ; softmax is shift invariant  (c = 8 is exact in f32 here)
    x += 8.0
    lse2, msp2 = hip.row_lse_msp(x, True, True)
    assert float((lse2 - lse - 8.0).abs().max()) < 2e-5
    assert float((msp2 - msp).abs().max()) < 1e-6
    x -= 8.0
    perm = torch.randperm(1000, device="cuda", generator=g)
    lse3, msp3 = hip.row_lse_msp(x[:100_000][:, perm].contiguous(), True, True)
    assert float((lse3 - lse[:100_000]).abs().max()) < 5e-6 and float((msp3 - msp[:100_000]).abs().max()) < 1e-6
    sl = slice(500_000, 500_512)
    xs = x[sl].cpu().numpy()
    assert rel_err(lse[sl].cpu().numpy(), oracle.energy_score(xs)) < 1e-5
    assert rel_err(msp[sl].cpu().numpy(), oracle.msp_score(xs)) < 1e-5


def _cfg3_features(n, d, c, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    centres = torch.randn(c, d, device="cuda", generator=g) * 0.5
    lab = torch.randint(0, c, (n,), device="cuda", generator=g)
    f = torch.relu(centres[lab] + torch.randn(n, d, device="cuda", generator=g))
    return f, lab


def test_cfg3_mahalanobis_1m_rows(hip):
    """1M x 2048 f32 features, 10 classes: row independence, class-permutation invariance, slice vs oracle."""
    n, d, c = 1_000_000, 2048, 10
    f, lab = _cfg3_features(n, d, c, 2024)
    cm, prec = oracle.mahalanobis_setup(f[:4000].cpu().numpy(), lab[:4000].cpu().numpy(), c)
    packed = hip.pack_weights(torch.from_numpy(prec).cuda())
    mu_p = torch.from_numpy(cm.astype(np.float64) @ prec).cuda()
    s = hip.mahalanobis_score(f, torch.from_numpy(cm).cuda(), packed, mu_p)
    assert s.shape == (n,) and bool(torch.isfinite(s).all()) and bool((s <= 0).all())
    sl = slice(123_456, 123_456 + 3000)
    assert torch.equal(hip.mahalanobis_score(f[sl].contiguous(), torch.from_numpy(cm).cuda(), packed, mu_p), s[sl])
    p = np.random.default_rng(0).permutation(c)
    s_perm = hip.mahalanobis_score(f[sl].contiguous(), torch.from_numpy(cm[p]).cuda(), packed,
                                   torch.from_numpy(cm[p].astype(np.float64) @ prec).cuda())
    assert torch.equal(s_perm, s[sl])  # max over classes does not depend on their order
    fs = f[sl][:64].cpu().numpy()
    assert rel_err(s[sl][:64].cpu().numpy(), oracle.mahalanobis_score(fs, cm, prec, c)) < 1e-9


def test_cfg3_knn_bank_50000x2048(hip):
    """bank 50 000 x 2048, k = 50, 65 536 queries: known answers + row independence + slice vs oracle."""
    m, d, k, nq = 50_000, 2048, 50, 65_536
    g = torch.Generator(device="cuda").manual_seed(7)
    bank = hip.l2_normalize(torch.randn(m, d, device="cuda", generator=g))
    q = torch.randn(nq, d, device="cuda", generator=g)
    q[:m // 2] = bank[: m // 2] * 3.0  # scaled copies of bank rows: nearest neighbour at distance 0 after normalisation
    qn = hip.l2_normalize(q)
    s = hip.knn_kth(qn, bank, k)
    assert s.shape == (nq,) and s.dtype == torch.float32 and bool((s <= 0).all()) and bool((s >= -4.0001).all())
    s1 = hip.knn_kth(qn[:4096].contiguous(), bank, 1)
    assert float(s1.abs().max()) < 2e-6  # k = 1 finds the copied row
    assert torch.equal(hip.knn_kth(qn[30_000:31_000].contiguous(), bank, k), s[30_000:31_000])
    # k-th distance is monotone in k
    s60 = hip.knn_kth(qn[:2048].contiguous(), bank, 60)
    assert bool((s60 <= s[:2048]).all())
    idx = [0, 1, 40_000, 65_535]
    exp = oracle.knn_kth_score(bank.cpu().numpy(), q[idx].cpu().numpy(), k, chunk=1)
    assert rel_err(s[idx].cpu().numpy(), exp) < 1e-5


def test_cfg4_per_proposal_entropy_and_larem(hip):
    """cfg4 shape: 100 proposals x 16 MC x 1024-d per image (here 300 images = 30 000 proposals) -> PCA-64 -> LaREM."""
    n_prop, n_mc, d, n = 30_000, 16, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(4)
    base = torch.randn(n_prop, 1, d, device="cuda", generator=g) + 2
    z = (base * (1 + 0.1 * torch.randn(n_prop, n_mc, d, device="cuda", generator=g))).reshape(n_prop * n_mc, d).contiguous()
    h = hip.kl_entropy_per_dim(z, n_mc, 5)
    assert h.shape == (n_prop, d) and bool(torch.isfinite(h).all())
    rng = np.random.default_rng(1)
    comp = np.linalg.qr(rng.standard_normal((d, n)))[0].T
    mean, var = rng.standard_normal(d), rng.random(n) + 0.1
    a = rng.standard_normal((n, n))
    prec = a @ a.T / n + np.eye(n)
    md_mean = rng.standard_normal((1, n)) * 0.1
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()  # noqa: E731
    s = hip.pca_md_score(h, hip.pack_weights(dev(comp.T)), dev((mean.reshape(1, -1) @ comp.T).ravel()), dev(np.sqrt(var)),
                         dev(md_mean.ravel()), hip.pack_weights(dev(prec)), n)
    zs = z[: 48 * n_mc].cpu().numpy()
    exp, h_exp = oracle.larem_pipeline(zs, n_mc, comp, mean, var, md_mean, prec)
    assert np.abs(h[:48].cpu().numpy() - h_exp).max() < 1e-11
    assert rel_err(s[:48].cpu().numpy(), exp) < 1e-9
    # proposals are independent rows
    s2 = hip.pca_md_score(h[10_000:10_700].contiguous(), hip.pack_weights(dev(comp.T)), dev((mean.reshape(1, -1) @ comp.T).ravel()),
                          dev(np.sqrt(var)), dev(md_mean.ravel()), hip.pack_weights(dev(prec)), n)
    assert torch.equal(s2, s[10_000:10_700])
