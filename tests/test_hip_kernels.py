"""GPU parity of every C-ABI stage against the CPU oracle and the committed goldens.
All calls go through librunia_hip.so (ctypes); the oracle is only the checker."""
import numpy as np
import pytest
import torch

import oracle
from conftest import generate_test_data, load_npz, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5  # BASELINE.json north_star: |d| <= 1e-5 * max(1, |ref|)


@pytest.fixture(scope="module")
def hip():
    from runia_core_amd import _hip

    _hip.require_gpu()
    return _hip


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype=dtype)


# ---------------- a2 entropy -------------------------------------------------------
@pytest.mark.parametrize(
    "n_mc,d,n_img",
    [(16, 512, 64), (16, 37, 9), (3, 20, 200), (2, 8, 5), (4, 12, 7), (5, 16, 6), (6, 20, 6), (8, 64, 10),
     (10, 24, 5), (32, 40, 6), (33, 12, 3), (64, 16, 3)],
)
def test_entropy_per_dim_vs_oracle(hip, n_mc, d, n_img):
    rng = np.random.default_rng(1000 * n_mc + d)
    z = rng.standard_normal((n_img * n_mc, d)).astype(np.float32)
    z[:n_mc, 0] = 0.75  # constant column -> min_dist clip
    z[0:n_mc:2, 1] = z[1, 1]  # ties
    if n_img > 1:
        z[n_mc : 2 * n_mc, 2] *= 1e-7  # tiny spread, partially clipped
    k = 5 if n_mc > 5 else n_mc - 1
    got = hip.kl_entropy_per_dim(dev(z, torch.float32), n_mc, k).cpu().numpy()
    exp = oracle.kl_entropy_per_dim_vectorized(z, n_mc, k)
    assert got.shape == exp.shape
    assert np.abs(got - exp).max() < 1e-11


def test_entropy_reference_goldens(hip, ref_vectors):
    # /root/reference/tests/unit_test_feature_extraction.py:175-247 and integration_tests.py:216-277
    np.random.seed(1)
    s = np.random.rand(3, 20).astype(np.float32)  # golden is defined on f64 draws; compare via oracle on same f32
    got = hip.kl_entropy_per_dim(dev(s, torch.float32), 3, 2).cpu().numpy()[0]
    assert np.abs(got - oracle.single_image_entropy_calculation(s, 2)).max() < 1e-12
    torch.manual_seed(1)
    z = torch.rand(600, 20)
    got = hip.kl_entropy_per_dim(z.cuda(), 3, 2).cpu().numpy()
    exp = np.array(ref_vectors["entropy_get_dl_h_z"]["lists"][0]["values"])
    assert np.allclose(got[0], exp, atol=1e-6)
    z = torch.full((3, 20), 0.3)
    got = hip.kl_entropy_per_dim(z.cuda(), 3, 2).cpu().numpy()
    assert np.abs(got - (-10.319778284410283)).max() < 1e-12


@pytest.mark.parametrize("n_mc,k", [(16, 3), (16, 1), (7, 6), (12, 11), (40, 9)])
def test_entropy_generic_k(hip, n_mc, k):
    rng = np.random.default_rng(n_mc + k)
    z = rng.standard_normal((4 * n_mc, 10)).astype(np.float32)
    got = hip.kl_entropy_per_dim(dev(z, torch.float32), n_mc, k).cpu().numpy()
    assert np.abs(got - oracle.kl_entropy_per_dim_vectorized(z, n_mc, k)).max() < 1e-11


@pytest.mark.parametrize("n_mc,d,n_img", [(16, 512, 5), (3, 20, 40), (6, 300, 3), (33, 17, 2), (64, 130, 2),
                                          # register form: short sample sets, rows shorter than a wave, several chunks
                                          (12, 512, 4), (16, 4, 9), (16, 2048, 3), (16, 1028, 2), (8, 64, 7), (5, 12, 6),
                                          (32, 130, 3), (20, 1024, 2), (9, 1500, 2), (2, 8, 5), (16, 510, 3)])
def test_entropy_joint_vs_oracle(hip, n_mc, d, n_img):
    rng = np.random.default_rng(n_mc * 7 + d)
    z = rng.standard_normal((n_img * n_mc, d)).astype(np.float32)
    k = 5 if n_mc > 5 else n_mc - 1
    got = hip.kl_entropy_joint(dev(z, torch.float32), n_mc, k).cpu().numpy()
    exp = oracle.kl_entropy_joint_vectorized(z, n_mc, k)[:, 0]
    assert rel_err(got, exp) < 1e-11
    # a view that starts 4 bytes into the buffer takes the LDS form (unaligned rows): same bits
    buf = torch.empty(z.size + 1, dtype=torch.float32, device="cuda")
    buf[1:] = dev(z, torch.float32).reshape(-1)
    off = hip.kl_entropy_joint(buf[1:].view(z.shape), n_mc, k).cpu().numpy()
    assert np.array_equal(off, got)


# ---------------- a7 energy / msp -----------------------------------------------------
@pytest.mark.parametrize("c", [2, 10, 16, 20, 32, 63, 64, 65, 100, 1000, 1001, 1024, 2048, 3000])
def test_lse_msp_vs_oracle(hip, c):
    rng = np.random.default_rng(c)
    x = (rng.standard_normal((777, c)) * 4).astype(np.float32)
    x[3, 0] = 88.0
    x[4, :] = -30.0
    lse, msp = hip.row_lse_msp(dev(x, torch.float32), True, True)
    assert lse.dtype == torch.float32
    assert rel_err(lse.cpu().numpy(), oracle.energy_score(x)) < TOL
    assert rel_err(msp.cpu().numpy(), oracle.msp_score(x)) < TOL
    # f32 agreement is in fact at the few-ulp level
    assert rel_err(lse.cpu().numpy(), oracle.energy_score(x)) < 2e-6


def test_lse_msp_fixtures_and_edge_rows(hip):
    g = load_npz("ref_energy_msp.npz")
    for nm in ("c1000", "c10", "unit_test"):
        x = g[f"{nm}_logits"]
        lse, msp = hip.row_lse_msp(dev(x, torch.float32), True, True)
        exp_l = g[f"{nm}_energy_scores"] if nm != "unit_test" else -g["unit_energy_scores"]
        exp_m = g[f"{nm}_msp_scores"] if nm != "unit_test" else -g["unit_msp_scores"]
        assert rel_err(lse.cpu().numpy(), exp_l) < TOL
        assert rel_err(msp.cpu().numpy(), exp_m) < TOL
    x = np.zeros((4, 10), dtype=np.float32)
    x[0, 3] = np.inf
    x[1, :] = -np.inf
    x[2, 5] = np.nan
    lse, msp = hip.row_lse_msp(dev(x, torch.float32), True, True)
    with np.errstate(all="ignore"):
        el, em = oracle.energy_score(x), oracle.msp_score(x)
    assert np.array_equal(np.isnan(lse.cpu().numpy()), np.isnan(el))
    assert np.array_equal(np.isnan(msp.cpu().numpy()), np.isnan(em))
    ok = ~np.isnan(el)
    assert np.array_equal(lse.cpu().numpy()[ok], el[ok])
    lse0, _ = hip.row_lse_msp(torch.empty((0, 10), device="cuda"), True, False)
    assert lse0.shape == (0,)


# ---------------- a4 PCA ------------------------------------------------------------------
def _pca_state(g, prefix):
    comp, mean, var = g[f"{prefix}_components"], g[f"{prefix}_mean"], g[f"{prefix}_var"]
    bias = (mean.reshape(1, -1) @ comp.T).ravel()
    scale = np.sqrt(var)
    scale = np.where(scale < np.finfo(np.float64).eps, np.finfo(np.float64).eps, scale)
    return comp, bias, scale


def test_pca_transform_goldens(hip, ref_vectors):
    g = load_npz("ref_pca.npz")
    for prefix, xs, exps in (
        ("unit", [g["unit_ood"], g["unit_ind"]], [g["unit_ood_transformed"], g["unit_train_transformed"]]),
        ("d512", [g["d512_test"]], [g["d512_test_transformed"]]),
    ):
        comp, bias, scale = _pca_state(g, prefix)
        packed = hip.pack_weights(dev(comp.T.copy(), torch.float64))
        for x, exp in zip(xs, exps):
            y = hip.pca_transform(dev(x, torch.float64), packed, dev(bias, torch.float64), dev(scale, torch.float64), comp.shape[0])
            assert rel_err(y.cpu().numpy(), exp) < 1e-11
    # f32 input rows (sklearn promotes to f64)
    comp, bias, scale = _pca_state(g, "d512")
    packed = hip.pack_weights(dev(comp.T.copy(), torch.float64))
    y = hip.pca_transform(dev(g["d512_test"].astype(np.float32), torch.float32), packed, dev(bias, torch.float64),
                          dev(scale, torch.float64), comp.shape[0])
    assert rel_err(y.cpu().numpy(), g["d512_test_f32_transformed"]) < 1e-11
    # the reference's own golden row (tests/unit_test_dim_reduction.py:92-103)
    comp, bias, scale = _pca_state(g, "unit")
    packed = hip.pack_weights(dev(comp.T.copy(), torch.float64))
    y = hip.pca_transform(dev(g["unit_ood"], torch.float64), packed, dev(bias, torch.float64), dev(scale, torch.float64), 10)
    exp0 = np.array(ref_vectors["pca_transform"]["lists"][0]["values"])
    assert abs((y.cpu().numpy()[0] - exp0).sum()) < 1e-7


@pytest.mark.parametrize("n_rows,d,n", [(1, 20, 1), (33, 20, 2), (70, 36, 4), (100, 512, 256), (257, 100, 300), (5, 7, 3)])
def test_pca_transform_shapes(hip, n_rows, d, n):
    rng = np.random.default_rng(n_rows + d + n)
    x = rng.standard_normal((n_rows, d))
    comp = rng.standard_normal((n, d))
    mean = rng.standard_normal(d)
    var = rng.random(n) + 0.1
    exp = oracle.pca_transform(x, comp, mean, var)
    bias = (mean.reshape(1, -1) @ comp.T).ravel()
    packed = hip.pack_weights(dev(comp.T.copy(), torch.float64))
    y = hip.pca_transform(dev(x, torch.float64), packed, dev(bias, torch.float64), dev(np.sqrt(var), torch.float64), n)
    assert rel_err(y.cpu().numpy(), exp) < 1e-11
    y = hip.pca_transform(dev(x, torch.float64), packed, dev(bias, torch.float64), None, n)
    assert rel_err(y.cpu().numpy(), oracle.pca_transform(x, comp, mean, var, whiten=False)) < 1e-11


# ---------------- a5 MD ----------------------------------------------------------------------
def test_md_goldens(hip, ref_vectors):
    g = load_npz("ref_md.npz")
    for name in ("unit", "baselines", "d256"):
        x, mean, prec, exp = g[f"{name}_test"], g[f"{name}_mean"], g[f"{name}_precision"], g[f"{name}_scores"]
        packed = hip.pack_weights(dev(prec, torch.float64))
        # dtypes as the reference saw them: the unit case is f32 features with an f32 mean
        tx = torch.float32 if x.dtype == np.float32 else torch.float64
        tm = torch.float32 if mean.dtype == np.float32 else torch.float64
        s = hip.md_score(dev(x, tx), dev(mean.ravel(), tm), packed).cpu().numpy()
        assert rel_err(s, exp) < 1e-10, name
    exp = np.array(ref_vectors["md_unit"]["lists"][0]["values"])
    x, mean, prec = g["unit_test"], g["unit_mean"], g["unit_precision"]
    s = hip.md_score(dev(x, torch.float32), dev(mean.ravel(), torch.float32), hip.pack_weights(dev(prec, torch.float64)))
    assert abs((exp - s.cpu().numpy()).sum()) < 1e-6  # the reference test's own assertion
    assert rel_err(s.cpu().numpy(), exp) < 1e-10
    # mixed dtypes promote to f64
    s = hip.md_score(dev(x, torch.float32), dev(mean.ravel().astype(np.float64), torch.float64),
                     hip.pack_weights(dev(prec, torch.float64))).cpu().numpy()
    assert rel_err(s, oracle.md_score(x, mean.astype(np.float64), prec)) < 1e-11


@pytest.mark.parametrize("n_rows,n", [(1, 1), (3, 2), (31, 4), (32, 16), (33, 256), (500, 300), (64, 600)])
def test_md_shapes(hip, n_rows, n):
    rng = np.random.default_rng(n_rows * 3 + n)
    a = rng.standard_normal((n, n))
    prec = a @ a.T / n + np.eye(n)
    mean = rng.standard_normal((1, n))
    x = rng.standard_normal((n_rows, n)) * 1.5
    s = hip.md_score(dev(x, torch.float64), dev(mean.ravel(), torch.float64), hip.pack_weights(dev(prec, torch.float64)))
    assert rel_err(s.cpu().numpy(), oracle.md_score(x, mean, prec)) < 1e-11


# ---------------- a6 Mahalanobis -----------------------------------------------------------------
def test_mahalanobis_goldens(hip, ref_vectors):
    g = load_npz("ref_mahalanobis.npz")
    for name, dt, suffix in (("unit", np.float32, ""), ("d96", np.float32, ""), ("d96", np.float64, "_f64")):
        x = g[f"{name}_test"].astype(dt)
        cm = g[f"{name}{suffix}_class_mean"].astype(dt)
        prec = g[f"{name}{suffix}_precision"]
        exp = g[f"{name}{suffix}_scores"]
        if name == "unit":
            exp = -exp  # fixture was produced with flip_sign=True
        tdt = torch.float32 if dt == np.float32 else torch.float64
        packed = hip.pack_weights(dev(prec, torch.float64))
        mu_p = dev(cm.astype(np.float64) @ prec, torch.float64)
        s = hip.mahalanobis_score(dev(x, tdt), dev(cm, tdt), packed, mu_p).cpu().numpy()
        assert rel_err(s, exp) < 1e-9, (name, dt)


def test_mahalanobis_empty_class_and_shapes(hip):
    rng = np.random.default_rng(5)
    d, c = 40, 6
    x = rng.standard_normal((77, d)).astype(np.float32)
    cm = rng.standard_normal((c, d)).astype(np.float32)
    cm[2, :] = np.nan  # class without training samples (mean of empty slice)
    a = rng.standard_normal((d, d))
    prec = a @ a.T / d + np.eye(d)
    with np.errstate(all="ignore"):
        exp = oracle.mahalanobis_score(x, cm, prec, c)
        mu_p = cm.astype(np.float64) @ prec
    s = hip.mahalanobis_score(dev(x, torch.float32), dev(cm, torch.float32), hip.pack_weights(dev(prec, torch.float64)),
                              dev(mu_p, torch.float64)).cpu().numpy()
    assert rel_err(s, exp) < 1e-10


@pytest.mark.parametrize("d,c,n,dt", [(40, 17, 77, np.float32), (64, 100, 300, np.float32), (300, 40, 50, np.float64),
                                      (33, 1000, 20, np.float32), (512, 64, 1000, np.float32)])
def test_mahalanobis_many_classes(hip, d, c, n, dt):
    """More than 16 classes: the class terms as a second contraction (S = G M^T ranks the classes, the f32-difference
    formula finishes the candidates) against the oracle and against the per-class loop; an empty class (NaN mean), two
    identical classes (a tie for the maximum) and rows that sit exactly on a class mean included."""
    rng = np.random.default_rng(d + c)
    centres = (rng.standard_normal((c, d)) * 2).astype(dt)
    centres[3] = np.nan
    centres[5] = centres[4]
    lab = rng.integers(0, c, n)
    lab[lab == 3] = 0
    x = (np.nan_to_num(centres[lab]) + rng.standard_normal((n, d)) * 0.7 + 10.0).astype(dt)  # far from the origin: cancellation
    centres = (centres + 10.0).astype(dt)
    x[1] = centres[7]
    a = rng.standard_normal((d, d))
    prec = a @ a.T / d + 0.05 * np.eye(d)
    with np.errstate(all="ignore"):
        exp = oracle.mahalanobis_score(x, centres, prec, c)
        mu_p = centres.astype(np.float64) @ prec
    tdt = torch.float32 if dt == np.float32 else torch.float64
    packed = hip.pack_weights(dev(prec, torch.float64))
    s = hip.mahalanobis_score(dev(x, tdt), dev(centres, tdt), packed, dev(mu_p, torch.float64)).cpu().numpy()
    loop = hip.mahalanobis_score(dev(x, tdt), dev(centres, tdt), packed, dev(mu_p, torch.float64), class_loop=True).cpu().numpy()
    assert np.isfinite(s).all()
    scale = np.maximum(1.0, np.abs(exp))
    assert (np.abs(s - exp) / scale).max() < 1e-9 and (np.abs(loop - exp) / scale).max() < 1e-9
    # every class empty -> -inf, as the reference (NaN -> -inf, then the maximum)
    allnan = np.full((c, d), np.nan, dtype=dt)
    s2 = hip.mahalanobis_score(dev(x[:5], tdt), dev(allnan, tdt), packed, dev(np.full((c, d), np.nan), torch.float64)).cpu().numpy()
    assert np.isneginf(s2).all()


@pytest.mark.parametrize("d,c,cond", [(64, 40, 1e5), (128, 100, 1e7), (96, 17, 1e9)])
def test_mahalanobis_many_classes_ill_conditioned_precision(hip, d, c, cond):
    """More than 16 classes with a precision matrix of condition 1e5 ... 1e9 and class means far closer to each other than
    the 1e-3 window in which candidate classes are re-evaluated with the reference's float32-difference form: the exact
    ranking + re-evaluation must still return the class loop's scores (oracle: reference funcs.py:88-100, vectorised)."""
    rng = np.random.default_rng(int(np.log10(cond)) + d)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    prec = (q * np.logspace(0, np.log10(cond), d)) @ q.T
    prec = (prec + prec.T) * 0.5
    base = rng.standard_normal(d).astype(np.float32) * 3 + 5.0
    # clusters of near-coincident class means: the best classes of a row differ by far less than the score's 1e-3
    centres = (base[None, :] + rng.standard_normal((c, d)).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
    centres[::3] += rng.standard_normal((len(centres[::3]), d)).astype(np.float32)
    n = 400
    lab = rng.integers(0, c, n)
    x = (centres[lab] + rng.standard_normal((n, d)).astype(np.float32) * np.float32(0.05)).astype(np.float32)
    exp = oracle.mahalanobis_score(x, centres, prec, c)
    mu_p = centres.astype(np.float64) @ prec
    packed = hip.pack_weights(dev(prec, torch.float64))
    s = hip.mahalanobis_score(dev(x, torch.float32), dev(centres, torch.float32), packed, dev(mu_p, torch.float64)).cpu().numpy()
    loop = hip.mahalanobis_score(dev(x, torch.float32), dev(centres, torch.float32), packed, dev(mu_p, torch.float64),
                                 class_loop=True).cpu().numpy()
    scale = np.maximum(1.0, np.abs(exp))
    assert (np.abs(loop - exp) / scale).max() < 1e-9
    assert (np.abs(s - exp) / scale).max() < 1e-9, float((np.abs(s - exp) / scale).max())


@pytest.mark.parametrize("c,m", [(1, 1), (2, 1), (7, 3), (10, 10), (10, 100), (16, 5), (16, 16), (17, 5), (64, 64), (100, 10),
                                 (1000, 100), (1000, 600), (1000, 1000), (3000, 100)])
def test_gen_score_widths_and_m(hip, c, m):
    """GEN over its three kernels (row per lane up to 16 classes, wave per row beyond; selected probabilities packed
    through LDS for M <= 512) and every relation of M to the number of classes, against the oracle (f32, 1e-5)."""
    rng = np.random.default_rng(c * 31 + m)
    lg = (rng.standard_normal((257, c)) * 3).astype(np.float32)
    lg[5] = 0.0  # ties: every probability equal
    got = hip.gen_score(dev(lg, torch.float32), 0.1, m).cpu().numpy()
    assert got.dtype == np.float32 and rel_err(got, oracle.gen_score(lg, 0.1, m)) < 1e-5
    for gamma in (1.0, 0.0, 2.5):
        got = hip.gen_score(dev(lg, torch.float32), gamma, m).cpu().numpy()
        assert rel_err(got, oracle.gen_score(lg, gamma, m)) < 1e-5, gamma
    # probabilities in the denormal range and exact zeros still count at gamma = 0.1 (1e-40 ** 0.1 = 1e-4): the kernel's
    # exponentials and logarithms run on the transcendental unit, whose plain instructions flush both - the rescaled forms
    # must not.  (Winners stay away from p = 1: (1 - p) ** gamma at p = 1 - 2.5e-7 moves by 2.5 % per ulp of p, i.e. with
    # the summation order of the softmax denominator, in any implementation.)
    if c >= 8:
        tails = np.stack([np.linspace(0.0, -115.0, c), np.concatenate([np.zeros(3), -rng.uniform(88.0, 98.0, c - 3)]),
                          np.concatenate([np.zeros(2), np.full(c - 2, -300.0)])]).astype(np.float32)
        tails[0, 1] = 0.0
        for gamma in (0.1, 1.0):
            got = hip.gen_score(dev(tails, torch.float32), gamma, m).cpu().numpy()
            assert rel_err(got, oracle.gen_score(tails, gamma, m)) < 1e-5, gamma


@pytest.mark.parametrize("n,d,dt", [(1, 5, np.float64), (9, 3, np.float32), (300, 70, np.float64), (1000, 127, np.float32),
                                     (2500, 129, np.float64), (700, 257, np.float32), (5000, 384, np.float32),
                                     (3001, 2048, np.float32), (513, 1030, np.float64)])
def test_covariance_tiles_of_the_upper_triangle_mirrored(hip, n, d, dt):
    """np.cov(x.T, bias=1) from 128 x 128 tile pairs of the upper triangle + the mirroring finish: widths around the tile
    and vector-load edges, row counts around the split and staging steps, a base pointer off the 16-byte grid, and exact
    symmetry of the result."""
    rng = np.random.default_rng(n + d)
    x = (rng.standard_normal((n, d)) * (0.5 + rng.random(d)) + 3 * rng.standard_normal(d)).astype(dt)
    exp = np.cov(x.astype(np.float64).T, bias=1).reshape(d, d)
    for off in (0, 1):
        buf = torch.empty(n * d + 1, dtype=torch.float32 if dt == np.float32 else torch.float64, device="cuda")
        xd = buf[off: off + n * d].view(n, d)
        xd.copy_(torch.from_numpy(x))
        mean, cov = hip.covariance(xd)
        c = cov.cpu().numpy()
        assert rel_err(mean.cpu().numpy(), x.astype(np.float64).mean(0)) < 1e-13
        assert np.array_equal(c, c.T) and rel_err(c, exp) < 1e-12, (n, d, off)


@pytest.mark.parametrize("d", [3, 64, 65, 300, 512, 513, 1000, 1024, 2048, 2049, 4096])
def test_ash_s_selection_widths_ties_and_distributions(hip, d):
    """The per-row k-th-largest search (bit by bit, the operand set packed down through LDS as the range narrows) over every
    register count of the kernel, with rows that stress the packing: all values in one binade, heavy ties at the
    threshold, all equal, mixed signs, a few huge outliers, denormals and zeros - kept SET and scale against the oracle."""
    rng = np.random.default_rng(d)
    rows = [np.maximum(rng.standard_normal(d), 0), rng.standard_normal(d), 1.0 + rng.random(d),        # relu, signed, one binade
            np.round(rng.random(d) * 4) / 4 + 0.25, np.full(d, 0.75), rng.integers(0, 3, d).astype(np.float64) + 0.5,
            np.exp(rng.standard_normal(d) * 8), rng.random(d) * 1e-41, -rng.random(d) - 0.5,
            np.where(rng.random(d) < 0.02, 1e30, rng.random(d))]
    x = np.stack(rows).astype(np.float32)
    for pct in (0, 10, 50, 65, 90, 99, 100):
        with np.errstate(all="ignore"):
            exp = oracle.ash_s_defined(x.copy(), pct)
        got = hip.ash_s(dev(x, torch.float32), pct).cpu().numpy()
        ok = np.isfinite(exp)
        assert np.array_equal(np.isfinite(got), ok), (d, pct)
        assert np.array_equal((got != 0) & ok, (exp != 0) & ok), (d, pct)  # the same entries are kept
        assert rel_err(got[ok], exp[ok]) < 1e-4, (d, pct)  # (the factor exp(sum / kept sum) amplifies f32 summation order at 99 %)
    # GEN's top-M over the same rows taken as probabilities (normalised, non-negative)
    p = np.abs(x[[0, 2, 3, 4, 5, 6]]).astype(np.float64) + 1e-12
    p = (p / p.sum(axis=1, keepdims=True)).astype(np.float32)
    for m in (1, 7, max(1, d // 10), max(1, d // 2), d):
        got = hip.gen_entropy(dev(p, torch.float32), 0.1, m).cpu().numpy()
        assert rel_err(got, oracle.generalized_entropy(p, 0.1, m)) < 1e-5, (d, m)


@pytest.mark.parametrize("n,d,c", [(1, 4, 1), (65, 512, 10), (1000, 516, 3), (333, 1536, 16), (100, 2048, 12), (70, 512, 17),
                                   (257, 510, 10), (64, 2048, 16), (129, 100, 16)])
def test_linear_head_small_and_wide(hip, n, d, c):
    """Final linear layer (ReAct / DICE / ASH / ViM logits): the row-streaming kernel for heads of up to 16 classes
    (weights in LDS, wave-level halving sums) and the matrix-core kernel otherwise, with and without clip and bias."""
    rng = np.random.default_rng(n + d + c)
    x = np.maximum(rng.standard_normal((n, d)), 0).astype(np.float32) * 2
    w = (rng.standard_normal((c, d)) / np.sqrt(d)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    for clip in (float("inf"), 0.9):
        xc = np.minimum(x, np.float32(clip)).astype(np.float64)
        exp = xc @ w.astype(np.float64).T
        got = hip.linear(dev(x, torch.float32), dev(w, torch.float32), dev(b, torch.float32), clip).cpu().numpy()
        scale = np.abs(xc) @ np.abs(w.astype(np.float64)).T + 1.0  # f32 accumulation error grows with sum |x||w|
        assert got.shape == (n, c) and (np.abs(got - (exp + b)) / scale).max() < 3e-6
        nb = hip.linear(dev(x, torch.float32), dev(w, torch.float32), None, clip).cpu().numpy()
        assert (np.abs(nb - exp) / scale).max() < 3e-6


# ---------------- a8 kNN ----------------------------------------------------------------------------
@pytest.mark.parametrize("m,d,k,n", [(200, 20, 10, 200), (1000, 64, 50, 33), (300, 2048, 50, 17), (70, 33, 1, 9), (64, 16, 64, 5)])
def test_knn_vs_oracle(hip, m, d, k, n):
    rng = np.random.default_rng(m + d + k)
    bank = rng.standard_normal((m, d)).astype(np.float32)
    q = rng.standard_normal((n, d)).astype(np.float32)
    q[0] = bank[3]  # exact hit -> distance 0 in the list
    bank_n = hip.l2_normalize(dev(bank, torch.float32))
    q_n = hip.l2_normalize(dev(q, torch.float32))
    assert rel_err(bank_n.cpu().numpy(), oracle.normalizer(bank)) < 1e-6
    got = hip.knn_kth(q_n, bank_n, k).cpu().numpy()
    exp = oracle.knn_kth_score(np.ascontiguousarray(oracle.normalizer(bank).astype(np.float32)), q, k)
    assert got.dtype == np.float32
    assert rel_err(got, exp) < TOL


def test_knn_k_larger_than_bank(hip):
    tr, _, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    got = hip.knn_kth(hip.l2_normalize(dev(te, torch.float32)), hip.l2_normalize(dev(tr, torch.float32)), 50).cpu().numpy()
    assert np.array_equal(got, np.full(10, -oracle.FLT_MAX, dtype=np.float32))


def test_knn_adversarial_inputs_through_the_raw_entry(hip):
    """runia_knn_kth_f32 called directly (no normaliser in front) on data built to break a histogram select:
    1 000 copies of one bank row with k on and around the tie boundary, un-normalised rows spanning six orders of
    magnitude, an outlier norm that crowds every other distance into a few key bins, NaN / infinite query and bank rows.
    Every row must be written on every path (the output buffer is pre-filled with a sentinel) and equal the oracle."""
    rng = np.random.default_rng(11)
    d = 96
    # (1) duplicated bank rows: the refinement window holds > 512 candidates -> exact slow path
    base = rng.standard_normal((300, d)).astype(np.float32)
    dup = np.repeat(base[:1], 1000, axis=0)
    bank = np.concatenate([dup, base[1:]]).astype(np.float32)
    q = np.concatenate([base[:1] + 0.01 * rng.standard_normal((6, d)).astype(np.float32),
                        rng.standard_normal((10, d)).astype(np.float32)])
    for k in (1, 2, 500, 999, 1000, 1001, 1100):
        sentinel = torch.full((q.shape[0],), 123.0, device="cuda")
        ws_bytes = hip.load_library().runia_knn_workspace_bytes(q.shape[0], bank.shape[0], d, k)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device="cuda")
        qd, bd = dev(q, torch.float32), dev(bank, torch.float32)
        rc = hip.load_library().runia_knn_kth_f32(qd.data_ptr(), bd.data_ptr(), sentinel.data_ptr(), ws.data_ptr(), ws_bytes,
                                                  q.shape[0], bank.shape[0], d, k, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        got = sentinel.cpu().numpy()
        assert not (got == 123.0).any(), f"k={k}: a row was left unwritten"
        assert rel_err(got, oracle.knn_kth_score(bank, q, k, normalize=False)) < 1e-5, k
    # (2) un-normalised, wide-range rows
    scale = 10.0 ** rng.uniform(-3, 3, size=(2000, 1))
    bank2 = (rng.standard_normal((2000, d)) * scale).astype(np.float32)
    q2 = (rng.standard_normal((64, d)) * 10.0 ** rng.uniform(-3, 3, size=(64, 1))).astype(np.float32)
    for k in (1, 50, 1999):
        got = hip.knn_kth(dev(q2, torch.float32), dev(bank2, torch.float32), k).cpu().numpy()
        assert rel_err(got, oracle.knn_kth_score(bank2, q2, k, normalize=False)) < 1e-5, k
    # (3) one outlier norm: the range bound is set by it, all other distances share a handful of key bins
    bank3 = rng.standard_normal((3000, d)).astype(np.float32) * 1e-2
    bank3[7] = 1e4
    q3 = rng.standard_normal((32, d)).astype(np.float32) * 1e-2
    for k in (1, 50, 2999, 3000):
        got = hip.knn_kth(dev(q3, torch.float32), dev(bank3, torch.float32), k).cpu().numpy()
        assert rel_err(got, oracle.knn_kth_score(bank3, q3, k, normalize=False)) < 1e-5, k
    # (4) NaN / infinite rows: faiss never inserts an incomparable distance -> they count as its FLT_MAX fill
    bank4 = rng.standard_normal((200, d)).astype(np.float32)
    bank4[3, 5] = np.nan
    bank4[9, 0] = np.inf
    q4 = rng.standard_normal((8, d)).astype(np.float32)
    q4[1, 2] = np.nan
    q4[5, :] = np.inf
    for k in (1, 50, 198, 199, 200):
        got = hip.knn_kth(dev(q4, torch.float32), dev(bank4, torch.float32), k).cpu().numpy()
        exp = oracle.knn_kth_score(bank4, q4, k, normalize=False)
        assert got[1] == -oracle.FLT_MAX and got[5] == -oracle.FLT_MAX
        assert np.isfinite(got).all() and rel_err(got, exp) < 1e-5, k
    # through the postprocessor: a NaN feature row scores -FLT_MAX, the others are untouched
    from runia_core_amd.inference import KNNLatentSpace

    knn = KNNLatentSpace()
    knn.setup(rng.standard_normal((500, 32)).astype(np.float32))
    xt = rng.standard_normal((20, 32)).astype(np.float32)
    clean = knn.postprocess(xt)
    xt2 = xt.copy()
    xt2[4, 0] = np.nan
    dirty = knn.postprocess(xt2)
    assert dirty[4] == -oracle.FLT_MAX and np.array_equal(np.delete(dirty, 4), np.delete(clean, 4))


@pytest.mark.parametrize("m", [9001, 12000, 20002])
def test_knn_sampled_threshold_path_and_its_fallbacks(hip, m):
    """Banks large enough for the one-read select (M >= 8192): 16-byte and scalar row loads (M % 4), k from 1 to the
    path's limit and beyond it (k > 512 -> three-read path), a bank with 3 000 copies of one row (more candidates below
    the sampled threshold than the LDS list holds -> three-read path, then the crowded-window slow path), rows whose
    sample misrepresents the row (bank sorted by distance to the query)."""
    rng = np.random.default_rng(m)
    d = 48
    bank = rng.standard_normal((m, d)).astype(np.float32)
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    q = rng.standard_normal((12, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    for k in (1, 5, 50, 512, 513, 2000):
        got = hip.knn_kth(dev(q, torch.float32), dev(bank, torch.float32), k).cpu().numpy()
        assert rel_err(got, oracle.knn_kth_score(bank, q, k, normalize=False)) < 1e-5, (m, k)
    dup = bank.copy()
    dup[1000:4000] = dup[17]
    for k in (1, 50, 2999, 3001):
        got = hip.knn_kth(dev(q, torch.float32), dev(dup, torch.float32), k).cpu().numpy()
        assert rel_err(got, oracle.knn_kth_score(dup, q, k, normalize=False)) < 1e-5, (m, k)
    # bank ordered by distance to query 0: the strided sample still spans the row, the first rows do not
    order = np.argsort(((bank - q[0]) ** 2).sum(1))
    sorted_bank = np.ascontiguousarray(bank[order])
    for k in (1, 50):
        got = hip.knn_kth(dev(q, torch.float32), dev(sorted_bank, torch.float32), k).cpu().numpy()
        assert rel_err(got, oracle.knn_kth_score(sorted_bank, q, k, normalize=False)) < 1e-5, (m, k)


@pytest.mark.parametrize("n,m,d", [(2048, 8192, 512), (1500, 8200, 300), (1024, 4096, 2048), (8192, 8300, 64), (4100, 8192, 100),
                                   (8192, 4096, 128), (16400, 8200, 16), (32768, 8192, 9), (16400, 8200, 4), (32768, 8192, 2),
                                   (20000, 8192, 3)])  # (round 6: from two features)
def test_knn_bf16_candidate_distances_equal_the_f32_path(hip, n, m, d):
    """Large problems take their candidate distances from bf16 piece products (csrc/knn_bf16.hip) when the caller's
    workspace holds the planes (runia_knn_workspace_bytes asks for them); with an f32-sized workspace the same entry
    takes the f32 matrix-core kernel.  Both feed the same selection + exact f32 re-measurement, so the scores agree bit
    for bit - on unit vectors, on rows spanning six orders of magnitude, with copied bank rows, NaN / infinite rows and
    ragged sizes (rows not a multiple of 256, width not a multiple of 64) - and both equal the oracle."""
    lib = hip.load_library()
    rng = np.random.default_rng(n + m + d)
    k = 50

    def run(q, bank, big):
        qd, bd = dev(q, torch.float32), dev(bank, torch.float32)
        full = lib.runia_knn_workspace_bytes(n, m, d, k)
        f32_only = (min(n, 8192) * m + min(n, 8192) + m + 4) * 4
        assert full > f32_only + 4 * m * d  # the pieces (h | m, 4 bytes per element) were asked for: this size takes the bf16 kernel
        ws_bytes = full if big else f32_only
        ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device="cuda")
        out = torch.full((n,), 123.0, device="cuda")
        rc = lib.runia_knn_kth_f32(qd.data_ptr(), bd.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, n, m, d, k,
                                   torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return out.cpu().numpy()

    # (1) unit vectors, some queries copies of bank rows
    bank = rng.standard_normal((m, d)).astype(np.float32)
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    q = rng.standard_normal((n, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[:100] = bank[500:600]
    a, b = run(q, bank, True), run(q, bank, False)
    assert not (a == 123.0).any() and np.array_equal(a, b)
    rows = [0, 1, 99, 100, n // 2, n - 1]
    assert rel_err(a[rows], oracle.knn_kth_score(bank, q[rows], k, normalize=False)) < 1e-5
    # (2) un-normalised rows over six orders of magnitude, a block of copied bank rows, NaN / infinite rows
    bank2 = (rng.standard_normal((m, d)) * 10.0 ** rng.uniform(-3, 3, size=(m, 1))).astype(np.float32)
    bank2[1000:1700] = bank2[17]
    bank2[5, min(3, d - 1)] = np.nan
    bank2[6, 0] = np.inf
    q2 = (rng.standard_normal((n, d)) * 10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)
    q2[3] = bank2[17]
    q2[7, min(1, d - 1)] = np.nan
    q2[8, :] = np.inf
    a, b = run(q2, bank2, True), run(q2, bank2, False)
    assert not (a == 123.0).any() and np.array_equal(a, b)
    assert a[7] == -oracle.FLT_MAX and a[8] == -oracle.FLT_MAX
    rows = [0, 3, 9, n - 1]
    assert rel_err(a[rows], oracle.knn_kth_score(bank2, q2[rows], k, normalize=False)) < 1e-5


def test_knn_all_baselines_mean(hip, ref_vectors):
    from test_oracle_goldens import _all_baselines_inputs

    d = _all_baselines_inputs()
    got = hip.knn_kth(hip.l2_normalize(dev(d["ood_f"], torch.float32)), hip.l2_normalize(dev(d["tr_f"], torch.float32)), 10)
    assert abs(float(got.cpu().numpy().mean()) - ref_vectors["all_baselines_means"]["scalars"][1]["value"]) < 1e-6


# ---------------- a9 KDE -------------------------------------------------------------------------------
def test_kde_goldens(hip):
    g = load_npz("ref_kde.npz")
    for name in ("unit", "baselines", "d12"):
        s = hip.kde_score(dev(g[f"{name}_train"], torch.float64), dev(g[f"{name}_test"], torch.float64)).cpu().numpy()
        assert rel_err(s, g[f"{name}_scores"]) < 1e-9, name
    # exact definition where the reference's tree evaluation is not converged (KDE quirk)
    s = hip.kde_score(dev(g["quirk64_train"], torch.float64), dev(g["quirk64_test"], torch.float64)).cpu().numpy()
    assert rel_err(s, oracle.kde_score(g["quirk64_train"], g["quirk64_test"])) < 1e-11


# ---------------- a1 mc_stack ------------------------------------------------------------------------------
@pytest.mark.parametrize("c,h,w,bs,p,n_mc,n", [(512, 4, 4, 2, 0.5, 16, 5), (20, 8, 8, 8, 0.5, 3, 2), (37, 7, 5, 3, 0.3, 6, 3),
                                               (300, 8, 8, 4, 0.4, 16, 2), (64, 4, 4, 2, 0.0, 4, 2), (70, 7, 7, 3, 0.4, 8, 3),
                                               (33, 2, 2, 1, 0.5, 5, 4), (40, 16, 16, 5, 0.3, 4, 2), (512, 4, 4, 2, 0.9, 16, 40),
                                               # maps of more than 64 positions: one contraction per image on the matrix cores
                                               (70, 14, 14, 5, 0.4, 16, 3), (33, 28, 28, 7, 0.3, 16, 2), (130, 32, 32, 4, 0.2, 16, 2),
                                               (20, 9, 11, 2, 0.3, 33, 2), (17, 14, 14, 14, 0.9, 40, 2), (64, 16, 16, 3, 0.0, 5, 2),
                                               (100, 13, 5, 3, 0.5, 7, 3), (5, 16, 31, 12, 0.1, 33, 1), (9, 10, 10, 3, 0.3, 48, 2),
                                               # small maps added to the register-resident kernel
                                               (50, 3, 3, 2, 0.4, 16, 4), (70, 5, 5, 3, 0.3, 16, 3), (40, 6, 6, 2, 0.5, 9, 3)])
def test_mc_stack_vs_oracle(hip, c, h, w, bs, p, n_mc, n):
    torch.manual_seed(c + h)
    x = torch.relu(torch.randn(n, c, h, w))
    rand = torch.rand(n, n_mc, h, w)
    got = hip.mc_stack(x.cuda(), rand.cuda() if p > 0 else None, n_mc, p, bs).cpu().numpy().reshape(n, n_mc, c)
    for i in range(n):
        with np.errstate(all="ignore"):
            exp = oracle.mc_stack(x[i : i + 1].numpy(), rand[i].numpy(), p, bs)
        ok = np.isfinite(exp)
        assert np.array_equal(np.isfinite(got[i]), ok)
        assert np.allclose(got[i][ok], exp[ok], rtol=4e-7, atol=1e-9)
    # shared draws for the whole batch (rand_image_stride = 0)
    if p > 0:
        got_s = hip.mc_stack(x.cuda(), rand[0].cuda(), n_mc, p, bs).cpu().numpy().reshape(n, n_mc, c)
        assert np.array_equal(got_s[0], got[0], equal_nan=True)


SAMPLER_CASES = ["c4x4_bs2", "c4x4_bs2_mc32", "c7x7_bs3", "c8x8_bs8", "c8x8_bs4", "c2x2_bs1", "c2x2_dead",
                 "c4x4_bs3_mc8", "c5x6_bs2", "c4x4_p0", "fc4x4_bs2", "rpn7x7_bs3"]


@pytest.mark.parametrize("name", SAMPLER_CASES)
def test_mc_stack_reference_run_fixture(hip, name):
    """The kernels against what the reference's own MCSamplerModule.forward returned (tests/golden/ref_sampler.npz,
    tools/make_goldens_r2.py): bit for bit, every kernel that serves the shape (register kernel, keep-flag table
    path, generic kernel, flattened "FC"/"RPN" kernel), NaN of a fully dropped map included."""
    g = load_npz("ref_sampler.npz")
    n_mc, bs, p, lt, _ = g[f"{name}_params"]
    n_mc, bs, p, lt = int(n_mc), int(bs), float(p), int(lt)
    x, draws, ref = dev(g[f"{name}_x"], torch.float32), dev(g[f"{name}_draws"], torch.float32), g[f"{name}_out"]
    _, c, h, w = x.shape
    if lt != 0:
        got = hip.mc_drop_flat(x, draws if p > 0 else None, n_mc, p, bs).cpu().numpy()
        assert np.array_equal(got, ref, equal_nan=True)
        return
    got = hip.mc_stack(x, draws if p > 0 else None, n_mc, p, bs).cpu().numpy()       # table path where supported
    assert np.array_equal(got, ref, equal_nan=True)
    lib = hip.load_library()
    out = torch.empty((n_mc, c), dtype=torch.float32, device="cuda")
    rc = lib.runia_mc_stack_f32(x.data_ptr(), draws.data_ptr() if p > 0 else None, 0, out.data_ptr(), 1, c, h, w, n_mc,
                                p, bs, torch.cuda.current_stream().cuda_stream)      # register / generic kernel
    assert rc == 0
    assert np.array_equal(out.cpu().numpy(), ref, equal_nan=True)
    # batch of 3 copies with per-image draws: every image reproduces the fixture
    got3 = hip.mc_stack(x.repeat(3, 1, 1, 1), draws[None].repeat(3, 1, 1, 1) if p > 0 else None, n_mc, p, bs).cpu().numpy()
    assert np.array_equal(got3.reshape(3, n_mc, c)[2], ref, equal_nan=True)


def test_sampler_module_layer_types(hip):
    """MCSamplerModule drop-in: "Conv" -> (n_mc, C), "FC"/"RPN" -> (n_mc, C*H*W); eval mode = identity layers."""
    from runia_core_amd.feature_extraction.abstract_classes import MCSamplerModule

    g = load_npz("ref_sampler.npz")
    for name, lt in (("c4x4_bs2", "Conv"), ("fc4x4_bs2", "FC"), ("rpn7x7_bs3", "RPN"), ("c7x7_bs3", "Conv")):
        n_mc, bs, p, _, seed = g[f"{name}_params"]
        m = MCSamplerModule(mc_samples=int(n_mc), block_size=int(bs), drop_prob=float(p), layer_type=lt)
        m.train()
        torch.manual_seed(int(seed))  # the module draws from the CPU default generator exactly as upstream
        got = m(torch.from_numpy(g[f"{name}_x"])).cpu().numpy()
        assert np.array_equal(got, g[f"{name}_out"], equal_nan=True), name
    m = MCSamplerModule(mc_samples=4, block_size=2, drop_prob=0.5)
    m.eval()
    assert np.array_equal(m(torch.from_numpy(g["eval_x"])).cpu().numpy(), g["eval_out"])


def test_nan_and_inf_activations_give_nan_entropies(hip):
    """A NaN MC sample makes the reference's entropy NaN (its f64 tree query / log propagate it); the kernels' min/max sort
    would silently drop it, so every entropy kernel probes for it: unfused per-dimension and joint kernels, generic-k
    kernel, fused sampler + entropy (NaN / infinite activation, fully dropped map).  Clean columns are untouched."""
    rng = np.random.default_rng(2)
    n_img, n_mc, d = 5, 16, 40
    z = rng.standard_normal((n_img * n_mc, d)).astype(np.float32)
    clean = hip.kl_entropy_per_dim(dev(z, torch.float32), n_mc, 5).cpu().numpy()
    cleanj = hip.kl_entropy_joint(dev(z, torch.float32), n_mc, 5).cpu().numpy()
    z2 = z.copy()
    z2[2 * n_mc + 3, 7] = np.nan
    got = hip.kl_entropy_per_dim(dev(z2, torch.float32), n_mc, 5).cpu().numpy()
    assert np.isnan(got[2, 7]) and np.isnan(got).sum() == 1
    mask = np.ones_like(got, bool)
    mask[2, 7] = False
    assert np.array_equal(got[mask], clean[mask])
    gj = hip.kl_entropy_joint(dev(z2, torch.float32), n_mc, 5).cpu().numpy()
    assert np.isnan(gj[2]) and np.array_equal(np.delete(gj, 2), np.delete(cleanj, 2))
    g3 = hip.kl_entropy_per_dim(dev(z2, torch.float32), n_mc, 3).cpu().numpy()  # generic-k kernel
    assert np.isnan(g3[2, 7]) and np.isnan(g3).sum() == 1
    # fused: NaN and infinite activations
    x = np.maximum(rng.standard_normal((3, 24, 4, 4)), 0).astype(np.float32)
    rand = rng.random((3, 16, 4, 4)).astype(np.float32)
    rand[:, :, 0, 0] = np.maximum(rand[:, :, 0, 0], 0.2)
    base = hip.mc_entropy(dev(x, torch.float32), dev(rand, torch.float32), 16, 0.5, 2, 5).cpu().numpy()
    assert np.isfinite(base).all()
    x2 = x.copy()
    x2[1, 5, 2, 2] = np.nan
    x2[2, 9, 0, 1] = np.inf
    got = hip.mc_entropy(dev(x2, torch.float32), dev(rand, torch.float32), 16, 0.5, 2, 5).cpu().numpy()
    assert np.isnan(got[1, 5]) and not np.isfinite(got[2, 9])
    keep = np.ones_like(got, bool)
    keep[1, 5] = keep[2, 9] = False
    assert np.array_equal(got[keep], base[keep])


# ---------------- throughput-mode draws (counter generator inside the keep-flag kernel) ---------------------------
@pytest.mark.parametrize("n,n_mc,h,w,first", [(5, 16, 4, 4, 0), (3, 16, 7, 7, 10), (2, 12, 8, 8, 2**33), (300, 32, 4, 4, 65530),
                                              (4, 16, 2, 2, 1)])
def test_counter_draws_equal_oracle(hip, n, n_mc, h, w, first):
    got = hip.mc_draws(n, n_mc, h, w, seed=0xDEADBEEF12345, first_image=first).cpu().numpy()
    assert np.array_equal(got, oracle.counter_draws(n, n_mc, h, w, 0xDEADBEEF12345, first))


@pytest.mark.parametrize("c,h,w,bs,p,n_mc,n", [(512, 4, 4, 2, 0.5, 16, 33), (64, 4, 4, 2, 0.4, 32, 9), (48, 8, 8, 3, 0.4, 12, 5),
                                               (40, 7, 7, 3, 0.4, 16, 5), (64, 2, 2, 1, 0.3, 16, 7), (24, 7, 7, 3, 0.4, 32, 4),
                                               (16, 8, 8, 3, 0.5, 32, 3), (16, 7, 7, 2, 0.5, 21, 70)])
def test_counter_mode_equals_explicit_draws(hip, c, h, w, bs, p, n_mc, n):
    """In-kernel draws (runia_mc_entropy_counter_f32) == the parity path fed with the same generator's explicit draws
    (runia_mc_draws_f32 -> runia_mc_entropy_f32), bit for bit; chunks and shards line up through first_image."""
    torch.manual_seed(c)
    x = torch.relu(torch.randn(n, c, h, w)).cuda()
    seed, first = 77, 1000
    explicit = hip.mc_draws(n, n_mc, h, w, seed, first)
    h_par = hip.mc_entropy(x, explicit, n_mc, p, bs, 5)
    h_ctr = hip.mc_entropy(x, hip.CounterDraws(seed, first), n_mc, p, bs, 5)
    assert torch.equal(torch.nan_to_num(h_ctr, nan=-7.0), torch.nan_to_num(h_par, nan=-7.0))
    # two-call form (bench's bracketed step) and a shifted chunk
    ev = []
    h_ev = hip.mc_entropy(x, hip.CounterDraws(seed, first), n_mc, p, bs, 5, kernel_events=ev)
    assert torch.equal(torch.nan_to_num(h_ev, nan=-7.0), torch.nan_to_num(h_par, nan=-7.0)) and len(ev) == 1
    h_tail = hip.mc_entropy(x[2:].contiguous(), hip.CounterDraws(seed, first + 2), n_mc, p, bs, 5)
    assert torch.equal(torch.nan_to_num(h_tail, nan=-7.0), torch.nan_to_num(h_par[2:], nan=-7.0))
    # sampler alone
    z_ctr = hip.mc_stack(x, hip.CounterDraws(seed, first), n_mc, p, bs)
    z_par = hip.mc_stack(x, explicit, n_mc, p, bs)
    assert torch.equal(torch.nan_to_num(z_ctr, nan=-7.0), torch.nan_to_num(z_par, nan=-7.0))


@pytest.mark.parametrize("c,h,w,bs,p,n_mc,n", [(64, 4, 4, 2, 0.9, 16, 400), (32, 4, 4, 3, 0.6, 32, 120), (24, 7, 7, 5, 0.95, 16, 60),
                                               (16, 8, 8, 7, 0.9, 12, 40), (40, 2, 2, 1, 0.8, 16, 200), (512, 4, 4, 2, 0.5, 16, 2000)])
def test_counter_redraw_of_fully_dropped_maps(hip, c, h, w, bs, p, n_mc, n):
    """CounterDraws(redraw_dead_layers=True): a drop layer that removes the whole map draws again from the image's next
    counter block (attempt = fourth Philox counter word).  Equal, bit for bit, to the parity path fed with the oracle's
    equivalent explicit draws; no NaN entropy; images without such a layer are untouched."""
    torch.manual_seed(c + n)
    x = torch.relu(torch.randn(n, c, h, w)).cuda()
    seed, first = 123, 77
    plain = oracle.counter_draws(n, n_mc, h, w, seed, first)
    dead0 = (oracle.dropblock_block_mask(plain.reshape(n * n_mc, h, w), p, bs).reshape(n, n_mc, -1).sum(2) == 0)
    assert dead0.any() or p <= 0.5  # the aggressive cases do contain fully dropped maps (several attempts deep)
    redrawn = oracle.counter_draws_redrawn(n, n_mc, h, w, seed, first, p, bs)
    h_exp = hip.mc_entropy(x, torch.from_numpy(redrawn).cuda(), n_mc, p, bs, 5)
    h_got = hip.mc_entropy(x, hip.CounterDraws(seed, first, True), n_mc, p, bs, 5)
    assert torch.equal(torch.nan_to_num(h_got, nan=-7.0), torch.nan_to_num(h_exp, nan=-7.0))
    still_dead = oracle.dropblock_block_mask(redrawn.reshape(n * n_mc, h, w), p, bs).reshape(n, n_mc, -1).sum(2) == 0
    assert not torch.isnan(h_got[torch.from_numpy(~still_dead.any(axis=1)).cuda()]).any()
    if p < 0.93:
        assert not still_dead.any() and not torch.isnan(h_got).any()
    # images that had no fully dropped map score exactly what the plain counter mode gives them
    h_plain = hip.mc_entropy(x, hip.CounterDraws(seed, first), n_mc, p, bs, 5)
    clean = torch.from_numpy(~dead0.any(axis=1)).cuda()
    assert torch.equal(h_plain[clean], h_got[clean])
    assert torch.isnan(h_plain[~clean]).all() if (~clean).any() else True
    # chunks line up through first_image; the table-only entry honours the flag too
    h_tail = hip.mc_entropy(x[3:].contiguous(), hip.CounterDraws(seed, first + 3, True), n_mc, p, bs, 5)
    assert torch.equal(torch.nan_to_num(h_tail, nan=-7.0), torch.nan_to_num(h_got[3:], nan=-7.0))
    tab = hip.mc_mask_table(hip.CounterDraws(seed, first, True), n, h, w, n_mc, p, bs)
    h_tab = hip.mc_entropy(x, None, n_mc, p, bs, 5, table=tab)
    assert torch.equal(torch.nan_to_num(h_tab, nan=-7.0), torch.nan_to_num(h_got, nan=-7.0))
    with pytest.raises(hip.RuniaHipError):  # the explicit-draw kernels cannot redraw: refused, not ignored
        hip.mc_drop_flat(x, hip.CounterDraws(seed, first, True), n_mc, p, bs)


# ---------------- full chain on pre-stacked samples -------------------------------------------------------------
def test_larem_chain_unfused(hip):
    rng = np.random.default_rng(77)
    n_img, n_mc, d, n = 96, 16, 512, 256
    base = rng.standard_normal((n_img, 1, d)) + 2.0
    z = (base * (1 + 0.1 * rng.standard_normal((n_img, n_mc, d))) + 0.05 * rng.standard_normal((n_img, n_mc, d)))
    z = z.reshape(n_img * n_mc, d).astype(np.float32)
    g = load_npz("ref_pca.npz")
    comp, bias, scale = _pca_state(g, "d512")
    m = load_npz("ref_md.npz")
    mean, prec = m["d256_mean"], m["d256_precision"]
    exp, h_exp = oracle.larem_pipeline(z, n_mc, comp, g["d512_mean"], g["d512_var"], mean, prec)
    h = hip.kl_entropy_per_dim(dev(z, torch.float32), n_mc, 5)
    y = hip.pca_transform(h, hip.pack_weights(dev(comp.T.copy(), torch.float64)), dev(bias, torch.float64),
                          dev(scale, torch.float64), n)
    s = hip.md_score(y, dev(mean.ravel(), torch.float64), hip.pack_weights(dev(prec, torch.float64))).cpu().numpy()
    assert np.abs(h.cpu().numpy() - h_exp).max() < 1e-11
    assert rel_err(s, exp) < 1e-9


# ---------------- a11 fused launches --------------------------------------------------------------------
@pytest.mark.parametrize("c,h,w,bs,p,n_mc,n", [(512, 4, 4, 2, 0.5, 16, 33), (100, 4, 4, 2, 0.7, 32, 5), (64, 4, 4, 3, 0.5, 7, 4),
                                               (70, 7, 7, 3, 0.4, 16, 3), (130, 8, 8, 4, 0.4, 12, 3), (33, 2, 2, 1, 0.5, 16, 6),
                                               (512, 4, 4, 2, 0.0, 16, 3), (40, 4, 4, 4, 0.6, 16, 9), (24, 8, 8, 5, 0.5, 16, 4),
                                               (24, 8, 8, 8, 0.9, 10, 4), (70, 2, 2, 2, 0.15, 16, 5), (64, 4, 4, 7, 0.05, 16, 12),
                                               (50, 7, 7, 2, 0.5, 9, 6), (16, 7, 7, 7, 0.3, 16, 5), (30, 7, 7, 4, 0.6, 13, 7),
                                               (20, 7, 7, 1, 0.5, 16, 300), (12, 7, 7, 5, 0.99, 16, 8),
                                               # up to 32 MC samples (the reference's default mcd_samples_nro) on 7x7 / 8x8 / 2x2
                                               (40, 7, 7, 3, 0.4, 32, 5), (24, 8, 8, 3, 0.5, 25, 4), (33, 2, 2, 1, 0.15, 32, 6),
                                               (20, 7, 7, 3, 0.4, 20, 3), (16, 8, 8, 2, 0.6, 32, 3), (130, 7, 7, 2, 0.3, 17, 2)])
def test_mc_entropy_fused_equals_unfused(hip, c, h, w, bs, p, n_mc, n):
    torch.manual_seed(c + n_mc)
    x = torch.relu(torch.randn(n, c, h, w)).cuda()
    rand = torch.rand(n, n_mc, h, w).cuda() if p > 0 else None
    k = 5
    assert hip.mc_entropy_supported(h, w, n_mc, k)
    hf, zf = hip.mc_entropy(x, rand, n_mc, p, bs, k, want_samples=True)
    z = hip.mc_stack(x, rand, n_mc, p, bs)
    hu = hip.kl_entropy_per_dim(z, n_mc, k)
    # same samples (as multisets per image/channel: the fused kernel visits drop layers in mask-sum order)
    zs = np.sort(z.cpu().numpy().reshape(n, n_mc, c), axis=1)
    zfs = np.sort(zf.cpu().numpy().reshape(n, n_mc, c), axis=1)
    assert np.array_equal(zs, zfs, equal_nan=True)
    a, b = hf.cpu().numpy(), hu.cpu().numpy()
    fin = np.isfinite(zs).all(axis=1)  # a fully dropped map is NaN upstream -> NaN entropy
    assert np.array_equal(a[fin], b[fin])
    assert np.isnan(a[~fin]).all()
    exp = oracle.kl_entropy_per_dim_vectorized(np.where(np.isfinite(z.cpu().numpy()), z.cpu().numpy(), 0.0), n_mc, k)
    assert fin.any() and np.abs(a[fin] - exp[fin]).max() < 1e-11


def test_mc_entropy_two_call_form_equals_one_call(hip):
    """runia_mc_mask_table_f32 + runia_mc_entropy_from_table_f32 (what bench.py times) = runia_mc_entropy_f32."""
    torch.manual_seed(5)
    x = torch.relu(torch.randn(37, 512, 4, 4)).cuda()
    rand = torch.rand(37, 16, 4, 4).cuda()
    ev = []
    a = hip.mc_entropy(x, rand, 16, 0.5, 2, 5, kernel_events=ev)
    b = hip.mc_entropy(x, rand, 16, 0.5, 2, 5)
    torch.cuda.synchronize()
    assert len(ev) == 1 and ev[0][0].elapsed_time(ev[0][1]) > 0
    assert torch.equal(a, b)


def test_mc_entropy_unsupported_shape_is_refused(hip):
    assert not hip.mc_entropy_supported(5, 5, 16, 5) and not hip.mc_entropy_supported(4, 4, 3, 2)
    x = torch.rand(2, 8, 5, 5, device="cuda")
    with pytest.raises(hip.RuniaHipError):
        hip.mc_entropy(x, torch.rand(2, 16, 5, 5, device="cuda"), 16, 0.5, 2, 5)


@pytest.mark.parametrize("n_rows,d,n,pca", [(1, 20, 4, True), (33, 512, 256, True), (1000, 512, 256, True), (40000, 64, 16, True),
                                            (70, 100, 300, True), (50, 24, 24, False), (17, 300, 300, False)])
def test_pca_md_fused_equals_unfused(hip, n_rows, d, n, pca):
    rng = np.random.default_rng(n_rows + d + n)
    h = rng.standard_normal((n_rows, d))
    a = rng.standard_normal((n, n))
    prec = a @ a.T / n + np.eye(n)
    md_mean = rng.standard_normal(n) * 0.3
    packed_p = hip.pack_weights(dev(prec, torch.float64))
    if pca:
        comp = rng.standard_normal((n, d)) / np.sqrt(d)
        mean = rng.standard_normal(d)
        var = rng.random(n) + 0.2
        bias = (mean.reshape(1, -1) @ comp.T).ravel()
        packed_ct = hip.pack_weights(dev(comp.T.copy(), torch.float64))
        s, y = hip.pca_md_score(dev(h, torch.float64), packed_ct, dev(bias, torch.float64), dev(np.sqrt(var), torch.float64),
                                dev(md_mean, torch.float64), packed_p, n, want_projection=True)
        y_ref = oracle.pca_transform(h, comp, mean, var)
        yu = hip.pca_transform(dev(h, torch.float64), packed_ct, dev(bias, torch.float64), dev(np.sqrt(var), torch.float64), n)
        assert torch.equal(y, yu)
        assert rel_err(y.cpu().numpy(), y_ref) < 1e-11
    else:
        s = hip.pca_md_score(dev(h, torch.float64), None, None, None, dev(md_mean, torch.float64), packed_p, n)
        y_ref = h
    assert rel_err(s.cpu().numpy(), oracle.md_score(y_ref, md_mean.reshape(1, -1), prec)) < 1e-11


# ---------------- edge cases: empty and single-row inputs through every entry point --------------------------
def test_empty_and_single_row_inputs(hip):
    e = lambda *shape, dt=torch.float32: torch.empty(shape, dtype=dt, device="cuda")  # noqa: E731
    r = lambda *shape, dt=torch.float32: torch.rand(shape, dtype=dt, device="cuda")  # noqa: E731
    assert hip.kl_entropy_per_dim(e(0, 8), 4, 3).shape == (0, 8)
    assert hip.kl_entropy_joint(e(0, 8), 4, 3).shape == (0,)
    assert hip.mc_stack(e(0, 8, 4, 4), e(0, 4, 4, 4), 4, 0.5, 2).shape == (0, 8)
    assert hip.mc_entropy(e(0, 8, 4, 4), e(0, 16, 4, 4), 16, 0.5, 2, 5).shape == (0, 8)
    assert hip.l2_normalize(e(0, 8)).shape == (0, 8)
    assert hip.knn_kth(e(0, 8), r(5, 8), 2).shape == (0,)
    assert hip.kde_score(r(5, 3, dt=torch.float64), e(0, 3, dt=torch.float64)).shape == (0,)
    assert hip.gen_score(e(0, 7), 0.1, 3).shape == (0,)
    assert hip.ash_s(e(0, 7), 50).shape == (0, 7)
    assert hip.linear(e(0, 7), r(3, 7), r(3)).shape == (0, 3)
    p = hip.pack_weights(r(8, 4, dt=torch.float64))
    assert hip.pca_transform(e(0, 8, dt=torch.float64), p, r(4, dt=torch.float64), r(4, dt=torch.float64), 4).shape == (0, 4)
    pp = hip.pack_weights(torch.eye(4, dtype=torch.float64, device="cuda"))
    assert hip.md_score(e(0, 4, dt=torch.float64), r(4, dt=torch.float64), pp).shape == (0,)
    assert hip.pca_md_score(e(0, 8, dt=torch.float64), p, r(4, dt=torch.float64), r(4, dt=torch.float64), r(4, dt=torch.float64), pp, 4).shape == (0,)
    assert hip.mahalanobis_score(e(0, 4), r(3, 4), pp, r(3, 4, dt=torch.float64)).shape == (0,)
    # single rows against the oracle
    rng = np.random.default_rng(9)
    z = rng.standard_normal((16, 5)).astype(np.float32)
    assert np.abs(hip.kl_entropy_per_dim(dev(z, torch.float32), 16, 5).cpu().numpy() - oracle.kl_entropy_per_dim_vectorized(z, 16)).max() < 1e-12
    x = rng.standard_normal((1, 1000)).astype(np.float32)
    lse, msp = hip.row_lse_msp(dev(x, torch.float32), True, True)
    assert rel_err(lse.cpu().numpy(), oracle.energy_score(x)) < 1e-6 and rel_err(msp.cpu().numpy(), oracle.msp_score(x)) < 1e-6
    # invalid arguments are refused, not launched
    with pytest.raises(hip.RuniaHipError):
        hip.kl_entropy_per_dim(r(8, 4), 4, 4)  # k must be < n_mc
    with pytest.raises(hip.RuniaHipError):
        hip.kl_entropy_per_dim(r(130, 4), 65, 5)  # n_mc > 64


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows", [1, 100, 2100, 10000, 70000])
def test_proj_sq_accumulate_equals_store_form(hip, n_rows):
    """runia_proj_sq_accumulate_f64 (column halves added into a cleared vector) = runia_proj_sq_score_f64, bit for bit,
    whatever tile shape the row count selects."""
    torch.manual_seed(n_rows)
    d, r = 96, 40
    h = torch.randn(n_rows, d, dtype=torch.float64, device="cuda")
    m = torch.randn(d, r, dtype=torch.float64, device="cuda") * 0.1
    c = torch.randn(r, dtype=torch.float64, device="cuda")
    pm = hip.pack_weights(m)
    a = hip.proj_sq_score(h, pm, c, r)
    out = torch.zeros(n_rows, dtype=torch.float64, device="cuda")
    b = hip.proj_sq_accumulate(h, pm, c, r, out)
    assert torch.equal(a, b)
    ref = -((h @ m + c) ** 2).sum(1)
    assert float(((a - ref).abs() / ref.abs().clamp_min(1.0)).max()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("d,r", [(512, 256), (96, 256), (520, 200), (512, 130)])
def test_proj_sq_bits_do_not_depend_on_the_launch_shape(hip, d, r):
    """K2' picks its kernel by the row count - one workgroup per 16-row tile over all columns, 16 x 128 column halves, 32 x 256
    tiles for whole rounds - and by the widths (rows staged by buffer_load ... lds when D is a whole number of chunks and the
    grid is resident at once, through registers otherwise).  All of them add a row's squares in the same column-group order:
    a row scores the same bits whatever batch it sits in, stored or accumulated, whatever the alignment of H."""
    torch.manual_seed(d + r)
    n = 70000
    h = torch.randn(n, d, dtype=torch.float64, device="cuda")
    m = torch.randn(d, r, dtype=torch.float64, device="cuda") * 0.1
    c = torch.randn(r, dtype=torch.float64, device="cuda")
    pm = hip.pack_weights(m)
    whole = hip.proj_sq_score(h, pm, c, r)
    ref = -((h[:3000] @ m + c) ** 2).sum(1)
    assert float(((whole[:3000] - ref).abs() / ref.abs().clamp_min(1.0)).max()) < 1e-12
    for a, b in ((0, 1), (5, 105), (1000, 3100), (20000, 30000), (30000, 45000), (3, 65536 + 3)):  # 15 000 rows: register-staged 16-row form
        part = hip.proj_sq_score(h[a:b].contiguous(), pm, c, r)
        assert torch.equal(part, whole[a:b]), (a, b)
        acc = torch.zeros(b - a, dtype=torch.float64, device="cuda")
        hip.proj_sq_accumulate(h[a:b].contiguous(), pm, c, r, acc)
        assert torch.equal(acc, whole[a:b]), (a, b, "accumulate")
    # rows that start 8 bytes off a 16-byte boundary
    odd = torch.empty(n * d + 1, dtype=torch.float64, device="cuda")[1:].view(n, d)
    odd.copy_(h)
    assert odd.data_ptr() % 16 == 8
    assert torch.equal(hip.proj_sq_score(odd[:10000], pm, c, r), whole[:10000])


@pytest.mark.parametrize("c,h,w,bs,p,n_mc,n", [(512, 4, 4, 2, 0.5, 16, 9), (33, 8, 8, 3, 0.4, 12, 3), (70, 7, 7, 3, 0.4, 16, 3),
                                               (20, 2, 2, 1, 0.3, 16, 5), (64, 4, 4, 2, 0.0, 16, 2)])
def test_mc_stack_table_path_equals_register_kernel(hip, c, h, w, bs, p, n_mc, n):
    """runia_mc_stack_table_f32 (what hip.mc_stack takes for the supported map shapes) = runia_mc_stack_f32, same
    samples in the same order, bit for bit."""
    torch.manual_seed(c)
    x = torch.relu(torch.randn(n, c, h, w)).cuda()
    rand = torch.rand(n, n_mc, h, w).cuda() if p > 0 else None
    a = hip.mc_stack(x, rand, n_mc, p, bs)
    lib = hip.load_library()
    b = torch.empty_like(a)
    rc = lib.runia_mc_stack_f32(x.data_ptr(), None if rand is None else rand.data_ptr(), 0 if rand is None else n_mc * h * w,
                                b.data_ptr(), n, c, h, w, n_mc, float(p), bs, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize("m,n,d,bw", [(300, 70, 100, 3.0), (1000, 257, 256, 6.0), (50, 5, 65, 2.0), (4097, 33, 512, 9.0)])
def test_kde_matrix_core_path_equals_exact_differences(hip, m, n, d, bw):
    """runia_kde_score_packed_f64 (|x|^2 + |t|^2 - 2 x.t on the f64 MFMA, online logsumexp) against the oracle's
    exact-difference log-density and against the direct kernel."""
    rng = np.random.default_rng(m + d)
    train = rng.standard_normal((m, d))
    x = rng.standard_normal((n, d)) * 1.1 + 0.1
    td, xd = torch.from_numpy(train).cuda(), torch.from_numpy(x).cuda()
    got = hip.kde_score_packed(hip.kde_pack_train(td), xd, bw).cpu().numpy()
    exp = oracle.kde_score(train, x, bw)
    assert np.abs(got - exp).max() / max(1.0, np.abs(exp).max()) < 1e-11
    direct = hip.kde_score(td, xd, bw).cpu().numpy()
    assert np.abs(got - direct).max() < 1e-9
    # embeddings far from the origin: the training mean is removed at setup, so the norm expansion stays exact
    far = hip.kde_score_packed(hip.kde_pack_train(td + 1.0e4), xd + 1.0e4, bw).cpu().numpy()
    assert np.abs(far - exp).max() / max(1.0, np.abs(exp).max()) < 1e-9


def test_knn_config_switch_keeps_the_f32_kernel(hip):
    """runia_core_amd.config.knn_bf16_candidates = False: hip.knn_kth hands the entry point the f32-sized workspace, which keeps
    the f32 matrix-core kernel; the scores are the same bits as with the bf16 candidate kernel."""
    from runia_core_amd import config

    torch.manual_seed(5)
    bank = torch.nn.functional.normalize(torch.randn(6000, 512, device="cuda"), dim=1)
    q = torch.nn.functional.normalize(torch.randn(1500, 512, device="cuda"), dim=1)
    assert hip.load_library().runia_knn_piece_products(1500, 6000, 512) == 3
    a = hip.knn_kth(q, bank, 50)
    try:
        config.knn_bf16_candidates = False
        b = hip.knn_kth(q, bank, 50)
    finally:
        config.knn_bf16_candidates = True
    assert torch.equal(a, b)


def test_knn_handful_of_queries_scores_the_bits_of_a_batch(hip):
    """Up to 12 queries against a bank of >= 1 024 rows take one pass over the bank with exact f32 distances
    (knn_one_dist_kernel for one or two, knn_small_dist_kernel - queries in LDS, two bank rows per wave - beyond) instead of
    the matrix-core tiles.  A row scores the same bits alone, among a handful and
    inside a large batch (f32 kernel and bf16 candidate kernel), on unit vectors, un-normalised rows, copied bank rows and
    NaN / infinite rows; the oracle on top."""
    rng = np.random.default_rng(21)
    m, d = 6000, 512
    bank = rng.standard_normal((m, d)).astype(np.float32)
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    bank[1000:1600] = bank[3]      # 600 copies: ties, the crowded-window path
    q = rng.standard_normal((1500, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = bank[3]
    q[5] = bank[77]
    bd, qd = dev(bank, torch.float32), dev(q, torch.float32)
    for k in (1, 50, 599, 601, 2000):
        whole = hip.knn_kth(qd, bd, k)                       # bf16 candidate kernel (1 500 x 6 000 x 512)
        mid = hip.knn_kth(qd[:100].contiguous(), bd, k)      # f32 matrix-core kernel
        assert torch.equal(mid, whole[:100]), k
        for a, b in ((0, 1), (5, 6), (0, 8), (3, 7), (92, 100), (0, 12), (83, 95), (40, 49), (4, 6), (0, 3)):
            few = hip.knn_kth(qd[a:b].contiguous(), bd, k)   # the handful path
            assert torch.equal(few, whole[a:b]), (k, a, b)
        exp = oracle.knn_kth_score(bank, q[:8], k, normalize=False)
        assert rel_err(hip.knn_kth(qd[:8].contiguous(), bd, k).cpu().numpy(), exp) < 1e-5, k
    # un-normalised rows over six orders of magnitude, NaN / infinite rows, a width that is not a multiple of 64
    m2, d2 = 3000, 100
    bank2 = (rng.standard_normal((m2, d2)) * 10.0 ** rng.uniform(-3, 3, size=(m2, 1))).astype(np.float32)
    bank2[4, 1] = np.nan
    bank2[9, 0] = np.inf
    q2 = (rng.standard_normal((8, d2)) * 10.0 ** rng.uniform(-3, 3, size=(8, 1))).astype(np.float32)
    q2[2, 0] = np.nan
    q2[6, :] = np.inf
    for k in (1, 50, 2998, 2999, 3000):
        got = hip.knn_kth(dev(q2, torch.float32), dev(bank2, torch.float32), k).cpu().numpy()
        exp = oracle.knn_kth_score(bank2, q2, k, normalize=False)
        assert got[2] == -oracle.FLT_MAX and got[6] == -oracle.FLT_MAX
        assert np.isfinite(got).all() and rel_err(got, exp) < 1e-5, k
        many = hip.knn_kth(dev(np.concatenate([q2, q2, q2]), torch.float32), dev(bank2, torch.float32), k).cpu().numpy()
        assert np.array_equal(many[:8], got) and np.array_equal(many[8:16], got), k
    # bank sizes around the four rows a wave takes per trip
    for m3 in (1025, 1026, 1027, 2999):
        b3, q3 = dev(bank2[:m3], torch.float32), dev(q2[[0, 1, 3, 4, 5]], torch.float32)
        got = hip.knn_kth(q3, b3, 50).cpu().numpy()
        assert rel_err(got, oracle.knn_kth_score(bank2[:m3], q2[[0, 1, 3, 4, 5]], 50, normalize=False)) < 1e-5, m3
        big = hip.knn_kth(dev(np.tile(q2[[0, 1, 3, 4, 5]], (5, 1)), torch.float32), b3, 50).cpu().numpy()  # 25 rows: the tile kernels
        assert np.array_equal(big[:5], got), m3


@pytest.mark.parametrize("d,c", [(2048, 10), (300, 16), (512, 3)])
def test_mahalanobis_small_batches_split_columns_and_keep_the_bits(hip, d, c):
    """Few row tiles (< one per compute unit): the 256-column blocks of a tile go to separate workgroups and a finishing
    launch adds their partial sums in the unsplit kernel's order (one row against a 2048 x 2048 precision: 1.6 -> 0.25 ms).
    A row scores the same bits alone, in a small batch (split launch, block-major order since round 5), in a large batch
    (one-workgroup-per-tile launch; round 5 measured large batches on the split launch too - slower, kept behind
    RUNIA_MAHA_SPLIT=1) and with the workspace withheld (``split=False``); the oracle's class loop on top."""
    torch.manual_seed(d + c)
    n = 70000
    a = torch.randn(d, d, dtype=torch.float64)
    prec = (a @ a.T / d + torch.eye(d, dtype=torch.float64)).numpy()
    cm = torch.randn(c, d).numpy().astype(np.float32)
    f = torch.relu(torch.randn(n, d) + 0.3).cuda()
    packed = hip.pack_weights(torch.from_numpy(prec).cuda())
    mu_p = torch.from_numpy(cm.astype(np.float64) @ prec).cuda()
    cmd = torch.from_numpy(cm).cuda()
    whole = hip.mahalanobis_score(f, cmd, packed, mu_p)
    one_launch = hip.mahalanobis_score(f, cmd, packed, mu_p, split=False)     # no workspace: never split
    assert torch.equal(whole, one_launch)
    for a0, b0 in ((0, 1), (5, 12), (100, 133), (1000, 1512), (2000, 7000), (60000, 70000)):
        part = hip.mahalanobis_score(f[a0:b0].contiguous(), cmd, packed, mu_p)
        assert torch.equal(part, whole[a0:b0]), (a0, b0)
        assert torch.equal(hip.mahalanobis_score(f[a0:b0].contiguous(), cmd, packed, mu_p, split=False), part)
    rows = [0, 5, 100, 65535, 65536, 69999]
    exp = oracle.mahalanobis_score(f[rows].cpu().numpy(), cm, prec, c)
    assert rel_err(whole[rows].cpu().numpy(), exp) < 1e-9


@pytest.mark.parametrize("d,c,clip", [(2048, 1000, float("inf")), (512, 100, 0.8), (300, 37, float("inf")), (33, 17, 1.0)])
def test_linear_few_rows_score_the_bits_of_a_batch(hip, d, c, clip):
    """Up to 8 rows take one thread per (row, class), 9 ... 512 rows a wave per 64 classes x 8 rows (round 4), each with the
    matrix-core kernel's fma chain (k order 0, 2, 1, 3 within every four: v_mfma_f32_32x32x2_f32 is an exact fma chain)
    instead of its 128 x 128 tiles: same bits as inside a batch of 2 000 rows (the matrix-core kernel), with and without bias
    and ReAct clip, widths that are not multiples of 4 or 32, NaN activations kept."""
    torch.manual_seed(d + c)
    x = torch.randn(2000, d, device="cuda")
    x[3, 5] = float("nan")
    w = torch.randn(c, d, device="cuda") * 0.1
    b = torch.randn(c, device="cuda")
    for bias in (b, None):
        whole = hip.linear(x, w, bias, clip)
        for a0, b0 in ((0, 1), (3, 4), (10, 18), (292, 300), (0, 9), (1, 65), (700, 1000), (1488, 2000), (3, 516)):
            few = hip.linear(x[a0:b0].contiguous(), w, bias, clip)
            same = (few == whole[a0:b0]) | (torch.isnan(few) & torch.isnan(whole[a0:b0]))
            assert bool(same.all()), (a0, b0, bias is None)


@pytest.mark.parametrize("n_mc,d,n_img", [(16, 512, 300), (16, 20, 40), (5, 64, 33), (8, 128, 20), (6, 8, 9), (12, 1024, 17), (17, 64, 11),
                                        (32, 100, 9), (32, 2050, 5), (16, 37, 12), (4, 16, 7), (33, 12, 4), (16, 1028, 3)])
def test_get_dl_h_z_single_read_equals_the_two_kernels(hip, n_mc, d, n_img):
    """runia_kl_entropy_both_f32 (round 5): the joint AND the per-dimension entropies of get_dl_h_z (reference
    evaluation/entropy.py:67-84 returns both from one call) from ONE pass over the samples - the joint kernel's threads hold
    every sample of their dims in registers and emit the per-dimension columns from there.  Same bits as the two kernels: fused
    shapes (5 <= n_mc <= 32, whole vectors per row), the fallback shapes (odd D, n_mc = 4 / 33), constant columns (min_dist
    clip), a NaN sample, lanes past the end of the last chunk (D = 1028: 257 float4 lanes), and the oracle on top."""
    rng = np.random.default_rng(n_mc * 1000 + d)
    z = (rng.standard_normal((n_img, 1, d)) + 0.2 * rng.standard_normal((n_img, n_mc, d))).astype(np.float32)
    z[1, :, :: 7] = z[1, :1, :: 7]        # columns constant over the samples
    if n_img > 2:
        z[2, n_mc // 2, d // 3] = np.nan
    zt = dev(z.reshape(n_img * n_mc, d), torch.float32)
    k = 5 if n_mc > 5 else n_mc - 1
    hj, hd = hip.kl_entropy_both(zt, n_mc, k)
    ej, ed = hip.kl_entropy_joint(zt, n_mc, k), hip.kl_entropy_per_dim(zt, n_mc, k)
    assert torch.equal(torch.nan_to_num(hj, nan=-7.0), torch.nan_to_num(ej, nan=-7.0))
    assert torch.equal(torch.nan_to_num(hd, nan=-7.0), torch.nan_to_num(ed, nan=-7.0))
    fused = bool(hip.load_library().runia_kl_entropy_both_fused(n_mc, d, k))
    assert fused == (5 <= n_mc <= 32 and d % (4 if n_mc <= 16 else 2) == 0)
    rows = [i for i in range(min(n_img, 5)) if i != 2]  # (image 2 carries the NaN: the oracle's k-d tree refuses it)
    oj, od = oracle.get_dl_h_z(z[rows].reshape(len(rows) * n_mc, d), n_mc)
    assert rel_err(hd[rows].cpu().numpy(), od) < 1e-11 and rel_err(hj[rows].cpu().numpy(), np.ravel(oj)) < 1e-11
    if n_img > 2:
        assert bool(torch.isnan(hj[2])) and bool(torch.isnan(hd[2, d // 3])) and int(torch.isnan(hd[2]).sum()) == 1
    if n_mc == 8:  # k = 4 has its own instantiation
        a, b = hip.kl_entropy_both(zt, n_mc, 4)
        assert torch.equal(torch.nan_to_num(b, nan=-7.0), torch.nan_to_num(hip.kl_entropy_per_dim(zt, n_mc, 4), nan=-7.0))
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(hip.kl_entropy_joint(zt, n_mc, 4), nan=-7.0))


@pytest.mark.parametrize("d,batch", [(1, 2), (37, 3), (64, 1), (65, 2), (127, 2), (128, 2), (129, 3), (300, 10), (767, 2), (833, 3), (1024, 2),
                                     (2048, 2)])
def test_tril_inverse_of_the_class_factors(hip, d, batch):  # (from 128: block forward substitution)
    """runia_tril_inverse_f64 (round 5; setup of GMMLatentSpace / DDU: the Cholesky factors torch's MultivariateNormal keeps,
    reference inference/postprocessors.py:490, 778): W = L^-1 per class by forward substitution, against the host's triangular
    solve; W L = I; the strict upper triangle is zero; the precision W^T W equals torch.cholesky_inverse."""
    rng = np.random.default_rng(d)
    a = rng.standard_normal((batch, d, 2 * d + 3))
    cov = a @ a.transpose(0, 2, 1) / (2 * d + 3) + 0.05 * np.eye(d)
    L = np.linalg.cholesky(cov)
    w = hip.tril_inverse(dev(L, torch.float64)).cpu().numpy()
    assert np.allclose(np.triu(w, 1), 0.0)
    for b in range(batch):
        assert np.allclose(w[b] @ L[b], np.eye(d), atol=1e-10)
        ref = torch.cholesky_inverse(torch.from_numpy(L[b])).numpy()
        assert rel_err(w[b].T @ w[b], ref) < 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("c", [4097, 5000, 21841, 128256])
def test_gen_and_mcd_uncertainty_on_heads_wider_than_4096(hip, c):
    """runia_gen_score_f32 / runia_gen_entropy_f32 / runia_mcd_uncertainty_f32 refused C > 4 096 (ImageNet-21k,
    LLM vocabularies) where the reference's torch / NumPy expressions (inference/funcs.py:347-375, 430-465) take any width:
    workgroup-per-row kernels now, against the oracle, with ties at the GEN threshold and M >= C."""
    rng = np.random.default_rng(c)
    n, n_mc = 6, 3
    lg = (rng.standard_normal((n * n_mc, c)) * 3).astype(np.float32)
    lg[1, 10:400] = lg[1, 5]      # ties among the largest probabilities
    lg[2] = 0.25                  # a constant row: every probability ties
    for m in (1, 100, 4096, c, c + 5):
        got = hip.gen_score(dev(lg, torch.float32), 0.1, m).cpu().numpy()
        assert got.dtype == np.float32 and rel_err(got, oracle.gen_score(lg, 0.1, m)) < 1e-5, m
    pr = np.exp(lg - lg.max(1, keepdims=True)).astype(np.float32)
    pr /= pr.sum(1, keepdims=True)
    assert rel_err(hip.gen_entropy(dev(pr, torch.float32), 0.5, 50).cpu().numpy(), oracle.generalized_entropy(pr, 0.5, 50)) < 1e-5
    ph, mi, probs = hip.mcd_uncertainty(dev(lg, torch.float32), n_mc, want_probs=True)
    e_ph, e_mi = oracle.predictive_uncertainty(lg, n_mc)
    assert rel_err(ph.cpu().numpy(), e_ph) < 1e-5 and rel_err(mi.cpu().numpy(), e_mi) < 1e-5
    assert rel_err(probs.cpu().numpy(), pr) < 1e-6


@pytest.mark.parametrize("m,d", [(10000, 256), (3000, 100), (700, 130)])
def test_kde_few_rows_split_columns_and_keep_the_bits(hip, m, d):
    """LaRED on few rows (fewer 16-row tiles than compute units): the 256-column blocks of a tile go to separate
    workgroups, which store their values, and a second launch replays the online logsumexp over them in block order with
    the fused kernel's own update and merges - a row scores the same bits alone, in a small batch and in a batch large
    enough for the fused launch (8 rows against 10 000 x 256: 0.6 -> 0.2 ms); the exact definition on top."""
    torch.manual_seed(m + d)
    tr = torch.randn(m, d, dtype=torch.float64, device="cuda")
    x = torch.randn(6000, d, dtype=torch.float64, device="cuda") * 1.2  # 375 tiles of 16 rows >= 256: fused launch
    st = hip.kde_pack_train(tr)
    for bw in (1.0, 4.0):
        whole = hip.kde_score_packed(st, x, bw)
        for a0, b0 in ((0, 1), (7, 20), (100, 228), (1000, 1999)):
            part = hip.kde_score_packed(st, x[a0:b0].contiguous(), bw)
            assert torch.equal(part, whole[a0:b0]), (bw, a0, b0)
        rows = [0, 7, 5999]
        exp = oracle.kde_score(tr.cpu().numpy(), x[rows].cpu().numpy(), bw)
        assert rel_err(whole[rows].cpu().numpy(), exp) < 1e-10


@pytest.mark.parametrize("n,m,d", [(3, 5000, 512), (300, 5000, 512), (1500, 6000, 512), (600, 9000, 300)])
def test_knn_prepared_bank_scores_the_same_bits(hip, n, m, d):
    """runia_knn_prepare_bank_f32 + runia_knn_kth_prepared_f32 (the bank's norms and bf16 pieces made once, as FlatL2Bank
    does at its first search) = runia_knn_kth_f32 bit for bit, on the handful path, the f32 kernel and the bf16 candidate
    kernel; with the bf16 candidates switched off too."""
    from runia_core_amd import config

    torch.manual_seed(n + m)
    bank = torch.randn(m, d, device="cuda")
    bank[100:400] = bank[7]
    q = torch.randn(n, d, device="cuda")
    q[0] = bank[7]
    state = hip.knn_prepare_bank(bank)
    for k in (1, 50, 299, 301):
        a = hip.knn_kth(q, bank, k)
        assert torch.equal(hip.knn_kth(q, bank, k, state=state), a), k
    try:
        config.knn_bf16_candidates = False
        assert torch.equal(hip.knn_kth(q, bank, 50, state=state), hip.knn_kth(q, bank, 50))
    finally:
        config.knn_bf16_candidates = True
    assert torch.equal(hip.knn_kth(q, bank, 50, state=state), hip.knn_kth(q, bank, 50))


@pytest.mark.gpu
def test_knn_candidate_filter_overflow_rounds_equal_the_f32_path(hip):
    """Candidate filter of the bf16 kernel (no Q x M matrix): rows whose list overflows - here EVERY query sits next to
    3 000 copies of one bank row, more hits below the sampled threshold than a list holds - go to the overflow list and
    through the dense kernels in rounds of KNN16_DENSE_ROWS rows (4 500 rows: three rounds); rows that do not overflow
    (the second half of the queries is random) stay on the lists.  Same bits as the f32 kernel either way."""
    lib = hip.load_library()
    n, m, d, k = 9000, 8192, 256, 50
    rng = np.random.default_rng(7)
    bank = rng.standard_normal((m, d)).astype(np.float32)
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    bank[1000:4000] = bank[17]
    q = rng.standard_normal((n, d)).astype(np.float32)
    q[:4500] = bank[17] + 0.05 * q[:4500] / np.sqrt(d)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    qd, bd = dev(q, torch.float32), dev(bank, torch.float32)

    def run(big):
        full = lib.runia_knn_workspace_bytes(n, m, d, k)
        f32_only = (min(n, 8192) * m + min(n, 8192) + m + 4) * 4
        ws_bytes = full if big else f32_only
        ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device="cuda")
        out = torch.full((n,), 123.0, device="cuda")
        rc = lib.runia_knn_kth_f32(qd.data_ptr(), bd.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, n, m, d, k,
                                   torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return out.cpu().numpy()

    assert lib.runia_knn_piece_products(n, m, d) == 3
    a, b = run(True), run(False)
    assert not (a == 123.0).any() and np.array_equal(a, b)
    rows = [0, 1, 4499, 4500, 8191, 8192, n - 1]
    assert rel_err(a[rows], oracle.knn_kth_score(bank, q[rows], k, normalize=False)) < 1e-5
    # against a prepared bank: the same bits again
    state = hip.knn_prepare_bank(bd)
    c = hip.knn_kth(qd, bd, k, state=state).cpu().numpy()
    assert np.array_equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("n_feat", [2048, 300, 257])
def test_md_few_rows_column_split_scores_the_bits_of_the_one_launch(hip, n_feat):
    """runia_md_score_ws_* (round 4): for few rows of wide features the 256-column blocks of a 16-row tile run on separate
    workgroups and a replay launch adds their products in the one-launch kernel's order - the same bits as
    runia_md_score_* (which large batches still take), for f64 rows, f32 rows with an f32 and with an f64 mean, ragged
    widths and row counts; and the oracle."""
    lib = hip.load_library()
    rng = np.random.default_rng(n_feat)
    a = rng.standard_normal((n_feat, n_feat))
    prec = a @ a.T / n_feat + np.eye(n_feat)
    packed = hip.pack_weights(dev(prec, torch.float64))
    mean = rng.standard_normal(n_feat)
    for n_rows in (1, 7, 16, 100, 513):
        x = rng.standard_normal((n_rows, n_feat))
        assert lib.runia_md_score_workspace_bytes(n_rows, n_feat) > 0
        for xdt, mdt, one in ((torch.float64, torch.float64, lib.runia_md_score_f64), (torch.float32, torch.float32, lib.runia_md_score_f32),
                              (torch.float32, torch.float64, lib.runia_md_score_f32x_f64mean)):
            xd, md = dev(x, xdt), dev(mean, mdt)
            got = hip.md_score(xd, md, packed)
            ref = torch.empty(n_rows, dtype=torch.float64, device="cuda")
            assert one(xd.data_ptr(), md.data_ptr(), packed.data_ptr(), ref.data_ptr(), n_rows, n_feat,
                       torch.cuda.current_stream().cuda_stream) == 0
            assert torch.equal(got, ref), (n_rows, xdt, mdt)
        xo, mo = x.astype(np.float32).astype(np.float64), mean
        assert rel_err(got.cpu().numpy(), -np.einsum("ij,jk,ik->i", xo - mo, prec, xo - mo)) < 1e-9
    assert lib.runia_md_score_workspace_bytes(100000, n_feat) == 0 and lib.runia_md_score_workspace_bytes(64, 200) == 0


def _gmm_case(rng, n, d, c, cond=0.05):
    """Class means, Cholesky factors (f32, as torch's MultivariateNormal keeps them) and test rows of a class-wise Gaussian."""
    loc = (rng.standard_normal((c, d)) * 0.7).astype(np.float32)
    a = rng.standard_normal((c, d, 2 * d + 3))
    cov = a @ a.transpose(0, 2, 1) / (2 * d + 3) + cond * np.eye(d)
    tril = np.linalg.cholesky(cov).astype(np.float32)
    x = (loc[rng.integers(0, c, n)] + rng.standard_normal((n, d))).astype(np.float32)
    return loc, tril, x


@pytest.mark.parametrize("n,d,c", [(1, 1, 1), (5, 3, 2), (130, 37, 3), (257, 128, 4), (300, 129, 2), (128, 256, 10), (1000, 300, 6),
                                   (513, 1024, 3), (96, 2048, 2)])
def test_gmm_log_prob_triangular_kernel_vs_torch(hip, n, d, c):
    """runia_gmm_log_prob_f32 (round 6): all class-wise Gaussian log densities from one launch, || L^-1 (x - mu) ||^2 on the f32
    matrix cores with the zero half of L^-1 skipped, against torch's own MultivariateNormal.log_prob (the reference's call,
    inference/postprocessors.py:490, 778) and scipy's logsumexp; the dense f64 form of rounds 4-5 beside it; ragged D (column
    tiles that end inside a 128 block), a NaN row, an infinite row; a workspace of one row tile gives the same bits."""
    from runia_core_amd.inference.funcs import GmmState

    rng = np.random.default_rng(100 * d + c)
    loc, tril, x = _gmm_case(rng, n, d, c)
    gmm = torch.distributions.MultivariateNormal(loc=torch.from_numpy(loc), scale_tril=torch.from_numpy(tril))
    want = gmm.log_prob(torch.from_numpy(x)[:, None, :]).numpy()
    want64 = torch.distributions.MultivariateNormal(loc=torch.from_numpy(loc).double(), scale_tril=torch.from_numpy(tril).double()
                                                    ).log_prob(torch.from_numpy(x).double()[:, None, :]).numpy()
    st = GmmState(gmm)
    xd = dev(x, torch.float32)
    got = st.log_prob_device(xd).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == (n, c)
    assert rel_err(got, want) < TOL
    # no further from the exact (f64) densities than torch's own f32 evaluation is, up to a factor and a floor of f32 rounding
    assert rel_err(got, want64) <= 4 * rel_err(want, want64) + 2e-6
    lse = st.energy_device(xd).cpu().numpy()
    assert lse.dtype == np.float32 and rel_err(lse, oracle.gmm_energy(gmm, x)) < TOL
    dense = GmmState(gmm, dense=True)
    assert rel_err(got, dense.log_prob_device(xd).cpu().numpy()) < TOL
    # same bits whatever the chunking of the rows (workspace of one row tile) and with both outputs from one call
    lib = hip.load_library()
    lp2 = torch.empty((n, c), dtype=torch.float32, device="cuda")
    lse2 = torch.empty((n,), dtype=torch.float32, device="cuda")
    one_tile = int(lib.runia_gmm_log_prob_workspace_bytes(128, d, c))
    ws = torch.empty(one_tile // 8, dtype=torch.float64, device="cuda")
    rc = lib.runia_gmm_log_prob_f32(xd.data_ptr(), st.means_dev.data_ptr(), st.w_tril.data_ptr(), st.const_dev.data_ptr(), lp2.data_ptr(),
                                    lse2.data_ptr(), ws.data_ptr(), one_tile, n, d, c, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(lp2.cpu().numpy(), got) and np.array_equal(lse2.cpu().numpy(), lse)
    # too small a workspace / no output / bad sizes are refused
    assert lib.runia_gmm_log_prob_f32(xd.data_ptr(), st.means_dev.data_ptr(), st.w_tril.data_ptr(), st.const_dev.data_ptr(), lp2.data_ptr(),
                                      None, ws.data_ptr(), 8, n + 200, d, c, None) < 0
    assert lib.runia_gmm_log_prob_f32(xd.data_ptr(), st.means_dev.data_ptr(), st.w_tril.data_ptr(), st.const_dev.data_ptr(), None,
                                      None, ws.data_ptr(), one_tile, n, d, c, None) < 0
    if n >= 5:
        xb = x.copy()
        xb[1, d // 2] = np.nan
        xb[3, 0] = np.inf
        gb = st.log_prob_device(dev(xb, torch.float32)).cpu().numpy()
        lb = st.energy_device(dev(xb, torch.float32)).cpu().numpy()
        assert np.isnan(gb[1]).all() and np.isnan(lb[1])
        assert not np.isfinite(gb[3]).any()   # torch: -inf or nan for an infinite coordinate
        keep = np.ones(n, bool)
        keep[[1, 3]] = False
        assert np.array_equal(gb[keep], got[keep]) and np.array_equal(lb[keep], lse[keep])  # other rows untouched


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,batch", [(1, 3), (2, 2), (63, 2), (64, 1), (127, 2), (128, 3), (129, 2), (200, 10), (513, 2), (767, 2), (768, 3),
                                     (833, 3), (1024, 1), (2048, 2)])  # (from 128: the panel form)
def test_cholesky_kernel_vs_lapack(hip, d, batch, dtype):
    """runia_cholesky_* (round 6; gmm_fit's jitter ladder on the device, the triangular factor of MD's precision): L L^T = a + jitter I
    against LAPACK's factor, zeros above the diagonal, info = j + 1 at the first pivot that is not positive (an indefinite and a NaN
    matrix in the same batch as good ones), same bits from run to run."""
    rng = np.random.default_rng(7 * d + batch)
    a = rng.standard_normal((batch, d, 3 * d + 5))
    cov = (a @ a.transpose(0, 2, 1) / (3 * d + 5) + 0.01 * np.eye(d)).astype(dtype)
    tt = torch.float32 if dtype == np.float32 else torch.float64
    L, info = hip.cholesky(dev(cov, tt), 0.0)
    L2, _ = hip.cholesky(dev(cov, tt), 0.0)
    assert torch.equal(L, L2) and int(info.abs().max()) == 0
    L = L.cpu().numpy()
    assert L.dtype == dtype and np.array_equal(np.triu(L, 1), np.zeros_like(L))
    ref = np.linalg.cholesky(cov.astype(np.float64))
    tol = 2e-4 if dtype == np.float32 else 1e-11
    assert np.max(np.abs(L - ref)) <= tol * max(1.0, np.abs(ref).max())
    eps = np.finfo(dtype).eps
    assert np.max(np.abs(L.astype(np.float64) @ L.astype(np.float64).transpose(0, 2, 1) - cov)) <= 8 * d * eps * np.abs(cov).max()
    # jitter enters the diagonal
    Lj, infoj = hip.cholesky(dev(cov, tt), 0.5)
    assert int(infoj.abs().max()) == 0
    assert np.max(np.abs(Lj.cpu().numpy() - np.linalg.cholesky(cov.astype(np.float64) + 0.5 * np.eye(d)))) <= tol * max(1.0, np.abs(ref).max())
    if d >= 2 and batch >= 2:
        bad = cov.copy()
        bad[0] = -np.eye(d, dtype=dtype)                     # first pivot negative
        bad[1][d - 1, d - 1] = -1.0                          # last pivot negative
        _, info_b = hip.cholesky(dev(bad, tt), 0.0)
        info_b = info_b.cpu().numpy()
        assert info_b[0] == 1 and info_b[1] == d and (info_b[2:] == 0).all()
        bad[0][0, 0] = np.nan
        assert hip.cholesky(dev(bad, tt), 0.0)[1].cpu().numpy()[0] == 1


@pytest.mark.parametrize("n,rows,xdt,mdt", [(512, 300, np.float64, np.float64), (640, 70, np.float32, np.float32), (1024, 1000, np.float32, np.float64),
                                            (2048, 9000, np.float32, np.float32), (100, 50, np.float64, np.float64), (777, 33, np.float64, np.float64)])
def test_md_score_from_the_triangular_factor(hip, n, rows, xdt, mdt):
    """runia_md_score_tril_* (round 6): -|| W (x - mean) ||^2 with precision = W^T W, W lower triangular, the zero half of W
    skipped by the kernel - against the oracle's -(x - mean) P (x - mean)^T and the dense kernel; the factor comes from
    runia_cholesky_f64 of the reversed precision (MDLatentSpace._triangular_factor)."""
    from runia_core_amd import config
    from runia_core_amd.inference import MDLatentSpace

    rng = np.random.default_rng(n + rows)
    a = rng.standard_normal((n, 2 * n + 7))
    cov = a @ a.T / (2 * n + 7) + 0.05 * np.eye(n)
    prec = np.linalg.inv(cov)
    prec = 0.5 * (prec + prec.T)
    mean = rng.standard_normal(n).astype(mdt)
    x = (rng.standard_normal((rows, n)) * 1.5 + 0.2).astype(xdt)
    tt = lambda d: torch.float32 if d == np.float32 else torch.float64  # noqa: E731
    # the factor, by hand: U upper triangular with P = U U^T
    g, info = hip.cholesky(torch.flip(dev(prec, torch.float64), dims=(0, 1)).contiguous())
    assert int(info.item()) == 0
    u = torch.flip(g, dims=(0, 1)).contiguous()
    assert np.allclose(np.tril(u.cpu().numpy(), -1), 0.0) and np.allclose((u @ u.t()).cpu().numpy(), prec, atol=1e-10 * np.abs(prec).max())
    got = hip.md_score_tril(dev(x, tt(xdt)), dev(mean, tt(mdt)), hip.pack_weights(u)).cpu().numpy()
    diff = (x - mean) if (xdt == np.float32 and mdt == np.float32) else (x.astype(np.float64) - mean.astype(np.float64))
    exp = -np.einsum("ij,jk,ik->i", diff.astype(np.float64), prec, diff.astype(np.float64))
    assert got.dtype == np.float64 and rel_err(got, exp) < 1e-11
    dense = hip.md_score(dev(x, tt(xdt)), dev(mean, tt(mdt)), hip.pack_weights(dev(prec, torch.float64))).cpu().numpy()
    assert rel_err(got, dense) < 1e-11
    # a row's bits do not depend on the batch it arrives in
    one = hip.md_score_tril(dev(x[:1], tt(xdt)), dev(mean, tt(mdt)), hip.pack_weights(u)).cpu().numpy()
    assert one[0] == got[0]
    # the postprocessor takes the factor from n = 512 upward, the P form below, and on request / for a singular precision
    md = MDLatentSpace()
    md.feats_mean, md.precision, md._setup_flag = mean.reshape(1, -1), prec, True
    s = md.postprocess(x)
    assert (md._device_state()["packed_wt"] is not None) == (n >= 512)
    assert rel_err(s, exp) < 1e-11
    if n >= 512:
        before = config.md_triangular
        try:
            config.md_triangular = False
            md2 = MDLatentSpace()
            md2.feats_mean, md2.precision, md2._setup_flag = mean.reshape(1, -1), prec, True
            assert md2._device_state()["packed_wt"] is None and rel_err(md2.postprocess(x), exp) < 1e-11
        finally:
            config.md_triangular = before
        sing = prec.copy()
        w, v = np.linalg.eigh(sing)
        w[:5] = 0.0                                   # a pinvh that dropped five directions
        sing = (v * w) @ v.T
        md3 = MDLatentSpace()
        md3.feats_mean, md3.precision, md3._setup_flag = mean.reshape(1, -1), 0.5 * (sing + sing.T), True
        s3 = md3.postprocess(x)
        assert md3._device_state()["packed_wt"] is None
        assert rel_err(s3, -np.einsum("ij,jk,ik->i", diff.astype(np.float64), md3.precision, diff.astype(np.float64))) < 1e-10


@pytest.mark.parametrize("d,n,rows", [(700, 300, 5), (700, 300, 3000), (512, 380, 2100), (2048, 1048, 9000), (1024, 536, 40000), (300, 257, 33),
                                      (640, 384, 1000)])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_proj_norm_with_a_ragged_last_column_block(hip, d, n, rows, dt):
    """runia_proj_norm_* (ViM's residual norm, reference inference/postprocessors.py:1106) when the 256-column blocks do not divide n:
    round 6 takes the whole blocks in one launch and the last <= 128 columns in a second, narrow-tile launch over the same packed
    matrix (ViM at D = 2048: 1 048 columns, four blocks + 24).  Against NumPy; a row's bits do not depend on the batch."""
    rng = np.random.default_rng(d + n + rows)
    ns = np.linalg.qr(rng.standard_normal((d, n)))[0]
    u = (rng.standard_normal(d) * 0.1).astype(dt)
    x = (rng.standard_normal((rows, d)) + 0.3).astype(dt)
    tt = torch.float32 if dt == np.float32 else torch.float64
    packed = hip.pack_weights(dev(ns, torch.float64))
    got = hip.proj_norm(dev(x, tt), dev(u, tt), packed, n).cpu().numpy()
    diff = (x - u) if dt == np.float32 else (x.astype(np.float64) - u.astype(np.float64))
    exp = np.linalg.norm(diff.astype(np.float64) @ ns, axis=-1)
    assert got.dtype == np.float64 and rel_err(got, exp) < 1e-12
    one = hip.proj_norm(dev(x[:1], tt), dev(u, tt), packed, n).cpu().numpy()
    assert one[0] == got[0]


def test_radix_select_and_percentile_flat(hip):
    """runia_select_hist_f32 / _hip.kth_smallest_flat (round 6): exact order statistics of a flat float32 array - negative values,
    a ReLU layer's share of exact zeros, -0.0, duplicates, +-inf, subnormals - against np.sort; device_fit.percentile_flat equals
    np.percentile bit for bit from 2^22 elements (below it NumPy's own call runs)."""
    from runia_core_amd import config
    from runia_core_amd.device_fit import percentile_flat

    rng = np.random.default_rng(5)
    n = 3_000_001
    a = rng.standard_normal(n).astype(np.float32)
    a[::7] = 0.0
    a[1::1001] = -0.0
    a[5::50_000] = np.inf
    a[6::70_000] = -np.inf
    a[11::9973] = np.float32(1e-42)
    a[12::9973] = a[13]
    srt = np.sort(a)
    ranks = [0, 1, 17, n // 7, n // 2, n // 2 + 1, n - 2, n - 1, int(0.9 * (n - 1))]
    got = hip.kth_smallest_flat(dev(a, torch.float32), ranks)
    for k, g in zip(ranks, got):
        assert np.float32(g) == srt[k] or (np.isinf(g) and g == srt[k]), (k, g, srt[k])
    relu = np.maximum(rng.standard_normal((2100, 2048)).astype(np.float32), 0)  # 4.3 M activations, 50 % zeros
    for q in (90, 65, 50, 10, 99.9, 100, 0):
        want = np.percentile(relu.flatten(), q)
        res = percentile_flat(relu, q)
        assert np.asarray(res).dtype == np.float32 and np.array_equal(res, want), (q, res, want)
    assert config.device_fit is None
    config.device_fit = False
    try:
        assert np.array_equal(percentile_flat(relu, 90), np.percentile(relu.flatten(), 90))  # the host call, same value
    finally:
        config.device_fit = None
    bad = relu.copy()
    bad[3, 3] = np.nan
    assert np.isnan(percentile_flat(bad, 90))  # NumPy's own answer for an array with a NaN


def test_upload_cache_returns_one_tensor_per_host_buffer(hip):
    a = np.random.default_rng(1).standard_normal((600, 512)).astype(np.float32)  # 1.2 MB
    small = np.ones((8, 8), dtype=np.float32)
    t0 = hip.to_device(a, torch.float32)
    assert hip.to_device(a, torch.float32).data_ptr() != t0.data_ptr()  # no cache outside the context
    with hip.upload_cache():
        t1 = hip.to_device(a, torch.float32)
        with hip.upload_cache():  # re-entrant
            assert hip.to_device(a, torch.float32).data_ptr() == t1.data_ptr()
            assert hip.to_device(a[:], torch.float32).data_ptr() == t1.data_ptr()          # another view object of the same buffer
            assert hip.to_device(a, torch.float64).data_ptr() != t1.data_ptr()             # another target dtype: its own entry
            assert hip.to_device(a[1:], torch.float32).data_ptr() != t1.data_ptr()         # another buffer address
        assert hip.to_device(a, torch.float32).data_ptr() == t1.data_ptr()                 # still open
        assert hip.to_device(small, torch.float32).data_ptr() != hip.to_device(small, torch.float32).data_ptr()  # below 1 MB
        assert torch.equal(t1, t0)
    t2 = hip.to_device(a, torch.float32)
    assert t2.data_ptr() != t1.data_ptr() or True  # (the allocator may hand the block out again; the cache is gone:)
    from runia_core_amd import _hip as H

    assert H._upload_cache.depth == 0 and H._upload_cache.entries is None
