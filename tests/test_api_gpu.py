"""GPU parity of the drop-in classes (the reference's own test cases re-run through
runia_core_amd) against the reference's golden numbers, the by-path fixtures and the oracle."""
import warnings

import numpy as np
import pytest
import torch

import oracle
import runia_core_amd as rc
from conftest import KDE_HD_MEASURED, generate_test_data, kde_hd_inputs, load_npz, rel_err
from runia_core_amd.inference import (
    KNN,
    MSP,
    Energy,
    KDELatentSpace,
    KNNLatentSpace,
    LaREMPipeline,
    Mahalanobis,
    MDLatentSpace,
    postprocessors_dict,
)
from test_oracle_goldens import _all_baselines_inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(params=["host_fit", "device_fit"])
def fit_on(request, monkeypatch):
    """Every golden / reference-run-fixture test with a setup() fit runs twice: with the reference's own host calls
    (sklearn / SciPy / NumPy: bit-identical fitted state) and with ``runia_core_amd.config.device_fit`` (covariance on the
    f64 matrix cores, pinvh and the PCA solvers on the hand-written Jacobi eigen-solver).  The contract (1e-5) holds for
    both; where a test asks for more than the contract (1e-8 ...) that is the host fit's bar."""
    from runia_core_amd import config

    monkeypatch.setattr(config, "device_fit", request.param == "device_fit")
    return request.param


def _tight(fit_on, host_tol):
    return host_tol if fit_on == "host_fit" else TOL


def _list(ref_vectors, key, i=0):
    return np.array(ref_vectors[key]["lists"][i]["values"])


def test_md_unit_golden(ref_vectors, fit_on):
    # /root/reference/tests/unit_test_postprocessors.py:205-233
    tr, _, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    md = MDLatentSpace()
    md.setup(tr)
    s = md.postprocess(te)
    assert isinstance(s, np.ndarray) and s.dtype == np.float64 and len(s) == 10 and np.all(np.isfinite(s))
    exp = _list(ref_vectors, "md_unit")
    assert abs((exp - s).sum()) < _tight(fit_on, 1e-6) * 10
    assert rel_err(s, exp) < _tight(fit_on, 1e-8)


def test_larem_lared_baselines_goldens(ref_vectors, fit_on):
    # /root/reference/tests/unit_test_baselines.py:463-568
    np.random.seed(1)
    f = np.random.rand(200, 20)
    md = MDLatentSpace()
    md.setup(f)
    assert np.allclose(md.precision[0], _list(ref_vectors, "larem_baselines", 0), atol=1e-6)
    s = md.postprocess(f)
    assert s.shape == (200,) and np.allclose(s[:20], _list(ref_vectors, "larem_baselines", 1), atol=1e-6)
    kde = KDELatentSpace()
    kde.setup(f)
    s = kde.postprocess(f)
    assert s.shape == (200,) and np.allclose(s[:20], _list(ref_vectors, "lared_baselines"), atol=1e-6)


def test_kde_unit_golden(ref_vectors):
    tr, _, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    kde = KDELatentSpace()
    kde.setup(tr)
    s = kde.postprocess(te)
    assert abs((_list(ref_vectors, "kde_unit") - s).sum()) < 1e-6


def test_knn_latent_k_gt_bank_golden(ref_vectors):
    tr, _, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    knn = KNNLatentSpace()
    knn.setup(tr)
    s = knn.postprocess(te)
    assert s.dtype == np.float32 and np.all(np.isfinite(s))
    assert abs((_list(ref_vectors, "knn_latent_unit") - s).sum()) < 1e-6


def test_energy_msp_goldens(ref_vectors):
    _, _, trl = generate_test_data(seed=42)
    _, _, tel = generate_test_data(seed=43)
    e = Energy(flip_sign=True)
    e.setup(trl)
    assert e._setup_flag and e.threshold is not None
    s = e.postprocess(tel)
    assert s.dtype == np.float32
    assert abs((_list(ref_vectors, "energy_unit") - s).sum()) < 1e-6
    st = e.postprocess(torch.Tensor(tel))  # tensor input
    assert np.array_equal(s, st)
    g = load_npz("ref_energy_msp.npz")
    assert abs(e.threshold - float(g["unit_energy_threshold"])) < 1e-5
    m = MSP(flip_sign=True)
    m.setup(trl)
    assert rel_err(m.postprocess(tel), g["unit_msp_scores"]) < TOL
    assert abs(m.threshold - float(g["unit_msp_threshold"])) < 1e-5


def test_mahalanobis_goldens(ref_vectors, fit_on):
    tr, lab, _ = generate_test_data(seed=42)
    va, _, _ = generate_test_data(seed=44)
    te, _, _ = generate_test_data(seed=43)
    m = Mahalanobis(flip_sign=True, num_classes=10)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.setup(tr, train_labels=lab, valid_feats=va)
    assert m._setup_flag and m.class_mean is not None and m.precision is not None and m.threshold is not None
    s = m.postprocess(te)
    assert s.dtype == np.float64
    assert abs((_list(ref_vectors, "mahalanobis_unit") - s).sum()) < 1e-6
    assert np.allclose(s, m.postprocess(torch.Tensor(te)))
    g = load_npz("ref_mahalanobis.npz")
    assert rel_err(s, g["unit_scores"]) < _tight(fit_on, 1e-8)
    assert abs(m.threshold - float(g["unit_threshold"])) < _tight(fit_on, 1e-6) * max(1.0, abs(float(g["unit_threshold"])))
    m = Mahalanobis(flip_sign=False, num_classes=7)
    m.setup(g["d96_train"], train_labels=g["d96_labels"], valid_feats=g["d96_train"][:100])
    assert rel_err(m.postprocess(g["d96_test"]), g["d96_scores"]) < _tight(fit_on, 1e-9)
    assert abs(m.threshold - float(g["d96_threshold"])) < _tight(fit_on, 1e-8) * max(1.0, abs(float(g["d96_threshold"])))


def test_all_baselines_means(ref_vectors, fit_on):
    # /root/reference/tests/unit_test_baselines.py:199-268 (msp, knn, energy, mdist)
    d = _all_baselines_inputs()
    sc = [s["value"] for s in ref_vectors["all_baselines_means"]["scalars"]]
    p = MSP(flip_sign=False)
    p.setup(ind_train_data=d["tr_l"])
    assert abs(p.postprocess(test_data=d["ood_l"]).mean() - sc[0]) < 1e-6
    p = KNN(flip_sign=False, k_neighbors=10)
    p.setup(ind_train_data=d["tr_f"], valid_feats=d["va_f"])
    assert abs(p.postprocess(test_data=d["ood_f"]).mean() - sc[1]) < 1e-6
    p = Energy(flip_sign=False)
    p.setup(ind_train_data=d["tr_l"])
    assert abs(p.postprocess(test_data=d["ood_l"]).mean() - sc[2]) < 1e-6
    p = Mahalanobis(flip_sign=False, num_classes=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p.setup(ind_train_data=d["tr_f"], train_labels=np.argmax(d["tr_l"], axis=-1), valid_feats=d["va_f"])
        assert abs(p.postprocess(test_data=d["ood_f"]).mean() - sc[8]) < 1e-6


def test_entropy_api_goldens(ref_vectors):
    # /root/reference/tests/unit_test_feature_extraction.py:175-247
    np.random.seed(1)
    sample = np.random.rand(3, 20)
    h = rc.single_image_entropy_calculation(sample, 2)
    assert h.shape == (20,)
    # the golden is defined on f64 draws; the device path (like the reference's own
    # tensor path) carries MC samples in f32 -> compare at the f32-input level
    assert np.allclose(h, _list(ref_vectors, "entropy_single_image"), atol=1e-5)
    assert np.abs(h - oracle.single_image_entropy_calculation(sample.astype(np.float32), 2)).max() < 1e-12
    torch.manual_seed(1)
    z = torch.rand(3 * 200, 20)
    h_mvn, h_z = rc.get_dl_h_z(z, 3, parallel_run=True)
    assert h_z.shape == (200, 20) and h_mvn.shape == (200, 1) and h_z.dtype == np.float64
    assert np.allclose(h_z[0], _list(ref_vectors, "entropy_get_dl_h_z"), atol=1e-6)
    o_mvn, o_z = oracle.get_dl_h_z(z.numpy(), 3)
    assert np.abs(h_z - o_z).max() < 1e-12 and np.abs(h_mvn - o_mvn).max() < 1e-10
    h_mvn2, h_z2 = rc.get_dl_h_z(z.numpy(), 3)
    assert np.array_equal(h_z, h_z2) and np.array_equal(h_mvn, h_mvn2)
    with pytest.raises(ValueError):
        rc.get_dl_h_z(np.zeros((7, 4), dtype=np.float32), 3)


def test_pca_api_goldens(ref_vectors, fit_on):
    # /root/reference/tests/unit_test_dim_reduction.py:24-107
    np.random.seed(1)
    ind = 0.5 + np.random.randn(1000, 20)
    ood = -0.5 + np.random.randn(1000, 20)
    tr, pca = rc.apply_pca_ds_split(ind, 10)
    assert abs((tr[0] - _list(ref_vectors, "pca_ds_split", 0)).sum()) < 1e-7
    assert abs((pca.components_[0] + _list(ref_vectors, "pca_ds_split", 1)).sum()) < 1e-7
    y = rc.apply_pca_transform(ood, pca)
    assert y.shape == (1000, 10) and y.dtype == np.float64
    assert abs((y[0] - _list(ref_vectors, "pca_transform")).sum()) < 1e-7
    assert rel_err(y, pca.transform(ood)) < 1e-12
    # fit_transform rows == transform of the same rows (sklearn identity) through the kernel
    assert rel_err(rc.apply_pca_transform(ind, pca), tr) < 1e-9


def test_metrics_postprocessors_goldens(ref_vectors, fit_on):
    # /root/reference/tests/unit_test_metrics.py:31-80 through the harness calling convention
    np.random.seed(1)
    valid = 0.5 + np.random.randn(1000, 20)
    train = 0.5 + np.random.randn(1000, 20)
    vl = np.random.randint(5, size=1000)
    tl = np.random.randint(5, size=1000)
    np.random.randint(5, size=1000)
    ood = -0.5 + np.random.randn(1000, 20)
    gold = [s["value"] for s in ref_vectors["metrics_postprocessors"]["scalars"]]
    for name, (auroc, aupr, fpr) in (("KDE", gold[:3]), ("MD", gold[3:])):
        p = postprocessors_dict[name](cfg=None)
        p._setup_flag = False
        p.setup(train, ind_train_labels=tl)
        ind_s = p.postprocess(valid, pred_labels=vl)
        ood_s = p.postprocess(ood, pred_labels=vl)
        r = rc.evaluation.get_auroc_results(f"test {name}", ind_s, ood_s)
        assert abs(r["auroc"].values[0] - auroc) < 1e-7
        assert abs(r["aupr"].values[0] - aupr) < 1e-7
        assert abs(r["fpr@95"].values[0] - fpr) < 1e-7


class _ToyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(1, 20, 5)
        self.fc = torch.nn.Linear(20, 10)

    def forward(self, x):
        f = torch.relu(torch.nn.functional.max_pool2d(self.conv(x), 3))
        return self.fc(f.mean(dim=(2, 3)))


def test_larex_inference_end_to_end(fit_on):
    """LaRExInference.get_score (image_level.py:96-120 of the reference) on a toy backbone,
    against the oracle fed with the same hooked activation and the same CPU-generator draws."""
    torch.manual_seed(1)
    np.random.seed(1)
    model = _ToyNet().eval()
    hook = rc.Hook(model.conv)
    feats = np.random.rand(200, 20)
    red, pca = rc.apply_pca_ds_split(feats, 8)
    md = MDLatentSpace()
    md.setup(red)
    inf = rc.LaRExInference(model=model, postprocessor=md, mcd_sampler=rc.MCSamplerModule, pca_transform=pca,
                            mcd_samples_nro=16, drop_block_prob=0.5, drop_block_size=4, layer_type="Conv")
    assert inf.mc_sampler.training and len(inf.mc_sampler.drop_blocks) == 16 and inf.device.type == "cuda"
    img = torch.randn(1, 1, 28, 28)
    torch.manual_seed(123)
    out, score = inf.get_score(img, layer_hook=hook)
    assert out.shape == (1, 10) and isinstance(score, np.ndarray) and score.shape == (1,) and score.dtype == np.float64
    # oracle on the same activation and the same random stream
    latent = hook.output.detach().cpu().numpy()
    torch.manual_seed(123)
    rand = torch.cat([torch.rand(1, latent.shape[2], latent.shape[3]) for _ in range(16)]).numpy()
    z = oracle.mc_stack(latent, rand, 0.5, 4)
    z_dev = inf.mc_sampler(hook.output, rand=torch.from_numpy(rand).cuda()[None]).cpu().numpy()
    assert np.allclose(z_dev, z, rtol=3e-6, atol=1e-7)
    # downstream stages from the device's own samples: f64-exact chain
    _, h = oracle.get_dl_h_z(z_dev, 16)
    y = oracle.pca_transform(h, pca.components_, pca.mean_, pca.explained_variance_)
    exp = oracle.md_score(y, md.feats_mean, md.precision)
    assert rel_err(score, exp) < 1e-9
    # batched additive API == per-image API
    torch.manual_seed(123)
    s2 = inf.get_scores_from_latents(hook.output)
    assert np.array_equal(s2, score)
    # LaRD path (no MC, no entropy)
    md2 = MDLatentSpace()
    md2.setup(np.random.rand(100, 20))
    lard = rc.LaRDInference(model, md2, pca_transform=None, layer_type="Conv")
    out, s = lard.get_score(img, hook)
    red_rows = latent.mean(axis=3, dtype=np.float32).mean(axis=2, dtype=np.float32)
    assert rel_err(s, oracle.md_score(red_rows.astype(np.float64), md2.feats_mean, md2.precision)) < 1e-5


def test_pipeline_full_size_properties():
    """cfg2 at full size (N=10 000 x 16 x 512 -> PCA-256 -> LaREM): size-independent properties."""
    torch.manual_seed(0)
    n, n_mc, d, k = 10000, 16, 512, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randn(n, 1, d, device="cuda", generator=g) + 2
    z = (base * (1 + 0.1 * torch.randn(n, n_mc, d, device="cuda", generator=g))).reshape(n * n_mc, d).contiguous()
    gp = load_npz("ref_pca.npz")
    gm = load_npz("ref_md.npz")
    from runia_core_amd.dimensionality_reduction import DevicePCA

    md = MDLatentSpace()
    md.feats_mean, md.precision, md._setup_flag = gm["d256_mean"], gm["d256_precision"], True
    pipe = LaREMPipeline(md, DevicePCA(gp["d512_components"], gp["d512_mean"], gp["d512_var"], True), n_mc)
    s = pipe.score_samples(z)
    assert s.shape == (n,) and bool(torch.isfinite(s).all()) and bool((s <= 0).all())
    # (1) permuting the MC samples of an image does not change its score (order statistics)
    perm = torch.randperm(n_mc, device="cuda")
    zp = z.reshape(n, n_mc, d)[:, perm, :].reshape(n * n_mc, d).contiguous()
    assert torch.equal(pipe.score_samples(zp), s)
    # (2) rows are independent: scoring a slice equals the slice of the scores (sharding property)
    assert torch.equal(pipe.score_samples(z[3000 * n_mc : 5000 * n_mc]), s[3000:5000])
    # (3) a bounded sample against the oracle
    idx = slice(0, 64 * n_mc)
    exp, _ = oracle.larem_pipeline(z[idx].cpu().numpy(), n_mc, gp["d512_components"], gp["d512_mean"], gp["d512_var"],
                                   gm["d256_mean"], gm["d256_precision"])
    assert rel_err(s[:64].cpu().numpy(), exp) < 1e-9
    # (4) translating all samples of an image by a constant leaves the entropies unchanged up to f32 input rounding;
    #     scaling by 2 adds exactly log(2) to every per-dimension entropy
    h = pipe.entropy(z[: 32 * n_mc])
    h2 = pipe.entropy((z[: 32 * n_mc] * 2).contiguous())
    assert float((h2 - h - np.log(2.0)).abs().max()) < 1e-12


def test_device_fit_matches_host_fit():
    """covariance on the f64 matrix cores + eigh/pinvh on the GPU vs the reference's host calls."""
    from runia_core_amd import _hip, config
    from runia_core_amd.device_fit import empirical_precision_device

    rng = np.random.default_rng(3)
    for n, d, dt in ((1000, 20, np.float64), (5000, 256, np.float64), (777, 96, np.float32), (300, 70, np.float64)):
        x = (rng.standard_normal((n, d)) * (0.5 + rng.random(d)) + rng.standard_normal(d)).astype(dt)
        mean, cov = _hip.covariance(torch.from_numpy(x).cuda())
        assert rel_err(mean.cpu().numpy(), x.astype(np.float64).mean(0)) < 1e-13
        assert rel_err(cov.cpu().numpy(), np.cov(x.astype(np.float64).T, bias=1)) < 1e-12
        assert rel_err(empirical_precision_device(x), oracle.empirical_precision(x)) < 1e-8
    # rank-deficient case of the reference's own unit test (10 samples x 32 dims): same pinvh cut-off
    tr, _, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    g = load_npz("ref_md.npz")
    before = config.device_fit
    config.device_fit = True
    try:
        md = MDLatentSpace()
        md.setup(tr)
        assert rel_err(md.precision, g["unit_precision"]) < 1e-8
        assert rel_err(md.postprocess(te), g["unit_scores"]) < 1e-8
        gm = load_npz("ref_mahalanobis.npz")
        m = Mahalanobis(flip_sign=False, num_classes=7)
        m.setup(gm["d96_train"], train_labels=gm["d96_labels"], valid_feats=gm["d96_train"][:100])
        assert rel_err(m.precision, gm["d96_precision"]) < 1e-8
        assert rel_err(m.postprocess(gm["d96_test"]), gm["d96_scores"]) < 1e-8
    finally:
        config.device_fit = before


def test_f4_kernels_and_classes(ref_vectors, fit_on):
    """ASH / ReAct / DICE / DICE+ReAct / GEN through the registry against the reference fixtures,
    the reference's all-baselines goldens and the oracle."""
    from runia_core_amd import _hip
    from runia_core_amd.inference import ASH, DICE, GEN, DICEReAct, ReAct
    from test_oracle_goldens import _fc_params, logsumexp_rows

    g = load_npz("ref_f4.npz")
    w, b, tr, va, te = g["w"], g["b"], g["train"], g["valid"], g["test"]
    fc = {"weight": w, "bias": b}
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    # kernels
    for pct in (90, 65, 100, 0):
        got = _hip.ash_s(dv(te), pct).cpu().numpy()
        exp = oracle.ash_s_defined(te.copy(), pct)
        assert np.array_equal(got != 0, exp != 0) and np.allclose(got, exp, rtol=2e-6, atol=0), pct
        if f"ash{pct}_transformed" in g:  # the reference itself: same kept positions; equal on its self-consistent rows
            ref = g[f"ash{pct}_transformed"]
            ok = np.all(np.isclose(ref, exp, rtol=1e-6), axis=1)
            assert np.array_equal(got != 0, ref != 0) and np.allclose(got[ok], ref[ok], rtol=2e-6)
    logits = _hip.linear(dv(te), dv(w), dv(b)).cpu().numpy()
    assert rel_err(logits, te @ w.T + b) < 1e-5
    assert rel_err(_hip.linear(dv(te), dv(w), dv(b), 0.7).cpu().numpy(), te.clip(max=np.float32(0.7)) @ w.T + b) < 1e-5
    for M in (37, 10, 100, 1):
        exp = g[f"gen{M}_scores"] if f"gen{M}_scores" in g else oracle.gen_score(g["logits_test"], 0.1, M)
        assert rel_err(_hip.gen_score(dv(g["logits_test"]), 0.1, M).cpu().numpy(), exp) < 1e-5, M
    big = (np.random.default_rng(2).standard_normal((300, 1000)) * 2).astype(np.float32)
    for M in (1000, 100):
        assert rel_err(_hip.gen_score(dv(big), 0.1, M).cpu().numpy(), oracle.gen_score(big, 0.1, M)) < 1e-5
    # classes vs the by-path reference fixtures
    for pct in (90, 65):
        p = ASH(flip_sign=False, ash_percentile=pct)
        p.setup(tr, valid_feats=va, final_linear_layer_params=fc)
        s_dev = p.postprocess(te)
        assert rel_err(s_dev, oracle.linear_energy(oracle.ash_s_defined(te.copy(), pct), w, b)) < 1e-5
        ok = np.all(np.isclose(g[f"ash{pct}_transformed"], oracle.ash_s_defined(te.copy(), pct), rtol=1e-6), axis=1)
        assert rel_err(s_dev[ok], g[f"ash{pct}_scores"][ok]) < 1e-5  # rows where the reference's scatter is consistent
    p = ReAct(flip_sign=False, react_percentile=90)
    p.setup(tr, valid_feats=va, final_linear_layer_params=fc)
    assert p.activation_threshold == float(g["react_clip"])
    assert rel_err(p.postprocess(te), g["react_scores"]) < 1e-5
    assert np.array_equal(p.postprocess(torch.Tensor(te)), p.postprocess(te))
    for M in (37, 10):
        p = GEN(flip_sign=False, gamma=0.1, num_classes=M)
        p.setup(g["logits_train"])
        assert rel_err(p.postprocess(g["logits_test"]), g[f"gen{M}_scores"]) < 1e-5
        assert abs(p.threshold - float(g[f"gen{M}_threshold"])) < 1e-4
    p = DICE(flip_sign=False, dice_percentile=90, num_classes=37)
    p.setup(tr, valid_feats=va, final_linear_layer_params=fc)
    mw = oracle.dice_masked_weight(tr, w, 90)
    assert np.array_equal(p.dice_layer.masked_w, mw)
    assert rel_err(p.postprocess(te), logsumexp_rows(oracle.dice_logits(te, mw, b))) < 1e-5
    # the reference's own goldens (tests/unit_test_baselines.py:255-268) through the harness calling convention
    d = _all_baselines_inputs()
    wf, bf = _fc_params()
    fcp = {"weight": wf, "bias": bf}
    sc = [s["value"] for s in ref_vectors["all_baselines_means"]["scalars"]]
    kw = dict(ind_train_data=d["tr_f"], valid_feats=d["va_f"], final_linear_layer_params=fcp)
    for cls, args, gold, tol in ((ASH, dict(ash_percentile=90), sc[3], 1e-3), (ReAct, dict(react_percentile=90), sc[5], 1e-5),
                                 (DICE, dict(dice_percentile=90, num_classes=20), sc[6], 1e-5),
                                 (DICEReAct, dict(dice_percentile=90, react_percentile=90, num_classes=20), sc[7], 1e-5)):
        p = cls(flip_sign=False, **args)
        p.setup(**kw)
        assert abs(p.postprocess(test_data=d["ood_f"]).mean() - gold) < tol, cls.__name__
    p = GEN(flip_sign=False, gamma=0.1, num_classes=20)
    p.setup(ind_train_data=d["tr_l"])
    assert abs(p.postprocess(test_data=d["ood_l"]).mean() - sc[4]) < 1e-5
    with pytest.raises(AssertionError, match="final_linear_layer_params must be provided for ReAct"):
        ReAct(flip_sign=False).setup(tr, valid_feats=va)
    with pytest.raises(AssertionError, match="valid_feats must be provided for ASH"):
        ASH(flip_sign=False).setup(tr, final_linear_layer_params=fc)
    with pytest.raises(AssertionError, match=r"setup\(\) must be called"):
        GEN(flip_sign=False, gamma=0.1, num_classes=5).postprocess(g["logits_test"])


def test_cmd_unit_golden(ref_vectors, fit_on):
    # /root/reference/tests/unit_test_postprocessors.py:236-317
    from runia_core_amd.inference import cMDLatentSpace

    tr, lab, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    np.random.seed(42)
    pred = np.random.randint(0, 10, len(te))
    p = cMDLatentSpace()
    assert p.num_classes == 10 and p.class_mean is None and not p._setup_flag

    class Cfg:
        num_classes = 5

    assert cMDLatentSpace(Cfg()).num_classes == 5
    with pytest.raises(ValueError, match="id_labels not provided"):
        p.setup(tr)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p.setup(tr, ind_train_labels=lab)
    assert p._setup_flag and p.class_mean.shape == (10, 32)
    with pytest.raises(ValueError, match="pred_logits not provided"):
        p.postprocess(te)
    s = p.postprocess(te, pred_labels=pred)
    assert s.dtype == np.float32 and np.all(np.isfinite(s))
    exp = _list(ref_vectors, "cmd_unit")
    assert abs((exp - s).sum()) < 1e-5  # float32 arithmetic in the reference (its own test allows 1e-6 on the signed sum)
    assert rel_err(s, exp) < 1e-5


def test_vim_fixture_and_contract(fit_on):
    from runia_core_amd.inference import ViM

    g = load_npz("ref_f4.npz")
    fc = {"weight": g["w"], "bias": g["b"]}
    p = ViM(flip_sign=False)
    assert p.u is None and p.NS is None and p.alpha is None
    with pytest.raises(AssertionError, match="train_logits must be provided for ViM"):
        p.setup(g["train"], final_linear_layer_params=fc, valid_feats=g["valid"], valid_logits=g["logits_valid"])
    p.setup(g["train"], final_linear_layer_params=fc, train_logits=g["logits_train"], valid_feats=g["valid"],
            valid_logits=g["logits_valid"])
    assert p.DIM == 150 and p.NS.shape == (300, 150)
    # the reference runs ViM in float32 end to end for float32 features (covariance, eigenvectors, norms, alpha);
    # the kernel projects in f64, so agreement is at the f32 level
    assert abs(p.alpha - float(g["vim_alpha"])) < 1e-6 * abs(p.alpha)
    s = p.postprocess(g["test"], logits=g["logits_test"])
    assert rel_err(s, g["vim_scores"]) < 1e-5
    assert abs(p.threshold - float(g["vim_threshold"])) < 1e-4
    assert np.allclose(p.postprocess(torch.Tensor(g["test"]), logits=torch.Tensor(g["logits_test"])), s)
    # the residual-norm kernel alone, against the fitted state of the reference
    from runia_core_amd import _hip

    dv = lambda a, t: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", t)  # noqa: E731
    nrm = _hip.proj_norm(dv(g["test"], torch.float32), dv(g["vim_u"], torch.float32),
                         _hip.pack_weights(dv(g["vim_NS"], torch.float64)), 150).cpu().numpy()
    exp = np.linalg.norm(np.matmul((g["test"] - g["vim_u"]).astype(np.float64), g["vim_NS"].astype(np.float64)), axis=-1)
    assert rel_err(nrm, exp) < 1e-12


def test_vim_device_fit_opt_in():
    """config.vim_device_fit: the residual space from the exact second moment + the Jacobi solver.  Orthonormal, real, in the
    features' dtype; scores within 5e-5 of the reference-run fixture (the reference's float32 eig sits 1.5e-5 from the exact
    null space: INTEGRATION.md 'Known divergences'); the default stays the reference's host call."""
    from runia_core_amd import config
    from runia_core_amd.inference import ViM

    g = load_npz("ref_f4.npz")
    fc = {"weight": g["w"], "bias": g["b"]}
    kw = dict(final_linear_layer_params=fc, train_logits=g["logits_train"], valid_feats=g["valid"], valid_logits=g["logits_valid"])
    assert config.vim_device_fit is False
    config.vim_device_fit = True
    try:
        p = ViM(flip_sign=False)
        p.setup(g["train"], **kw)
        s = p.postprocess(g["test"], logits=g["logits_test"])
    finally:
        config.vim_device_fit = False
    assert p.NS.shape == (300, 150) and p.NS.dtype == np.float32 and not np.iscomplexobj(p.NS)
    gram = p.NS.astype(np.float64).T @ p.NS.astype(np.float64)
    assert np.max(np.abs(gram - np.eye(150))) < 1e-6
    assert s.dtype == g["vim_scores"].dtype and rel_err(s, g["vim_scores"]) < 5e-5
    assert abs(p.alpha - float(g["vim_alpha"])) < 5e-5 * abs(p.alpha)
    # the same subspace as the exact decomposition of the same moment (NumPy f64 eigh): projector difference at round-off
    x = g["train"].astype(np.float64) - np.asarray(p.u, dtype=np.float64)
    w_, v_ = np.linalg.eigh(x.T @ x / x.shape[0])
    ns = v_[:, np.argsort(-w_)[150:]]
    y = g["test"].astype(np.float64) - np.asarray(p.u, dtype=np.float64)
    a, b = np.linalg.norm(y @ ns, axis=-1), np.linalg.norm(y @ p.NS.astype(np.float64), axis=-1)
    assert np.max(np.abs(a - b) / a) < 2e-6  # (the basis is rounded to float32)
    h = ViM(flip_sign=False)
    h.setup(g["train"], **kw)
    assert rel_err(h.postprocess(g["test"], logits=g["logits_test"]), g["vim_scores"]) < 1e-5


def test_gmm_and_ddu_fixtures(fit_on):
    """GMM (LaREG) and DDU against the reference run by path on well-conditioned data (float32 arithmetic there)."""
    from runia_core_amd.inference import DDU, GMMLatentSpace, gmm_fit

    g = load_npz("ref_f4.npz")
    tr, lab, te = g["gmm_train"], g["gmm_labels"], g["gmm_test"]
    p = GMMLatentSpace()
    assert p.num_classes == 10 and p.gmm is None
    with pytest.raises(ValueError, match="id_labels not provided"):
        p.setup(tr)
    p.setup(tr, ind_train_labels=lab)
    assert hasattr(p.gmm, "loc") and list(p.gmm.loc.shape) == [6, 24]  # empty classes 6..9 dropped
    assert np.allclose(p.gmm.loc.numpy(), g["gmm_loc"], atol=1e-6) and np.allclose(p.gmm.scale_tril.numpy(), g["gmm_tril"], atol=1e-5)
    s = p.postprocess(te)
    assert s.dtype == np.float32 and rel_err(s, g["gmm_scores"]) < 1e-5
    assert rel_err(s, oracle.gmm_energy(p.gmm, te)) < 1e-5
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p.setup(tr, ind_train_labels=lab)
        assert len(w) == 1 and "already trained" in str(w[0].message)
    d = DDU(flip_sign=False, num_classes=6)
    with pytest.raises(AssertionError, match="train_labels must be provided for DDU"):
        d.setup(tr, valid_feats=tr[:10])
    d.setup(tr, valid_feats=tr[:200], train_labels=lab)
    assert rel_err(d.postprocess(te), g["ddu_scores"]) < 1e-5
    assert abs(d.threshold - float(g["ddu_threshold"])) < 1e-3
    gmm, jitter = gmm_fit(torch.tensor([[0.0, 1.0, 2.0], [0.1, 1.1, 2.1], [5.0, 6.0, 7.0], [5.1, 6.1, 7.1]]),
                          torch.tensor([0, 0, 1, 1]), num_classes=2)
    assert list(gmm.loc.shape) == [2, 3] and isinstance(jitter, (int, float))  # reference tests/unit_test_baselines.py:193-203


def test_latent_methods_harness_goldens(fit_on, ref_vectors=None):
    """The reference's harness-level test (tests/unit_test_latent_methods.py:36-115: log_evaluate_larex with a PCA sweep
    over [1, 2, 4] components, postprocessors KNN / MD / GMM, best AUROC per postprocessor) replayed on the drop-in
    classes with the harness calling convention (evaluation/metrics.py:322-340, evaluation/latent_space.py:135-170)."""
    torch.manual_seed(1)
    np.random.seed(1)
    np.random.rand(20, 20)
    np.random.rand(20)
    r = lambda m: np.float32(m + np.random.randn(200, 20))  # noqa: E731
    tr_f, tr_l, tr_z, va_f, va_l, va_z = r(0.5), r(0.5), r(0.4), r(0.5), r(0.5), r(0.4)
    ood_f, ood_l, ood_z = r(-0.5), r(-0.5), r(-0.4)
    train_labels, valid_labels, ood_labels = (np.argmax(a, axis=-1) for a in (tr_l, va_l, ood_l))

    class Cfg:
        k_neighbors = 10

    def auroc(name, train, valid, ood):
        p = postprocessors_dict[name](cfg=Cfg())
        p._setup_flag = False
        p.setup(train, ind_train_labels=train_labels)
        ind_s = p.postprocess(valid, pred_labels=valid_labels)
        ood_s = p.postprocess(ood, pred_labels=ood_labels)
        return float(rc.evaluation.get_auroc_results(name, ind_s, ood_s)["auroc"].values[0])

    names = ["KNN", "MD", "GMM"]
    best = {n: auroc(n, tr_z, va_z, ood_z) for n in names}
    for n_comp in (1, 2, 4):  # PCA fits consume the global NumPy RNG in this order, as in the reference harness
        tr_p, pca = rc.apply_pca_ds_split(tr_z, n_comp)
        va_p, ood_p = rc.apply_pca_transform(va_z, pca), rc.apply_pca_transform(ood_z, pca)
        for n in names:
            best[n] = max(best[n], auroc(n, tr_p, va_p, ood_p))
    assert abs(best["KNN"] - 0.9881750345230103) < 1e-6
    assert abs(best["MD"] - 0.837399959564209) < 1e-6
    assert abs(best["GMM"] - 0.801800012588501) < 1e-6


@pytest.mark.parametrize("device_resident", [False, True])
def test_log_evaluate_larex_reference_goldens(device_resident):
    """runia_core_amd.evaluation.log_evaluate_larex (round 5: the reference's harness loop, evaluation/latent_space.py:30-221, with
    its signature) on the inputs of the reference's own test (tests/unit_test_latent_methods.py:36-115): the best-AUROC goldens of
    KNN / MD / GMM, the row names of the results table, the best-configuration names and the thresholds dict - in the host-array
    form and in the additive device-resident form (same table values)."""
    from runia_core_amd.evaluation import log_evaluate_larex

    torch.manual_seed(1)
    np.random.seed(1)
    np.random.rand(20, 20)
    np.random.rand(20)
    r = lambda m: np.float32(m + np.random.randn(200, 20))  # noqa: E731
    tr_f, tr_l, tr_z, va_f, va_l, va_z = r(0.5), r(0.5), r(0.4), r(0.5), r(0.5), r(0.4)
    ood_f, ood_l, ood_z = r(-0.5), r(-0.5), r(-0.4)
    ind = {"train latent_space_means": tr_z, "valid latent_space_means": va_z, "train labels": np.argmax(tr_l, axis=-1),
           "valid labels": np.argmax(va_l, axis=-1), "msp": np.max(va_l, axis=1)}
    ood = {"test_ood latent_space_means": ood_z, "test_ood labels": np.argmax(ood_l, axis=-1)}
    cfg = {"ood_datasets": ["test_ood"], "n_pca_components": [1, 2, 4], "k_neighbors": 10, "ind_dataset": "test_id"}

    class Cfg:
        pass

    c = Cfg()
    for k, v in cfg.items():
        setattr(c, k, v)
    df, best, thresholds, ood_out = log_evaluate_larex(c, ["msp"], {"test_ood msp": np.max(ood_l, axis=1)}, ind, ood, "my_run", False,
                                                       postprocessors=["KNN", "MD", "GMM"], device_resident=device_resident)
    assert abs(best["KNN"]["auroc"] - 0.9881750345230103) < 1e-6
    assert abs(best["MD"]["auroc"] - 0.837399959564209) < 1e-6
    assert abs(best["GMM"]["auroc"] - 0.801800012588501) < 1e-6
    want_rows = ["test_ood msp"] + [f"test_ood {p}{ext}" for ext in ("", " PCA 1", " PCA 2", " PCA 4") for p in ("KNN", "MD", "GMM")]
    assert list(df.index) == want_rows and list(df.columns) == ["auroc", "fpr@95", "aupr", "fpr", "tpr"]
    assert set(best) == {"best", "KNN", "MD", "GMM"} and len(best["best"]) == 3
    assert set(thresholds) == {best[p]["best_comp"] for p in ("KNN", "MD", "GMM")}
    for p in ("KNN", "MD", "GMM"):
        assert f"test_ood {best[p]['best_comp']}" in ood_out and ood_out[f"test_ood {best[p]['best_comp']}"].shape == (200,)
    if not device_resident:
        assert isinstance(df.loc["test_ood MD", "fpr"], list) and df.loc["test_ood MD", "tpr"][-1] == 1.0
    with pytest.raises(NotImplementedError, match="mlflow"):
        log_evaluate_larex(c, [], {}, ind, ood, "run", True)


def test_folded_single_contraction_equals_two_stage():
    """LaREMPipeline's folded weights (score = -||M h + c||^2) against the two-stage kernel and the oracle,
    incl. a rank-deficient precision matrix (pinvh dropped directions) and the no-PCA case."""
    from runia_core_amd.dimensionality_reduction import DevicePCA

    rng = np.random.default_rng(12)
    gp, gm = load_npz("ref_pca.npz"), load_npz("ref_md.npz")
    h = rng.standard_normal((700, 512)) * 0.7 + 0.3
    hd = torch.from_numpy(h).cuda()
    md = MDLatentSpace()
    md.feats_mean, md.precision, md._setup_flag = gm["d256_mean"], gm["d256_precision"], True
    pca = DevicePCA(gp["d512_components"], gp["d512_mean"], gp["d512_var"], True)
    pipe = LaREMPipeline(md, pca, 16)
    assert pipe.fold_weights
    s_fold = pipe.score_entropies(hd).cpu().numpy()
    pipe2 = LaREMPipeline(md, pca, 16)
    pipe2.fold_weights = False
    s_two = pipe2.score_entropies(hd).cpu().numpy()
    y = oracle.pca_transform(h, gp["d512_components"], gp["d512_mean"], gp["d512_var"])
    exp = oracle.md_score(y, gm["d256_mean"], gm["d256_precision"])
    assert rel_err(s_two, exp) < 1e-11 and rel_err(s_fold, exp) < 1e-10
    # rank-deficient precision (the reference's unit case: 10 samples x 32 dims), no PCA
    tr, _, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    md2 = MDLatentSpace()
    md2.setup(tr.astype(np.float64))
    p3 = LaREMPipeline(md2, None, 16)
    s3 = p3.score_entropies(torch.from_numpy(te.astype(np.float64)).cuda()).cpu().numpy()
    assert p3._folded_state() is not None and p3._folded_state()[2] == 9  # rank 9 = samples - 1
    assert rel_err(s3, oracle.md_score(te.astype(np.float64), md2.feats_mean, md2.precision)) < 1e-9
    # an indefinite "precision" cannot be factored: the pipeline keeps the two-stage kernel
    md3 = MDLatentSpace()
    a = rng.standard_normal((8, 8))
    md3.feats_mean, md3.precision, md3._setup_flag = np.zeros((1, 8)), (a + a.T), True
    p4 = LaREMPipeline(md3, None, 16)
    x8 = rng.standard_normal((20, 8))
    s4 = p4.score_entropies(torch.from_numpy(x8).cuda()).cpu().numpy()
    assert p4._folded_state() is None
    assert rel_err(s4, oracle.md_score(x8, md3.feats_mean, md3.precision)) < 1e-11


def test_prepared_draws_equal_inline_table():
    """`prepare_draws` (K0 of a coming batch on a side stream, two table buffers in turn) + `score_latents(prepared=)`
    score the same bits as the in-line table launch - host draws and counter draws, several batches in flight."""
    from runia_core_amd import _hip
    from runia_core_amd.dimensionality_reduction import DevicePCA

    rng = np.random.default_rng(21)
    gp, gm = load_npz("ref_pca.npz"), load_npz("ref_md.npz")
    md = MDLatentSpace()
    md.feats_mean, md.precision, md._setup_flag = gm["d256_mean"], gm["d256_precision"], True
    pipe = LaREMPipeline(md, DevicePCA(gp["d512_components"], gp["d512_mean"], gp["d512_var"], True), 16, 0.5, 2)
    g = torch.Generator(device="cuda").manual_seed(5)
    batches = []
    for n in (3000, 3000, 1777, 3000, 1):
        x = torch.relu(torch.randn(n, 512, 4, 4, device="cuda", generator=g)).contiguous()
        r = torch.rand(n, 16, 4, 4, device="cuda", generator=g)
        batches.append((x, r))
    inline = [pipe.score_latents(x, r).cpu().numpy() for x, r in batches]
    inline_c = [pipe.score_latents(x, _hip.CounterDraws(9, 100 * i)).cpu().numpy() for i, (x, _) in enumerate(batches)]
    # tables prepared one batch ahead, as bench.py does
    prep = pipe.prepare_draws(batches[0][1], batches[0][0].shape[0], 4, 4)
    got = []
    for i, (x, r) in enumerate(batches):
        s = pipe.score_latents(x, r, prepared=prep)
        if i + 1 < len(batches):
            prep = pipe.prepare_draws(batches[i + 1][1], batches[i + 1][0].shape[0], 4, 4)
        got.append(s)
    for a, b in zip(inline, got):
        assert np.array_equal(a, b.cpu().numpy(), equal_nan=True)
    prep = pipe.prepare_draws(_hip.CounterDraws(9, 0), batches[0][0].shape[0], 4, 4)
    for i, (x, _) in enumerate(batches):
        s = pipe.score_latents(x, None, prepared=prep)
        if i + 1 < len(batches):
            prep = pipe.prepare_draws(_hip.CounterDraws(9, 100 * (i + 1)), batches[i + 1][0].shape[0], 4, 4)
        assert np.array_equal(inline_c[i], s.cpu().numpy(), equal_nan=True)
    with pytest.raises(ValueError):
        pipe.score_latents(batches[2][0], batches[2][1], prepared=pipe.prepare_draws(batches[0][1], 3000, 4, 4))
    assert pipe.prepare_draws(torch.rand(4, 16, 5, 6, device="cuda"), 4, 5, 6) is None  # no fused kernel for 5x6 maps


# ---------------- LaRED above D ~ 20: reference-run fixtures + cfg4 leg --------------------------------------------
@pytest.mark.parametrize("d", [16, 64, 256])
def test_lared_high_dim_reference_run_fixture(d, fit_on):
    """KDELatentSpace on the GPU against the reference's own LaRED scores (tests/golden/ref_kde_hd.npz).  D = 16: parity
    at 1e-5 and identical AUROC / FPR@95.  D = 64 / 256: the kernels compute the exact log-density (1e-9 against the
    oracle); the reference's sklearn tree returns the rounding residue of its node bounds instead (see
    tests/test_oracle_goldens.py::test_kde_high_dim_reference_run_fixture) - the measured gap, max |dscore|, dAUROC and
    dFPR@95, is printed and asserted so that the divergence stays a documented fact (LaRED)."""
    g = load_npz("ref_kde_hd.npz")
    train, ind, ood = kde_hd_inputs(d, int(g[f"d{d}_seed"]))
    kde = KDELatentSpace()
    kde.setup(train)
    s_i, s_o = kde.postprocess(ind), kde.postprocess(ood)
    assert s_i.dtype == np.float64 and s_i.shape == (600,)
    assert rel_err(s_i, oracle.kde_score(train, ind)) < 1e-9 and rel_err(s_o, oracle.kde_score(train, ood)) < 1e-9
    ref_i, ref_o = g[f"d{d}_ref_ind"], g[f"d{d}_ref_ood"]
    a_gpu, a_ref = oracle.auroc_fpr95_aupr(s_i, s_o), oracle.auroc_fpr95_aupr(ref_i, ref_o)
    gap = max(np.abs(s_i - ref_i).max(), np.abs(s_o - ref_o).max())
    print(f"LaRED D={d}: max|dscore| {gap:.3g}, AUROC gpu {a_gpu[0]:.6f} ref {a_ref[0]:.6f} (d {a_gpu[0] - a_ref[0]:+.4f}), "
          f"FPR@95 gpu {a_gpu[1]:.6f} ref {a_ref[1]:.6f} (d {a_gpu[1] - a_ref[1]:+.4f})")
    m = KDE_HD_MEASURED[d]
    assert abs(a_gpu[0] - m["auroc"][1]) < 1e-6 and abs(a_gpu[1] - m["fpr95"][1]) < 1e-6
    if d == 16:
        assert rel_err(s_i, ref_i) < 1e-5 and rel_err(s_o, ref_o) < 1e-5
        assert a_gpu[0] == a_ref[0] and a_gpu[1] == a_ref[1]
    else:
        assert np.all(ref_i >= s_i - 1e-8) and np.all(ref_o >= s_o - 1e-8)
        assert 0.5 * m["max_abs"] < gap < 2 * m["max_abs"]


@pytest.mark.parametrize("n_pca", [64, 256])
def test_cfg4_lared_leg(n_pca, fit_on):
    """BASELINE config 4, LaRED leg at full shape: 12 000 proposals x 16 MC x 1024-d -> per-dimension entropy -> PCA-64 /
    PCA-256 (whitened) -> KDELatentSpace fitted on 4 000 in-distribution proposals.  Sampled rows against the oracle's
    exact definition, slices against the whole, AUROC against the oracle's on the sample."""
    from runia_core_amd import _hip

    n_tr, n_te, n_mc, d = 4000, 12_000, 16, 1024
    g = torch.Generator(device="cuda").manual_seed(40 + n_pca)

    def proposals(n, spread):
        base = torch.randn(n, 1, d, device="cuda", generator=g) + 2
        rel = spread * (0.5 + torch.rand(1, 1, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)))
        return (base * (1 + rel * torch.randn(n, n_mc, d, device="cuda", generator=g))).reshape(n * n_mc, d).contiguous()

    h_tr = _hip.kl_entropy_per_dim(proposals(n_tr, 0.10), n_mc, 5).cpu().numpy()
    z_ind, z_ood = proposals(n_te, 0.10), proposals(n_te // 4, 0.13)
    h_ind, h_ood = _hip.kl_entropy_per_dim(z_ind, n_mc, 5), _hip.kl_entropy_per_dim(z_ood, n_mc, 5)
    np.random.seed(4)
    red, pca = rc.apply_pca_ds_split(h_tr, n_pca)
    kde = KDELatentSpace()
    kde.setup(red)
    y_ind = rc.apply_pca_transform(h_ind.cpu().numpy(), pca)
    y_ood = rc.apply_pca_transform(h_ood.cpu().numpy(), pca)
    s_ind, s_ood = kde.postprocess(y_ind), kde.postprocess(y_ood)
    assert s_ind.shape == (n_te,) and s_ind.dtype == np.float64 and np.isfinite(s_ind).all() and np.isfinite(s_ood).all()
    # oracle chain on a sample of proposals (entropy from the same MC samples, sklearn-closed-form PCA, exact KDE)
    idx = np.r_[0:64, n_te - 64:n_te]
    zs = z_ind.reshape(n_te, n_mc, d)[idx].reshape(-1, d).cpu().numpy()
    h_o = oracle.kl_entropy_per_dim_vectorized(zs, n_mc)
    y_o = oracle.pca_transform(h_o, pca.components_, pca.mean_, pca.explained_variance_)
    assert rel_err(s_ind[idx], oracle.kde_score(red, y_o)) < 1e-7
    # device-resident form of the same leg (entropy rows stay in HBM) = host API
    from runia_core_amd.dimensionality_reduction import device_pca_for

    s_dev = kde.postprocess_device(device_pca_for(pca).transform_device(h_ind)).cpu().numpy()
    assert rel_err(s_dev, s_ind) < 1e-12
    # rows are independent: a slice scores the same as within the whole
    assert np.array_equal(kde.postprocess(y_ind[5000:5300]), s_ind[5000:5300])
    a = oracle.auroc_fpr95_aupr(s_ind, s_ood)
    a_o = oracle.auroc_fpr95_aupr(oracle.kde_score(red, y_ind[:1500]), oracle.kde_score(red, y_ood[:1500]))
    a_g = oracle.auroc_fpr95_aupr(s_ind[:1500], s_ood[:1500])
    assert a_g == a_o and 0.5 < a[0] <= 1.0


# ---------------- f2: AUROC / FPR@95 / AUPR on the device -------------------------------------------------------------
def _scal(ref_vectors, key):
    return [v["value"] for v in ref_vectors[key]["scalars"]]


def test_device_metrics_reference_goldens(ref_vectors):
    """runia_ood_metrics_f64 on the reference's own metric goldens (/root/reference/tests/unit_test_metrics.py:21-29,
    TOL = 1e-7 there) and on the LaRED / LaREM end-to-end goldens of :31-80 (scores from KDELatentSpace / MDLatentSpace)."""
    from runia_core_amd.evaluation.metrics import auroc_fpr95_aupr_device

    np.random.seed(1)
    ind = 0.5 + np.random.randn(1000)
    ood = -0.5 + np.random.randn(1000)
    fpr95, aupr, auroc = _scal(ref_vectors, "metrics_hz")
    a, f, p = auroc_fpr95_aupr_device(ind, ood)
    assert abs(a - auroc) < 2e-7 and abs(f - fpr95) < 1e-7 and abs(p - aupr) < 2e-7
    assert (a, f, p) == pytest.approx(oracle.auroc_fpr95_aupr(ind, ood), abs=2e-7)


@pytest.mark.parametrize("case", ["f64_far", "f64_unit", "f32", "ties", "saturated", "tiny", "big", "clustered", "clustered_unit",
                                  "constant", "infinite", "two", "larem_like", "energy_like_f32", "ties_big", "infinite_big"])
def test_device_metrics_vs_oracle(case):
    """Device sort + scan metrics against the oracle's restatement of torchmetrics / sklearn: scores inside [0, 1] (no
    sigmoid), far outside (f64 sigmoid), float32 scores (f32 sigmoid), heavy ties, LaREM-like scores whose sigmoid
    underflows to 0 (everything below -745 ties, as in the reference), 3 + 2 scores, and 1.2 M + 0.9 M scores.  Round 4
    (the sort is a split into 4 096 buckets that are linear in the score + a sort per bucket): 300 000 scores inside a
    relative 1e-9 of each other next to a few outliers, squashed and inside [0, 1] - ONE bucket holds them, its keys differ in
    their low bytes only: the single-workgroup radix path; a constant score set (no bit differs: no pass); infinite scores
    (the end buckets); 1 + 1 scores."""
    from runia_core_amd.evaluation.metrics import auroc_fpr95_aupr_device

    # a fixed seed per case (hash(str) is salted per process: a red case could not have been reproduced)
    seeds = {"f64_far": 101, "f64_unit": 102, "f32": 103, "ties": 104, "saturated": 105, "tiny": 106, "big": 107, "clustered": 108,
             "clustered_unit": 109, "constant": 110, "infinite": 111, "two": 112, "larem_like": 113, "energy_like_f32": 114,
             "ties_big": 115, "infinite_big": 116}
    rng = np.random.default_rng(seeds[case])
    if case == "f64_far":
        ind, ood = rng.standard_normal(5000) * 3 + 1, rng.standard_normal(3000) * 3 - 1
    elif case == "f64_unit":
        ind, ood = rng.beta(4, 2, 4000), rng.beta(2, 3, 6000)
    elif case == "f32":
        ind, ood = (rng.standard_normal(7000) - 2).astype(np.float32), (rng.standard_normal(5000) - 3).astype(np.float32)
    elif case == "ties":
        ind, ood = rng.integers(0, 12, 9000).astype(np.float64) / 11.0, rng.integers(0, 9, 7000).astype(np.float64) / 11.0
    elif case == "saturated":
        ind, ood = -200 - 300 * rng.random(6000), -400 - 900 * rng.random(6000)
    elif case == "tiny":
        ind, ood = np.array([0.9, 0.4, 0.7]), np.array([0.1, 0.4])
    elif case == "clustered":
        ind = np.concatenate([3.0 + 1e-9 * rng.random(200_000), [-50.0, 40.0, 3.0]])
        ood = np.concatenate([3.0 + 1e-9 * rng.random(100_000) - 2e-10, [-60.0, 3.0]])
    elif case == "clustered_unit":
        ind = np.concatenate([0.5 + 1e-10 * rng.random(150_000), [0.0, 1.0]])
        ood = np.concatenate([0.5 + 1e-10 * rng.random(150_000) - 3e-11, [0.0, 0.25]])
    elif case == "constant":
        ind, ood = np.full(5000, 0.25), np.full(70_000, 0.25)
    elif case == "infinite":
        ind = np.concatenate([rng.standard_normal(3000), [np.inf, np.inf, -np.inf]])
        ood = np.concatenate([rng.standard_normal(3000) - 1, [np.inf, -np.inf, -np.inf]])
    elif case == "two":
        ind, ood = np.array([0.3]), np.array([0.7])
    elif case == "larem_like":      # round 5: -chi2(64) x 3, every score far outside [0, 1] - the sigmoids crowd against 0; the
        ind = -3.0 * rng.chisquare(64, 220_000)          # buckets follow the RAW score and are equalised with a sketch (> 262 144 scores)
        ood = -3.4 * rng.chisquare(64, 200_000)
    elif case == "energy_like_f32":  # float32 energies around 9: float32 sigmoids within 1e-4 of 1.0, heavy ties after the squash
        ind = (rng.standard_normal(180_000) * 2 + 9).astype(np.float32)
        ood = (rng.standard_normal(170_000) * 2 + 7).astype(np.float32)
    elif case == "ties_big":        # 40 distinct values over 400 000 scores: every equalised bucket boundary falls inside a run
        ind = rng.integers(0, 40, 210_000).astype(np.float64) * 0.37 - 5.0
        ood = rng.integers(0, 33, 190_000).astype(np.float64) * 0.37 - 6.0
    elif case == "infinite_big":
        ind = np.concatenate([rng.standard_normal(200_000) * 4, [np.inf] * 5, [-np.inf] * 3])
        ood = np.concatenate([rng.standard_normal(150_000) * 4 - 1, [np.inf, -np.inf, -np.inf]])
    else:
        ind, ood = rng.standard_normal(1_200_000) + 0.3, rng.standard_normal(900_000) - 0.3
    got = auroc_fpr95_aupr_device(ind, ood)
    exp = oracle.auroc_fpr95_aupr(ind, ood)
    tol = 3e-7 if case not in ("big", "larem_like", "energy_like_f32", "ties_big", "infinite_big") else 2e-6  # the reference adds its float32 trapezoid terms in float32
    assert got == pytest.approx(exp, abs=tol), (case, got, exp)
    # device-resident inputs, no host round trip
    import torch as _t

    a = _t.from_numpy(np.ascontiguousarray(ind)).cuda()
    b = _t.from_numpy(np.ascontiguousarray(ood)).cuda()
    out = auroc_fpr95_aupr_device(a, b, to_host=False)
    assert out.is_cuda and tuple(out.cpu().numpy()) == pytest.approx(got, abs=1e-12)
    # round 4: the term sums are added from per-workgroup records in index order (no float atomics): the same bits every run
    for _ in range(3):
        assert _t.equal(_t.nan_to_num(auroc_fpr95_aupr_device(a, b, to_host=False), nan=-1.0), _t.nan_to_num(out, nan=-1.0)), case


def test_get_auroc_results_drop_in_on_the_device(ref_vectors):
    """The harness entry point of the reference (evaluation/metrics.py:37-100) on the device sort: the reference's own
    metric goldens at its own tolerance (/root/reference/tests/unit_test_metrics.py:21-29, TOL = 1e-7), the table layout,
    the mlflow dict, and the returned ROC curve against the oracle's restatement of torchmetrics' binary roc."""
    np.random.seed(1)
    ind = 0.5 + np.random.randn(1000)
    ood = -0.5 + np.random.randn(1000)
    r = rc.evaluation.get_auroc_results("test", ind, ood, False)
    fpr95, aupr, auroc = _scal(ref_vectors, "metrics_hz")
    assert list(r.columns) == ["auroc", "fpr@95", "aupr", "fpr", "tpr"] and list(r.index) == ["test"]
    assert abs(r["auroc"].values[0] - auroc) < 1e-7
    assert abs(r["fpr@95"].values[0] - fpr95) < 1e-7
    assert abs(r["aupr"].values[0] - aupr) < 1e-7
    _, ml = rc.evaluation.get_auroc_results("test", ind, ood, True)
    assert set(ml) == {"auroc", "aupr", "fpr_95"}
    # the curve in the table = torchmetrics' roc (leading (0, 0), float32 ratios), from the device's runs
    from runia_core_amd.evaluation.metrics import auroc_fpr95_aupr

    rng = np.random.default_rng(3)
    cases = {
        "f64": (ind, ood),
        "f32": ((rng.standard_normal(5000) - 2).astype(np.float32), (rng.standard_normal(3000) - 3).astype(np.float32)),
        "ties": (rng.integers(0, 12, 9000) / 11.0, rng.integers(0, 9, 7000) / 11.0),
        # -0.0 and +0.0 are ONE run for torchmetrics (preds[1:] - preds[:-1] != 0), on either side of the classes
        "signed_zero": (np.array([0.0, -0.0, 0.5, 0.25, -0.0]), np.array([-0.0, 0.0, 0.25, 0.0])),
        "saturated": (-200 - 300 * rng.random(4000), -400 - 900 * rng.random(4000)),
    }
    for name, (a, b) in cases.items():
        got = auroc_fpr95_aupr(a, b)
        exp = oracle.auroc_fpr95_aupr(a, b)
        assert got[:3] == pytest.approx(exp, abs=1e-12), (name, got[:3], exp)
        dt = np.float32 if (a.dtype == np.float32 and b.dtype == np.float32) else np.float64
        sc = np.concatenate([a, b]).astype(dt)
        if not np.all((sc >= 0) & (sc <= 1)):
            with np.errstate(over="ignore"):
                sc = (dt(1) / (dt(1) + np.exp(-sc))).astype(dt)
        fps, tps, _ = oracle.binary_clf_curve(sc.astype(np.float64), np.r_[np.ones(a.size, np.int64), np.zeros(b.size, np.int64)])
        fpr = np.concatenate([[0], fps]).astype(np.float32) / np.float32(fps[-1])
        tpr = np.concatenate([[0], tps]).astype(np.float32) / np.float32(tps[-1])
        assert got[3].dtype == np.float32 and got[4].dtype == np.float32
        if name == "f32":
            # the float32 sigmoid differs by an ulp between expf on the device and on the host: a pair of scores may tie
            # on one side only (a point more or less on the curve); the curves are the same function
            assert abs(len(got[3]) - len(fpr)) <= 2
            grid = np.linspace(0, 1, 2001)
            assert np.abs(np.interp(grid, got[3], got[4]) - np.interp(grid, fpr, tpr)).max() < 1e-3
        else:
            assert np.array_equal(got[3], fpr) and np.array_equal(got[4], tpr), name
        # device-resident scores take the same path without an upload
        got_dev = auroc_fpr95_aupr(torch.from_numpy(np.ascontiguousarray(a)).cuda(), torch.from_numpy(np.ascontiguousarray(b)).cuda())
        assert got_dev[:3] == got[:3] and np.array_equal(got_dev[3], got[3])
    # the device-only scalars agree on the signed-zero case too (one tie group, not two)
    from runia_core_amd.evaluation.metrics import auroc_fpr95_aupr_device

    a, b = cases["signed_zero"]
    assert auroc_fpr95_aupr_device(a, b) == pytest.approx(oracle.auroc_fpr95_aupr(a, b), abs=3e-7)


# ---------------- f1 / f4: Jacobi eigen-solver, pinvh, PCA fit, eigen_score ----------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 10, 33, 64, 257])
def test_jacobi_eigh_vs_numpy(n):
    from runia_core_amd import _hip

    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    a = a @ a.T + np.diag(rng.random(n))
    if n > 3:  # rank-deficient block + repeated eigenvalues
        a[:, -2] = a[:, -3]
        a[-2, :] = a[-3, :]
    w, v = _hip.eigh(torch.from_numpy(a).cuda())
    w, v = w.cpu().numpy(), v.cpu().numpy()
    w_ref = np.linalg.eigvalsh(a)
    scale = max(1.0, np.abs(w_ref).max())
    assert np.abs(w - w_ref).max() < 1e-12 * scale
    assert np.abs(v.T @ v - np.eye(n)).max() < 1e-12
    assert np.abs(a @ v - v * w).max() < 1e-11 * scale


@pytest.mark.parametrize("kind", ["whitened", "identity_noise", "two_clusters", "indefinite", "rank_deficient", "zeros"])
@pytest.mark.parametrize("blocked", [True, False])
def test_jacobi_eigh_terminates_on_clustered_spectra(kind, blocked):
    """Spectra on which a Jacobi sweep criterion can fail to settle: the precision matrix of PCA-whitened data (every
    eigenvalue ~ 1: what LaREMPipeline folds at first use), an identity plus rounding noise, two exact clusters, a
    zero-diagonal indefinite matrix, a rank-10 Gram matrix, the zero matrix.  Both solver forms must stop within the sweep
    limit and agree with LAPACK."""
    from runia_core_amd import _hip

    rng = np.random.default_rng(len(kind))
    n = 200 if kind != "two_clusters" else 64
    if kind == "whitened":
        x = rng.standard_normal((2048, n))
        x -= x.mean(0)
        w, v = np.linalg.eigh(x.T @ x / 2048)
        a = np.linalg.pinv(np.cov((x @ v / np.sqrt(w)).T, bias=True))
    elif kind == "identity_noise":
        e = rng.standard_normal((n, n)) * 1e-16
        a = np.eye(n) + e + e.T
    elif kind == "two_clusters":
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        a = (q * np.r_[np.ones(n // 2), 3 * np.ones(n - n // 2)]) @ q.T
    elif kind == "indefinite":
        g = rng.standard_normal((n, n))
        a = g + g.T
        np.fill_diagonal(a, 0.0)
    elif kind == "rank_deficient":
        g = rng.standard_normal((n, 10))
        a = g @ g.T
    else:
        a = np.zeros((n, n))
    a = (a + a.T) * 0.5
    w, v = _hip.eigh(torch.from_numpy(a).cuda(), max_sweeps=30, blocked=blocked)
    w, v = w.cpu().numpy(), v.cpu().numpy()
    nrm = max(1e-300, float(np.abs(a).max()))
    assert np.abs(w - np.linalg.eigvalsh(a)).max() / nrm < 1e-11
    assert np.abs(a @ v - v * w).max() / nrm < 1e-11 and np.abs(v.T @ v - np.eye(n)).max() < 1e-11


def test_device_fit_pinvh_and_pca_without_vendor_solver(monkeypatch):
    """config.device_fit: MDLatentSpace.setup (covariance on the matrix cores + Jacobi pinvh, rank-deficient unit case
    included) against the reference-run fixtures, and apply_pca_ds_split(svd_solver="covariance_eigh") against sklearn."""
    from sklearn.decomposition import PCA

    import runia_core_amd.config as cfg

    monkeypatch.setattr(cfg, "device_fit", True)
    g = load_npz("ref_md.npz")
    for name in ("unit", "baselines"):
        md = MDLatentSpace()
        md.setup(g[f"{name}_train"])
        scale = np.abs(g[f"{name}_precision"]).max()
        assert np.abs(md.precision - g[f"{name}_precision"]).max() < 1e-8 * scale
        assert rel_err(md.postprocess(g[f"{name}_test"]), g[f"{name}_scores"]) < 1e-6
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3000, 96)) * (0.3 + rng.random(96)) + rng.standard_normal(96)
    red, fitted = rc.apply_pca_ds_split(x, 24, svd_solver="covariance_eigh")
    ref = PCA(n_components=24, svd_solver="covariance_eigh", whiten=True).fit(x)
    assert np.abs(fitted.components_ - ref.components_).max() < 1e-9
    assert np.abs(fitted.explained_variance_ - ref.explained_variance_).max() < 1e-10
    assert np.abs(fitted.mean_ - ref.mean_).max() < 1e-12
    assert rel_err(red, ref.transform(x)) < 1e-8
    xt = rng.standard_normal((50, 96))
    assert rel_err(rc.apply_pca_transform(xt, fitted), ref.transform(xt)) < 1e-8
    assert rel_err(fitted.transform(xt), ref.transform(xt)) < 1e-8


def test_eigen_score_reference_golden():
    """/root/reference/tests/unit_test_llm_uncertainty.py:69-92: seeded synthetic hidden states, golden
    -6.775187082486514 at the reference's own tolerance 1e-6; alpha dependence; device inputs."""
    from runia_core_amd.llm_uncertainty import eigen_score

    np.random.seed(42)
    torch.manual_seed(42)
    hs = tuple(tuple(torch.randn(1, 10, 768) for _ in range(20)) for _ in range(5))
    s = eigen_score(hs, alpha=1e-3)
    assert isinstance(s, float) and abs(s - (-6.775187082486514)) < 1e-6
    # the definition, in float64 on the host
    e = hs[-1][15].squeeze().double().numpy()
    cov = np.cov(e.T)
    sv = np.linalg.svd(cov + 1e-3 * np.eye(768), compute_uv=False)
    assert abs(s - float(np.mean(np.log(sv)))) < 1e-9
    hs2 = tuple(tuple(torch.randn(1, 5, 64) for _ in range(20)) for _ in range(3))
    s1, s2 = eigen_score(hs2, alpha=1e-3), eigen_score(hs2, alpha=1e-2)
    assert abs(s1 - s2) > 1e-3 and eigen_score(hs2) == s1
    hs_dev = tuple(tuple(t.cuda() for t in layer) for layer in hs2)
    assert abs(eigen_score(hs_dev) - s1) < 1e-12


def test_eigen_score_at_llama_width():
    """BASELINE config 5's shape: 10 samples per prompt, hidden = 4096 (Llama-3.1-8B), against the float64 host
    definition mean(log(svd(cov + alpha I))) - the O(hidden^3) computation the Gram form replaces - and for n > hidden."""
    from runia_core_amd.llm_uncertainty import eigen_score

    torch.manual_seed(7)
    for n, hidden, alpha in ((10, 4096, 1e-3), (10, 4096, 1e-2), (40, 32, 1e-3)):
        layer = tuple(torch.randn(1, n, hidden) * (0.5 + torch.rand(hidden)) for _ in range(20))
        hs = (layer,)
        e = layer[15].squeeze().double().numpy()
        cov = np.cov(e.T)
        sv = np.linalg.svd(cov + alpha * np.eye(hidden), compute_uv=False)
        s = eigen_score(hs, alpha=alpha)
        # the Gram matrix is formed from float32 embeddings (as the reference's torch.cov is): 1e-6 of the mean log
        assert abs(s - float(np.mean(np.log(sv)))) < 2e-6, (n, hidden, alpha)


def test_refit_invalidates_device_caches():
    """Fitted-state caches follow the live attributes: refitting the same sklearn PCA object, or reassigning
    precision / feats_mean, must change the scores."""
    from sklearn.decomposition import PCA

    rng = np.random.default_rng(8)
    x1, x2 = rng.standard_normal((400, 20)), rng.standard_normal((400, 20)) * 2 + 1
    xt = rng.standard_normal((30, 20))
    pca = PCA(n_components=5, whiten=True).fit(x1)
    y1 = rc.apply_pca_transform(xt, pca)
    pca.fit(x2)
    y2 = rc.apply_pca_transform(xt, pca)
    assert rel_err(y2, pca.transform(xt)) < 1e-9 and np.abs(y1 - y2).max() > 1e-3
    md = MDLatentSpace()
    md.setup(x1)
    s1 = md.postprocess(xt)
    md.precision = md.precision * 4.0
    assert rel_err(md.postprocess(xt), 4.0 * s1) < 1e-12
    md.feats_mean = md.feats_mean + 1.0
    assert np.abs(md.postprocess(xt) - 4.0 * s1).max() > 1e-3


# ---------------- f3: roi_align -> per-ROI MC DropBlock -> entropy on the device ---------------------------------------
@pytest.mark.parametrize("name", ["p7", "p4x2", "p8_adaptive"])
def test_per_roi_entropy_reference_run_fixture(name):
    """runia_core_amd.feature_extraction.object_level._dropblock_rois_get_entropy (same signature as the reference's)
    against what the reference's own function returned (tests/golden/ref_roi.npz): roi_align kernel + fused sampler /
    entropy kernels; the module's CPU-generator draws are the reference's stream (detection after detection)."""
    from runia_core_amd import MCSamplerModule
    from runia_core_amd.feature_extraction.object_level import _dropblock_rois_get_entropy, _reduce_features_to_rois, roi_align

    g = load_npz("ref_roi.npz")
    n_rep, osz, ih, iw, sr, n_mc, bs, p, seed = g[f"{name}_params"]
    n_rep, osz, sr, n_mc, bs = int(n_rep), int(osz), int(sr), int(n_mc), int(bs)
    fms = [torch.from_numpy(g[f"{name}_fm{i}"]).cuda() for i in range(n_rep)]
    boxes = torch.from_numpy(g[f"{name}_boxes"])
    sampler = MCSamplerModule(mc_samples=n_mc, block_size=bs, drop_prob=float(p), layer_type="Conv")
    sampler.train()
    torch.manual_seed(int(seed))
    ent = _dropblock_rois_get_entropy(fms, (osz,) * n_rep, boxes, (int(ih), int(iw)), sr, n_rep, n_mc, sampler)
    assert isinstance(ent, torch.Tensor) and ent.dtype == torch.float32 and not ent.is_cuda
    ref = g[f"{name}_entropy"]
    assert ent.shape == ref.shape and np.abs(ent.numpy() - ref).max() < 2e-5
    # explicit draws = the fixture's
    ent2 = _dropblock_rois_get_entropy(fms, (osz,) * n_rep, boxes, (int(ih), int(iw)), sr, n_rep, n_mc, sampler,
                                       rand=torch.from_numpy(g[f"{name}_draws"]).cuda())
    assert torch.equal(ent, ent2)
    # roi_align kernel against the oracle's restatement (float32, same sample order; fma contraction may differ by ulps)
    for i, fm in enumerate(fms):
        r = roi_align(fm, [boxes], osz, fm.shape[3] / iw, sr, True).cpu().numpy()
        exp = oracle.roi_align(g[f"{name}_fm{i}"], g[f"{name}_boxes"], osz, fm.shape[3] / iw, sr, True)
        assert np.allclose(r, exp, rtol=2e-6, atol=2e-6)
    means, stds = _reduce_features_to_rois(fms, (osz,) * n_rep, boxes, (int(ih), int(iw)), sr, n_rep, boxes.shape[0], True)
    assert np.allclose(torch.cat(means).cpu().numpy(), g[f"{name}_means"], rtol=1e-5, atol=1e-6)
    assert np.allclose(torch.cat(stds).cpu().numpy(), g[f"{name}_stds"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("sr,aligned,osz", [(2, True, 7), (1, True, 7), (2, False, 7), (2, True, (4, 8)), (2, True, 8), (0, True, 7),
                                            (3, True, 7), (2, True, 9)])
def test_roi_align_options_and_edge_boxes_vs_oracle(sr, aligned, osz):
    """runia_roi_align_f32 against the oracle's restatement of torchvision's algorithm over its options - fixed 1x1 / 2x2 /
    3x3 and adaptive sampling lattices, aligned or not, square and rectangular bins - on boxes inside, across the border,
    entirely outside the map and degenerate (zero area), two images with batch indices."""
    from runia_core_amd import _hip

    rng = np.random.default_rng(17)
    b, c, h, w = 2, 37, 20, 31
    fm = rng.standard_normal((b, c, h, w)).astype(np.float32)
    boxes = np.array([[10.0, 12.0, 90.0, 70.0], [-30.0, -20.0, 40.0, 35.0], [200.0, 100.0, 260.0, 170.0],
                      [300.0, 300.0, 340.0, 350.0], [50.0, 50.0, 50.0, 50.0], [0.0, 0.0, 247.0, 159.0],
                      [120.5, 33.25, 121.0, 140.75], [5.0, 150.0, 240.0, 158.0]], dtype=np.float32)
    bidx = np.array([0, 1, 0, 1, 1, 0, 1, 0], dtype=np.int32)
    scale = w / 248.0
    got = _hip.roi_align(torch.from_numpy(fm).cuda(), torch.from_numpy(boxes).cuda(), osz, scale, sr, aligned,
                         batch_idx=torch.from_numpy(bidx).cuda()).cpu().numpy()
    exp = np.concatenate([oracle.roi_align(fm[bi:bi + 1], boxes[i:i + 1], osz, scale, sr, aligned) for i, bi in enumerate(bidx)])
    assert got.shape == exp.shape
    assert np.allclose(got, exp, rtol=2e-6, atol=2e-6), float(np.abs(got - exp).max())


def test_cfg4_per_roi_path_at_size():
    """Config 4 shape end to end on the device: 100 proposals per image on a 256-channel 50x80 feature map, 7x7 ROI
    bins, 16 MC DropBlock layers (counter draws) -> (100, 256) entropies per image; rows are independent of the other
    boxes in the call, and equal the unfused chain (roi_align -> mc_stack -> entropy)."""
    from runia_core_amd import MCSamplerModule, _hip
    from runia_core_amd.feature_extraction.object_level import _dropblock_rois_get_entropy, roi_align

    g = torch.Generator(device="cuda").manual_seed(3)
    fm = torch.relu(torch.randn(1, 256, 50, 80, device="cuda", generator=g))
    k = 100
    xy = torch.rand(k, 2, generator=torch.Generator().manual_seed(1)) * torch.tensor([400.0, 250.0])
    wh = 30 + torch.rand(k, 2, generator=torch.Generator().manual_seed(2)) * torch.tensor([200.0, 120.0])
    boxes = torch.cat([xy, xy + wh], dim=1)
    sampler = MCSamplerModule(mc_samples=16, block_size=3, drop_prob=0.4).train().use_counter_draws(seed=9)
    ent = _dropblock_rois_get_entropy([fm], (7,), boxes, (400, 640), 2, 1, 16, sampler)
    assert ent.shape == (k, 256) and bool(torch.isfinite(ent).all())
    sampler.use_counter_draws(seed=9)  # same stream again
    part = _dropblock_rois_get_entropy([fm], (7,), boxes[:10], (400, 640), 2, 1, 16, sampler)
    assert torch.equal(part, ent[:10])
    rois = roi_align(fm, [boxes], 7, 80 / 640, 2, True)
    z = _hip.mc_stack(rois, _hip.CounterDraws(9, 0), 16, 0.4, 3)
    h = _hip.kl_entropy_per_dim(z, 16, 5).to(torch.float32).cpu()
    assert float((h - ent).abs().max()) < 1e-6


def test_fast_mcd_samples_extractor_batched():
    """FastMCDSamplesExtractor.get_ls_samples (reference feature_extraction/image_level.py:127-249) over a dataloader with
    batches of 1 and of 3 images: the same MC samples as the oracle's MCSamplerModule.forward per image with the
    reference's draw stream (one torch.rand(1, H, W) per sample, image after image), gt labels and raw predictions."""
    from runia_core_amd import Hook
    from runia_core_amd.feature_extraction import FastMCDSamplesExtractor

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 12, 3, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(4)).cuda().eval()
    hook = Hook(net[2])
    imgs = torch.randn(7, 3, 16, 16)
    labels = torch.arange(7)
    for bs in (1, 3):
        loader = [(imgs[i : i + bs], labels[i : i + bs]) for i in range(0, 7, bs)]
        ex = FastMCDSamplesExtractor(net, [hook], torch.device("cuda"), "Conv", "fullmean", return_raw_predictions=True,
                                     mcd_nro_samples=16, dropblock_probs=0.5, dropblock_sizes=2, return_gt_labels=True)
        torch.manual_seed(123)
        res = ex.get_ls_samples(loader)
        z = res["latent_space_means"]
        assert z.is_cuda and z.shape == (7 * 16, 12) and res["raw_preds"].shape == (7, 12, 4, 4)
        assert res["gt_labels"].numel() == 7
        with torch.no_grad():  # the same batches the extractor ran: a convolution may round differently per batch shape
            lat = np.concatenate([net(im.cuda()).cpu().numpy() for im, _ in loader])
        torch.manual_seed(123)
        draws = torch.cat([torch.rand(1, 4, 4) for _ in range(7 * 16)]).reshape(7, 16, 4, 4).numpy()
        exp = np.concatenate([oracle.mc_stack(lat[i : i + 1], draws[i], 0.5, 2) for i in range(7)])
        if bs == 1:
            assert np.array_equal(z.cpu().numpy(), exp, equal_nan=True)
        else:  # batches draw per batch: the stream is the same, image after image
            assert np.array_equal(z.cpu().numpy(), exp, equal_nan=True)


def _ref_samples_one_image(latents, draws, probs, sizes, reduction, want_std):
    """/root/reference/runia_core/feature_extraction/image_level.py:162-249 for one image, on the CPU in torch: latents =
    per-layer (1, C, H, W) arrays, draws[i][s] = the (H, W) uniform draws of layer i in MC sample s."""
    means, stds = [], []
    for s in range(len(next(d for d in draws if d is not None))):
        m_layers, s_layers = [], []
        for i, x in enumerate(latents):
            c, h, w = x.shape[1:]
            if probs[i] != 0.0:
                y = oracle.mc_stack(x, draws[i][s : s + 1], probs[i], sizes[i], layer_type="FC").reshape(1, c, h, w)
            else:
                y = x
            y = torch.from_numpy(np.ascontiguousarray(y))
            if reduction == "fullmean":
                m = torch.squeeze(torch.mean(torch.mean(y, dim=3, keepdim=True), dim=2, keepdim=True))
            else:
                m = torch.squeeze(torch.mean(y, dim=3, keepdim=True))
            m_layers.append(m.reshape(-1))
            if want_std:
                s_layers.append(torch.squeeze(torch.std(torch.std(y, dim=3, keepdim=True), dim=2, keepdim=True)).reshape(-1))
        means.append(torch.cat(m_layers).reshape(1, -1))
        if want_std:
            stds.append(torch.cat(s_layers).reshape(1, -1))
    return torch.cat(means).numpy(), (torch.cat(stds).numpy() if want_std else None)


@pytest.mark.parametrize("reduction,want_std,n_layers", [("mean", False, 1), ("fullmean", True, 1), ("mean", True, 2),
                                                          ("fullmean", False, 3)])
def test_fast_mcd_samples_extractor_other_options(reduction, want_std, n_layers):
    """The remaining options of the reference's extractor loop (image_level.py:162-249): reduction_method="mean",
    return_stds, several hooked layers with their own DropBlock layer (one of them with drop_prob 0: it draws nothing),
    against a torch CPU restatement of that loop fed with the same generator stream."""
    from runia_core_amd import Hook
    from runia_core_amd.feature_extraction import FastMCDSamplesExtractor

    class Branches(torch.nn.Module):  # a layer whose INPUT is a list of feature maps (as an FPN head's)
        def forward(self, feats):
            return sum(f.mean() for f in feats)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = torch.nn.Conv2d(3, 10, 3, padding=1)
            self.c2 = torch.nn.Conv2d(10, 6, 3, padding=1, stride=2)
            self.c3 = torch.nn.Conv2d(6, 5, 3, padding=1, stride=2)
            self.head = Branches()
            self.single = torch.nn.Identity()

        def forward(self, x):
            a = torch.relu(self.c1(x))
            b = torch.relu(self.c2(a))
            c = torch.relu(self.c3(b))
            self.single(a)
            return self.head([a, b, c][: self.n_out])

    torch.manual_seed(1)
    net = Net().cuda().eval()
    net.n_out = n_layers
    imgs = torch.randn(5, 3, 8, 8)
    probs_all, sizes_all = [0.4, 0.0, 0.3], [3, 1, 2]
    if n_layers == 1:
        hook, probs, sizes, out_hook = Hook(net.single), 0.4, 3, True
    else:
        hook, probs, sizes, out_hook = Hook(net.head), probs_all[:n_layers], sizes_all[:n_layers], False
    n_mc = 6
    ex = FastMCDSamplesExtractor(net, [hook], torch.device("cuda"), "Conv", reduction, return_stds=want_std,
                                 mcd_nro_samples=n_mc, hook_layer_output=out_hook, dropblock_probs=probs, dropblock_sizes=sizes)
    loader = [(imgs[0:2], torch.zeros(2)), (imgs[2:5], torch.zeros(3))]
    torch.manual_seed(77)
    res = ex.get_ls_samples(loader)
    parts = []
    with torch.no_grad():  # per loader batch: a convolution may round differently per batch shape
        for im, _ in loader:
            a = torch.relu(net.c1(im.cuda()))
            b = torch.relu(net.c2(a))
            c = torch.relu(net.c3(b))
            parts.append([t.cpu().numpy() for t in (a, b, c)])
    lat_all = [np.concatenate([p[i] for p in parts]) for i in range(3)][:n_layers]
    pl, sl = ([probs], [sizes]) if n_layers == 1 else (probs, sizes)
    torch.manual_seed(77)
    exp_m, exp_s = [], []
    for i in range(5):
        draws = [[] if p != 0.0 else None for p in pl]
        for _ in range(n_mc):
            for li, p in enumerate(pl):
                if p != 0.0:
                    draws[li].append(torch.rand(1, *lat_all[li].shape[2:]).numpy())
        draws = [np.concatenate(d) if d is not None else None for d in draws]
        m, sd = _ref_samples_one_image([t[i : i + 1] for t in lat_all], draws, pl, sl, reduction, want_std)
        exp_m.append(m)
        exp_s.append(sd)
    got = res["latent_space_means"].cpu().numpy()
    exp = np.concatenate(exp_m)
    assert got.shape == exp.shape
    assert np.allclose(got, exp, rtol=2e-6, atol=1e-7, equal_nan=True)
    if want_std:
        gs, es = res["stds"].cpu().numpy(), np.concatenate(exp_s)
        assert gs.shape == es.shape and np.allclose(gs, es, rtol=1e-5, atol=1e-7, equal_nan=True)


def test_fast_mcd_samples_extractor_fc_layer():
    """layer_type="FC": torch.nn.Dropout on the hooked vector, mcd_nro_samples rows per image, image-major; a kept entry is
    x / (1 - p), a dropped one 0 (the device generator makes the masks, as upstream on a GPU)."""
    from runia_core_amd import Hook
    from runia_core_amd.feature_extraction import FastMCDSamplesExtractor

    torch.manual_seed(2)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(48, 200)).cuda().eval()
    hook = Hook(net[1])
    imgs = torch.randn(4, 3, 4, 4)
    ex = FastMCDSamplesExtractor(net, [hook], torch.device("cuda"), "FC", "fullmean", mcd_nro_samples=50, dropblock_probs=0.25,
                                 dropblock_sizes=0)
    z = ex.get_ls_samples([(imgs[:1], torch.zeros(1)), (imgs[1:], torch.zeros(3))])["latent_space_means"]
    assert z.shape == (4 * 50, 200)
    with torch.no_grad():
        full = net(imgs.cuda())
    rows = z.reshape(4, 50, 200)
    kept = rows != 0
    assert torch.allclose(rows[kept], (full[:, None, :] / 0.75).expand(4, 50, 200)[kept], rtol=1e-6)
    assert abs(float(kept.float().mean()) - 0.75) < 0.02
    with pytest.raises(NotImplementedError):
        FastMCDSamplesExtractor(net, [hook], torch.device("cuda"), "FC", "fullmean", mcd_nro_samples=4, dropblock_probs=[0.1, 0.2],
                                dropblock_sizes=[0, 0])


def test_device_fit_randomized_pca_reproduces_sklearn(monkeypatch):
    """apply_pca_ds_split with the reference's DEFAULT solver ("randomized") on the device: for the same state of NumPy's
    global generator it returns sklearn's components / variances / transform (covariance-space range finder + Jacobi
    eigen-solver, device_fit.pca_fit_randomized_device), and it leaves the generator in the same state as sklearn does."""
    from sklearn.decomposition import PCA

    import runia_core_amd.config as cfg

    rng = np.random.default_rng(0)
    n, d = 6000, 192
    basis = np.linalg.qr(rng.standard_normal((d, d)))[0]
    spec = np.exp(-np.arange(d) / 40.0) * 3 + 0.05
    x = (rng.standard_normal((n, d)) * spec) @ basis.T + rng.standard_normal(d)
    xt = (rng.standard_normal((64, d)) * spec) @ basis.T
    for k in (96, 8):  # n_iter = 4 and n_iter = 7 in sklearn's rule
        np.random.seed(7)
        ref = PCA(n_components=k, svd_solver="randomized", whiten=True)
        ref_red = ref.fit_transform(x)
        after_ref = np.random.random()
        monkeypatch.setattr(cfg, "device_fit", True)
        np.random.seed(7)
        red, fitted = rc.apply_pca_ds_split(x, k)  # default solver
        after_dev = np.random.random()
        monkeypatch.setattr(cfg, "device_fit", False)
        assert fitted.svd_solver == "randomized" and after_dev == after_ref
        assert np.abs(fitted.components_ - ref.components_).max() < 1e-8
        assert np.abs(fitted.explained_variance_ / ref.explained_variance_ - 1).max() < 1e-9
        assert np.abs(fitted.explained_variance_ratio_ - ref.explained_variance_ratio_).max() < 1e-10
        assert rel_err(red, ref_red) < 1e-7
        assert rel_err(rc.apply_pca_transform(xt, fitted), ref.transform(xt)) < 1e-7


def test_larex_inference_fc_layer_type():
    """LaRExInference with layer_type="FC": the sampler returns the flattened drop-layer outputs (C*H*W dims per sample,
    reference feature_extraction/abstract_classes.py:95-99), entropies / PCA / LaREM follow; against the oracle chain."""
    from runia_core_amd import LaRExInference, MCSamplerModule

    rng = np.random.default_rng(4)
    c, h, w, n_mc, n = 6, 4, 4, 16, 5
    dflat = c * h * w
    comp = np.linalg.qr(rng.standard_normal((dflat, 12)))[0].T
    pca = type("P", (), {})()
    pca.components_, pca.mean_, pca.explained_variance_, pca.whiten = comp, rng.standard_normal(dflat), rng.random(12) + 0.1, True
    a = rng.standard_normal((12, 12))
    md = MDLatentSpace()
    md.feats_mean, md.precision, md._setup_flag = rng.standard_normal((1, 12)) * 0.1, a @ a.T / 12 + np.eye(12), True
    inf = LaRExInference(torch.nn.Identity(), md, 0.5, 2, n_mc, MCSamplerModule, pca_transform=pca, layer_type="FC")
    x = np.maximum(rng.standard_normal((n, c, h, w)), 0).astype(np.float32) + 0.1
    rand = rng.random((n, n_mc, h, w)).astype(np.float32)
    rand[:, :, 0, 0] = np.maximum(rand[:, :, 0, 0], 0.2)
    s = inf.get_scores_from_latents(torch.from_numpy(x), rand=torch.from_numpy(rand).cuda())
    z = np.concatenate([oracle.mc_stack(x[i : i + 1], rand[i], 0.5, 2, "FC") for i in range(n)])
    assert z.shape == (n * n_mc, dflat)
    exp, _ = oracle.larem_pipeline(z, n_mc, comp, pca.mean_, pca.explained_variance_, md.feats_mean, md.precision)
    assert s.shape == (n,) and rel_err(s, exp) < 1e-9


# ---- round 4: the free functions of inference/funcs.py (fixtures: tools/make_goldens_r4.py, the reference's own file run by path)
@pytest.mark.gpu
def test_funcs_mirror_against_reference_run_fixture():
    from runia_core_amd.inference import (ash_s_conv_layer, ash_s_linear_layer, generalized_entropy,
                                          get_dice_feat_mean_react_percentile, get_mcd_pred_uncertainty_score,
                                          get_predictive_uncertainty_score)

    g = load_npz("ref_funcs_r4.npz")
    # ash_s_linear_layer: tie-free rows; the reference's scatter quirk (values of np.partition at the indices of
    # np.argpartition) permutes kept values inside some rows, so rows are compared as sorted multisets + by their support
    for tag in ("a", "b"):
        x, pct, ref = g[f"ashl_{tag}_x"], int(g[f"ashl_{tag}_pct"]), g[f"ashl_{tag}_y"]
        got = ash_s_linear_layer(x.copy(), pct)
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got != 0, ref != 0)
        assert rel_err(np.sort(got, axis=1), np.sort(ref, axis=1)) < 1e-5
        assert rel_err(got, oracle.ash_s_defined(x, pct)) < 1e-5
    # ash_s_conv_layer: output and the argument left pruned in place, on a CPU tensor (as the reference's callers hold)
    # and on a device tensor
    for tag in ("a", "b"):
        x, pct = g[f"ashc_{tag}_x"], int(g[f"ashc_{tag}_pct"])
        for on_gpu in (False, True):
            xin = torch.from_numpy(x.copy())
            xin = xin.cuda() if on_gpu else xin
            y = ash_s_conv_layer(xin, pct)
            assert y.device == xin.device and y.shape == xin.shape
            assert rel_err(y.cpu().numpy(), g[f"ashc_{tag}_y"]) < 1e-5
            assert np.array_equal(xin.cpu().numpy(), g[f"ashc_{tag}_x_after"])
    # generalized_entropy: dtype of the probabilities kept
    for tag in ("a", "b", "c"):
        p, (gamma, m), ref = g[f"gen_{tag}_p"], g[f"gen_{tag}_gm"], g[f"gen_{tag}_s"]
        got = generalized_entropy(p, float(gamma), int(m))
        assert got.dtype == ref.dtype and rel_err(got, ref) < 1e-5
        assert rel_err(generalized_entropy(torch.from_numpy(p), float(gamma), int(m)), ref) < 1e-5
    # pred_h / mi: heads of 10 (lane per image), 43, 100 and 1000 classes (wave per image); 2 ... 32 MC samples
    for tag in ("a", "b", "c", "d"):
        logits, n_mc = torch.from_numpy(g[f"pu_{tag}_logits"]), int(g[f"pu_{tag}_nmc"])
        ph, mi = get_predictive_uncertainty_score(logits, n_mc)
        assert ph.dtype == torch.float32 and ph.device == logits.device
        assert rel_err(ph.numpy(), g[f"pu_{tag}_pred_h"]) < 1e-5 and rel_err(mi.numpy(), g[f"pu_{tag}_mi"]) < 1e-5
        ph2, mi2 = get_predictive_uncertainty_score(logits.cuda(), n_mc)
        assert ph2.is_cuda and np.array_equal(ph2.cpu().numpy(), ph.numpy()) and np.array_equal(mi2.cpu().numpy(), mi.numpy())
    # the dataloader form, fed the MC outputs the reference's model produced (recorded through a forward hook)
    rows = torch.from_numpy(g["mcd_logits"])
    n_mc = int(g["mcd_nmc"])

    class Replay(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.i = 0

        def forward(self, image):
            out = rows[self.i: self.i + 1].to(image.device)
            self.i += 1
            return out

    loader = [(torch.zeros(1, 3, 4, 4), torch.zeros(1)) for _ in range(rows.shape[0] // n_mc)]
    samples, ph, mi = get_mcd_pred_uncertainty_score(Replay(), loader, n_mc)
    assert samples.shape == g["mcd_samples"].shape
    assert rel_err(samples.cpu().numpy(), g["mcd_samples"]) < 1e-5
    assert rel_err(ph.cpu().numpy(), g["mcd_pred_h"]) < 1e-5 and rel_err(mi.cpu().numpy(), g["mcd_mi"]) < 1e-5

    # DICE info + ReAct threshold with the reference's toy model
    class Feat(torch.nn.Module):
        dice_precompute = True

        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 12, 3, padding=1)

        def forward(self, x):
            return torch.relu(self.conv(x))

    fm = Feat()
    with torch.no_grad():
        fm.conv.weight.copy_(torch.from_numpy(g["dice_w"]))
        fm.conv.bias.copy_(torch.from_numpy(g["dice_b"]))
    fm = fm.cuda()
    batches = [(torch.from_numpy(g["dice_inputs"][i: i + 1]), torch.zeros(1, dtype=torch.long)) for i in range(g["dice_inputs"].shape[0])]
    mean, thr = get_dice_feat_mean_react_percentile(fm, batches, 90)
    assert rel_err(mean, g["dice_mean"]) < 1e-5 and abs(float(thr) - float(g["dice_thr"])) < 1e-5 * max(1.0, abs(float(g["dice_thr"])))


@pytest.mark.gpu
def test_route_dice_forward_and_long_rows():
    """RouteDICE.forward against its definition in NumPy (the reference's forward calls .cuda() and cannot run in the
    build container; its arithmetic is pinned through the DICE postprocessor's all-baselines golden), and the
    radix-select ASH-S kernel on rows longer than the register kernel takes (ties at the threshold included)."""
    from runia_core_amd.inference import RouteDICE, ash_s_conv_layer, ash_s_linear_layer

    rng = np.random.default_rng(5)
    info = np.abs(rng.standard_normal(64)).astype(np.float32)
    layer = RouteDICE(64, 10, bias=True, p=90, info=info)
    x = rng.standard_normal((33, 64)).astype(np.float32)
    out = layer(torch.from_numpy(x))
    w, b = layer.weight.detach().numpy(), layer.bias.detach().numpy()
    contrib = info[None, :] * w
    masked = w * (contrib > np.percentile(contrib, 90))
    assert out.is_cuda and rel_err(out.cpu().numpy(), x @ masked.T + b) < 1e-5
    assert np.allclose(layer.thresh, np.percentile(contrib, 90)) and np.array_equal(layer.masked_w.detach().cpu().numpy(), masked.astype(np.float32))
    # long rows: 5 000 features (linear form) and 2 x 128 x 7 x 7 = 6 272 per sample (conv form), with exact ties
    xl = np.abs(rng.standard_normal((9, 5000))).astype(np.float32) + 0.01
    xl[:, 100:140] = xl[:, 99:100]  # 41 equal values
    assert rel_err(ash_s_linear_layer(xl, 85), oracle.ash_s_defined(xl, 85)) < 1e-5
    xc = np.abs(rng.standard_normal((2, 128, 7, 7))).astype(np.float32) + 0.01
    flat = xc.reshape(2, -1)
    k = flat.shape[1] - int(np.round(flat.shape[1] * 65 / 100.0))
    kept = np.zeros_like(flat)
    idx = np.argsort(-flat, axis=1, kind="stable")[:, :k]
    np.put_along_axis(kept, idx, np.take_along_axis(flat, idx, axis=1), axis=1)
    want = kept * np.exp(flat.sum(1) / kept.sum(1))[:, None]
    got = ash_s_conv_layer(torch.from_numpy(xc.copy()), 65)
    assert rel_err(got.numpy().reshape(2, -1), want) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("osz,sr,n_mc,c", [(7, 2, 16, 256), (7, 2, 32, 100), (7, 1, 12, 64), (4, 2, 16, 300), (8, 2, 32, 65)])
def test_roi_align_folded_into_the_sampler_equals_the_two_calls(osz, sr, n_mc, c):
    """runia_roi_mc_entropy_f32 (round 4: roi_align folded into the load of the fused sampler + entropy kernel; channels-last
    feature map, per-ROI table of bilinear samples read through the scalar cache) against runia_roi_align_f32 followed by
    runia_mc_entropy_f32: the SAME BITS - entropies and MC samples - for boxes inside, across and beyond the map's
    border (samples more than a pixel outside contribute 0), two images in the batch, ragged channel counts, 12 / 16 /
    32 drop layers, one and four samples per bin."""
    from runia_core_amd import _hip

    rng = np.random.default_rng(osz * 100 + sr * 10 + n_mc)
    b, hh, ww = 2, 23, 37
    fm = torch.relu(torch.from_numpy(rng.standard_normal((b, c, hh, ww)).astype(np.float32))).cuda()
    k = 90
    xy = rng.uniform(-40, 560, size=(k, 2)).astype(np.float32)
    wh = rng.uniform(2, 300, size=(k, 2)).astype(np.float32)
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], axis=1))
    boxes[0] = torch.tensor([-500.0, -500.0, -400.0, -300.0])   # wholly outside: every sample contributes 0
    boxes[1] = torch.tensor([0.0, 0.0, 592.0, 368.0])           # the whole map
    bidx = torch.from_numpy(rng.integers(0, b, size=k).astype(np.int32))
    rand = torch.from_numpy(rng.random((k, n_mc, osz, osz)).astype(np.float32)).cuda()
    scale = ww / 592.0
    kk = 5
    rois = _hip.roi_align(fm, boxes.cuda(), osz, scale, sr, True, bidx)
    h_ref, z_ref = _hip.mc_entropy(rois, rand, n_mc, 0.4, 2, kk, 1e-5, want_samples=True)
    h, z = _hip.roi_mc_entropy(_hip.nchw_to_nhwc(fm), boxes, osz, scale, sr, True, rand, n_mc, 0.4, 2, kk, 1e-5, batch_idx=bidx,
                               return_samples=True)
    assert torch.equal(_hip.nchw_to_nhwc(fm), fm.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(z, z_ref)
    assert torch.equal(torch.nan_to_num(h, nan=-7.0), torch.nan_to_num(h_ref, nan=-7.0))
    assert bool(torch.isfinite(h[2:]).any())


@pytest.mark.gpu
def test_every_roi_shape_the_query_accepts_has_a_kernel_and_the_others_take_the_two_calls():
    """runia_roi_mc_entropy_supported promised shapes the entry point had no kernel for (output 4 / 8
    with one sample per bin, 4x4 with 5..8 drop layers).  The query and the dispatch now expand ONE shape list: every
    (PH, n_mc, sampling_ratio) the query accepts launches and returns the two-call bits, and
    _dropblock_rois_get_entropy on a shape it refuses still returns the two-call result."""
    from runia_core_amd import MCSamplerModule, _hip
    from runia_core_amd.feature_extraction.object_level import _dropblock_rois_get_entropy

    rng = np.random.default_rng(5)
    fm = torch.relu(torch.from_numpy(rng.standard_normal((1, 64, 12, 17)).astype(np.float32))).cuda()
    nhwc = _hip.nchw_to_nhwc(fm)
    xy = rng.uniform(-10, 200, size=(6, 2)).astype(np.float32)
    boxes = torch.from_numpy(np.concatenate([xy, xy + rng.uniform(8, 120, size=(6, 2)).astype(np.float32)], axis=1))
    accepted = 0
    for osz in (2, 4, 7, 8):
        for sr in (-1, 0, 1, 2, 3):
            for n_mc in range(5, 33):
                ok = _hip.roi_mc_entropy_supported(osz, osz, n_mc, 5, sr)
                if not ok:
                    continue
                accepted += 1
                rand = torch.from_numpy(rng.random((6, n_mc, osz, osz)).astype(np.float32)).cuda()
                h = _hip.roi_mc_entropy(nhwc, boxes, osz, 17 / 272.0, sr, True, rand, n_mc, 0.3, 2, 5)   # raised RuniaHipError before
                if n_mc in (9, 16, 17, 32):
                    rois = _hip.roi_align(fm, boxes.cuda(), osz, 17 / 272.0, sr, True)
                    h_ref = _hip.mc_entropy(rois, rand, n_mc, 0.3, 2, 5)
                    assert torch.equal(torch.nan_to_num(h, nan=-7.0), torch.nan_to_num(h_ref, nan=-7.0)), (osz, sr, n_mc)
    assert accepted == 8 * 12  # 8 listed (shape, samples-per-bin) pairs x n_mc 9..16 / 17..32 -> 24 values each / 2 register sizes
    for osz, sr, n_mc in ((4, 1, 16), (8, 1, 16), (4, 2, 6), (7, 0, 16)):
        assert not _hip.roi_mc_entropy_supported(osz, osz, n_mc, 5, sr)
    # the reference-shaped caller on a shape without a fused kernel: the two calls, as before round 4
    for osz, sr, n_mc in ((4, 1, 16), (8, 1, 12)):
        sampler = MCSamplerModule(mc_samples=n_mc, block_size=2, drop_prob=0.3).train().use_counter_draws(seed=3)
        ent = _dropblock_rois_get_entropy([fm], (osz,), boxes, (192, 272), sr, 1, n_mc, sampler)
        rois = _hip.roi_align(fm, boxes.cuda(), osz, 17 / 272.0, sr, True)
        h_ref = _hip.mc_entropy(rois, _hip.CounterDraws(3, 0), n_mc, 0.3, 2, 5).to(torch.float32).cpu()
        assert torch.equal(torch.nan_to_num(ent, nan=-7.0), torch.nan_to_num(h_ref, nan=-7.0))


@pytest.mark.gpu
def test_roi_align_folded_into_the_sampler_non_finite_pixels_and_degenerate_boxes():
    """The fused ROI launch against the two launches where the arithmetic leaves the finite range: NaN and infinite
    activations in the map (a sample that touches one is NaN / infinite in both forms; a sample OUTSIDE the map must stay
    0 even next to them: its taps are not read), boxes of zero width / height, a box that is one point, boxes whose two
    pixel rows coincide at the map's last row (the loader keeps pixel rows in registers across sample rows).  MC samples
    compared as bit patterns."""
    from runia_core_amd import _hip

    rng = np.random.default_rng(77)
    b, c, hh, ww = 2, 70, 9, 13
    fm = torch.relu(torch.from_numpy(rng.standard_normal((b, c, hh, ww)).astype(np.float32)))
    fm[0, :, 0, 0] = float("inf")       # the pixel at byte offset 0 (what an unmasked outside tap would have read)
    fm[0, 3, 4, 5] = float("nan")
    fm[1, :, hh - 1, ww - 1] = float("-inf")
    fm = fm.cuda()
    boxes = torch.tensor([[-300.0, -300.0, -200.0, -250.0],    # wholly outside, image 0 (pixel 0 of that image is inf)
                          [-20.0, -20.0, 30.0, 30.0],          # across the corner with the infinite pixel
                          [40.0, 40.0, 40.0, 90.0],            # zero width
                          [40.0, 40.0, 90.0, 40.0],            # zero height
                          [55.5, 33.25, 55.5, 33.25],          # a point
                          [0.0, 120.0, 200.0, 160.0],          # hangs over the last rows: clamped rows coincide
                          [150.0, 100.0, 260.0, 190.0],        # over the last row / column of image 1 (-inf pixel)
                          [10.0, 10.0, 120.0, 100.0]], dtype=torch.float32)
    bidx = torch.tensor([0, 0, 0, 1, 1, 0, 1, 1], dtype=torch.int32)
    n_mc, osz = 16, 7
    rand = torch.from_numpy(rng.random((len(boxes), n_mc, osz, osz)).astype(np.float32)).cuda()
    scale = ww / 208.0
    rois = _hip.roi_align(fm, boxes.cuda(), osz, scale, 2, True, bidx)
    h_ref, z_ref = _hip.mc_entropy(rois, rand, n_mc, 0.3, 2, 5, 1e-5, want_samples=True)
    h, z = _hip.roi_mc_entropy(_hip.nchw_to_nhwc(fm), boxes, osz, scale, 2, True, rand, n_mc, 0.3, 2, 5, 1e-5, batch_idx=bidx,
                               return_samples=True)
    assert torch.equal(z.view(torch.int32), z_ref.view(torch.int32))
    assert torch.equal(torch.nan_to_num(h, nan=-7.0, posinf=-8.0, neginf=-9.0), torch.nan_to_num(h_ref, nan=-7.0, posinf=-8.0, neginf=-9.0))
    assert bool((rois[0] == 0).all())                      # outside the map: zeros, not inf * 0
    assert bool(torch.isfinite(z[0]).all()) and bool(torch.isfinite(z[7]).any())


@pytest.mark.gpu
@pytest.mark.parametrize("d", [2, 8, 11, 16, 23])
def test_lared_row_scores_do_not_depend_on_the_batch(d):
    """DetectorKDE picked its algorithm by batch size (direct kernel above 2 048 rows at D < 12, above 16 384 at
    D < 24), so a row's bits followed the batch it arrived in and sharded != unsharded.  One algorithm now: the same row
    scores the same bits alone, in a 2 048-row call, a 2 049-row call, a 20 000-row call and in the shards of a 3-rank cut;
    and the exact definition (oracle) on top."""
    from runia_core_amd.distributed import shard_bounds
    from runia_core_amd.inference.postprocessors import KDELatentSpace

    rng = np.random.default_rng(d)
    train = rng.standard_normal((3000, d)) + 0.5
    x = rng.standard_normal((20000, d)) * 1.3
    kde = KDELatentSpace()
    kde.setup(train)
    whole = kde.postprocess(x)
    for a, b in ((0, 1), (5, 2053), (5, 2054), (0, 16385), (17000, 20000)):
        assert np.array_equal(kde.postprocess(x[a:b]), whole[a:b]), (a, b)
    parts = [kde.postprocess(x[slice(*shard_bounds(len(x), 3, r))]) for r in range(3)]
    assert np.array_equal(np.concatenate(parts), whole)
    rows = [0, 2048, 2049, 19999]
    assert rel_err(whole[rows], oracle.kde_score(train, x[rows])) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["tophat", "epanechnikov", "exponential", "linear", "cosine", "gaussian"])
def test_detector_kde_other_kernels_vs_sklearn(kernel):
    """DetectorKDE(kernel=...) forwards any sklearn kernel (reference inference/postprocessors.py:78-128): the direct kernel
    against sklearn's KernelDensity itself (the call the reference makes) in the low dimensions where its tree is converged -
    D = 3 and 8, bandwidths that leave some queries without a training row in range."""
    from sklearn.neighbors import KernelDensity

    from runia_core_amd.inference.postprocessors import DetectorKDE

    rng = np.random.default_rng(11)
    for d, h in ((3, 0.6), (8, 2.5), (8, 1.2)):
        train = rng.standard_normal((700, d))
        x = np.concatenate([rng.standard_normal((90, d)), rng.standard_normal((10, d)) * 6.0])
        ref = KernelDensity(kernel=kernel, bandwidth=h).fit(train).score_samples(x)
        got = DetectorKDE(train, kernel=kernel, bandwidth=h).get_density_scores(x)
        dist = np.sqrt(((x[:, None, :] - train[None]) ** 2).sum(-1))
        reach = (dist < h).any(axis=1) if kernel not in ("gaussian", "exponential") else np.ones(len(x), dtype=bool)
        assert reach.sum() >= 20
        # (sklearn's cosine normalisation is the log of an alternating sum that is negative at d = 8: NaN there, NaN here)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        reach &= ~np.isnan(ref)
        reach &= ~(ref < -36.0)  # (densities below ~e^-36 of the peak: sklearn's tree is at its bound residue, e.g. -208.3 for -194.1)
        assert rel_err(got[reach], ref[reach]) < 1e-5, (kernel, d, h)
        # no training row within the bandwidth of a compact kernel: the density is 0, log = -inf; sklearn's tree returns
        # the rounding residue of its log-space bounds there (about -39 for these sizes) - the same artefact as its
        # gaussian scores above D ~ 20 (known divergences)
        if kernel not in ("gaussian", "exponential"):
            gone = ~(dist < h).any(axis=1) & ~np.isnan(ref)
            assert np.all(got[gone] == -np.inf) and np.all(ref[gone] < -25.0)
    with pytest.raises(ValueError):
        DetectorKDE(train, kernel="triangular")


def test_gmm_fit_on_the_device_equals_the_host_fit():
    """Round 6: gmm_fit's moments and float32 factorisations on the device (device_fit.gmm_fit_device: runia_covariance_f32in per
    class + runia_cholesky_f32 under the reference's jitter ladder) against the host fit in float32 torch (the reference's own
    arithmetic, inference/funcs.py:265-344): same present classes, means and factors at 1e-6 / 1e-5, same jitter - on the
    reference-run fixture, on 512-d rows with an empty class and a single-row class, and on rank-deficient class covariances
    (the ladder has to move: the jitter chosen may differ by one step where a pivot is within rounding of zero)."""
    from runia_core_amd import config
    from runia_core_amd.inference import GMMLatentSpace, gmm_fit

    g = load_npz("ref_f4.npz")
    rng = np.random.default_rng(5)
    big = (rng.standard_normal((6000, 512)) * (0.5 + rng.random(512))).astype(np.float32)
    big_lab = rng.integers(0, 9, 6000)
    big_lab[big_lab == 4] = 3            # class 4 is empty
    big_lab[0] = 9                       # class 9 has a single row: zero covariance -> the ladder moves
    cases = [(g["gmm_train"], g["gmm_labels"], 10, True), (big[:, :40], rng.integers(0, 5, 6000), 5, True), (big, big_lab, 10, False)]
    low = (rng.standard_normal((300, 4)) @ rng.standard_normal((4, 48))).astype(np.float32)   # rank 4 in 48 dims
    cases.append((low, rng.integers(0, 3, 300), 3, False))
    before = config.device_fit
    try:
        for x, lab, c, same_jitter in cases:
            config.device_fit = False
            gh, jh = gmm_fit(torch.from_numpy(x), torch.from_numpy(np.asarray(lab)), c)
            config.device_fit = True
            gd, jd = gmm_fit(torch.from_numpy(x), torch.from_numpy(np.asarray(lab)), c)
            assert hasattr(gd, "_runia_device_params") and not hasattr(gh, "_runia_device_params")
            assert gd.loc.shape == gh.loc.shape and gd.scale_tril.shape == gh.scale_tril.shape
            assert np.allclose(gd.loc.numpy(), gh.loc.numpy(), atol=1e-6 * max(1.0, float(gh.loc.abs().max())))
            if same_jitter:
                assert jd == jh
                assert np.allclose(gd.scale_tril.numpy(), gh.scale_tril.numpy(), atol=1e-5 * max(1.0, float(gh.scale_tril.abs().max())))
            else:
                ladder = [0.0] + [10.0 ** e for e in range(-20, 0)]
                assert abs(ladder.index(float(jd)) - ladder.index(float(jh))) <= 1
            # both fits give a usable distribution: finite densities on the training rows
            for gm in (gd, gh):
                assert bool(torch.isfinite(gm.log_prob(torch.from_numpy(x[:50])[:, None, :])).all())
        # the postprocessor scores through the device fit equal the reference-run fixture (also covered under both fits by
        # test_gmm_and_ddu_fixtures)
        config.device_fit = True
        p = GMMLatentSpace()
        p.setup(g["gmm_train"], ind_train_labels=g["gmm_labels"])
        assert rel_err(p.postprocess(g["gmm_test"]), g["gmm_scores"]) < 1e-5
    finally:
        config.device_fit = before


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_setup_device_equals_setup_for_the_latent_space_family(dtype):
    """Round 6 (the harness's device-resident sweep): setup_device on training rows that already sit in HBM gives the fitted state
    and the scores of setup on the same rows as host arrays - KDE, MD, cMD, KNN, GMM; public attributes the reference exposes
    (centered_data, activation_log, feats_mean, precision, class_mean, detector) stay readable; the objects pickle."""
    import pickle

    from runia_core_amd.inference.postprocessors import postprocessors_dict

    rng = np.random.default_rng(31)
    n, d, c = 3000, 48, 5
    lab = rng.integers(0, c, n)
    centres = rng.standard_normal((c, d))
    tr = (centres[lab] + rng.standard_normal((n, d)) * (0.5 + rng.random(d))).astype(dtype)
    te = (centres[rng.integers(0, c, 400)] + rng.standard_normal((400, d))).astype(dtype)
    tr_d, te_d = torch.from_numpy(tr).cuda(), torch.from_numpy(te).cuda()

    class Cfg:
        num_classes, k_neighbors = c, 7

    tol = {"KDE": 1e-12, "MD": 1e-10, "cMD": 2e-6, "KNN": 1e-6, "GMM": 1e-5}
    if dtype == np.float32:   # float32 rows: means / norms formed in float32 by the host calls, in f64 then rounded on the device
        tol.update(MD=2e-6, KNN=2e-6)
    for name in ("KDE", "MD", "cMD", "KNN", "GMM"):
        a, b = postprocessors_dict[name](cfg=Cfg()), postprocessors_dict[name](cfg=Cfg())
        a.setup(tr, ind_train_labels=lab)
        b.setup_device(tr_d, ind_train_labels=lab)
        assert b._setup_flag
        sa = a.postprocess(te, pred_labels=np.zeros(400, int))
        sb = b.postprocess_device(te_d).cpu().numpy()
        assert sa.dtype == sb.dtype and rel_err(sb, sa) < tol[name], (name, rel_err(sb, sa))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            b.setup_device(tr_d, ind_train_labels=lab)
            assert len(w) == 1 and "already trained" in str(w[0].message)
        b2 = pickle.loads(pickle.dumps(b))
        assert np.array_equal(b2.postprocess_device(te_d).cpu().numpy(), sb)   # device copies rebuilt from the pickled host state
        f32 = dtype == np.float32
        if name == "MD":
            assert b.feats_mean.shape == (1, d) and b.feats_mean.dtype == a.feats_mean.dtype
            assert np.allclose(b.feats_mean, a.feats_mean, atol=2e-6 if f32 else 1e-13)
            assert np.allclose(b.precision, a.precision, rtol=1e-4 if f32 else 1e-9, atol=1e-5 if f32 else 1e-12)
            assert b.centered_data.shape == (n, d) and np.allclose(b.centered_data, a.centered_data, atol=2e-6 if f32 else 1e-12)
        if name == "KNN":
            assert b.K == 7 and b.index.ntotal == n
            assert b.activation_log.dtype == a.activation_log.dtype
            assert np.allclose(b.activation_log, a.activation_log, atol=2e-7 if f32 else 1e-15)
        if name == "cMD":
            assert np.allclose(b.class_mean.numpy(), a.class_mean.numpy(), atol=1e-6) and b.precision.dtype == torch.float32
        if name == "GMM":
            assert np.allclose(b.gmm.loc.numpy(), a.gmm.loc.numpy(), atol=1e-6)
    with pytest.raises(ValueError, match="id_labels not provided"):
        postprocessors_dict["cMD"](cfg=Cfg()).setup_device(tr_d)
    with pytest.raises(ValueError, match="id_labels not provided"):
        postprocessors_dict["GMM"](cfg=Cfg()).setup_device(tr_d)


@pytest.mark.parametrize("n_ind,n_ood", [(1, 1), (31, 33), (40, 25), (600, 424), (2100, 1996), (30_000, 35_536), (30_000, 35_537),
                                         (131_072, 131_072), (131_073, 131_072), (300_000, 200_000)])
@pytest.mark.parametrize("kind", ["normal", "squashed_ties", "unit_ties"])
def test_fused_metrics_launches_equal_the_curve_chain(n_ind, n_ood, kind):
    """Round 6: the three scalars come from six launches (four without the sketch) - buckets by splitter KEYS, bucket sort and
    curve terms in one launch - while the curve API keeps the eight-launch chain (it needs every run's counts in memory).  Both
    against each other and the oracle across the sizes where the new path changes shape: bucket counts 64 ... 4 096, the small-tile
    launches up to 65 536 scores, the sketch + splitter launches beyond 262 144; float32 scores whose sigmoids tie across many
    buckets (the squash merges raw scores the raw bins keep apart) and scores inside [0, 1] with ties."""
    from runia_core_amd import _hip

    rng = np.random.default_rng(n_ind + 7 * n_ood)
    if kind == "normal":
        ind, ood = rng.standard_normal(n_ind) * 2 + 0.5, rng.standard_normal(n_ood) * 2 - 0.5
    elif kind == "squashed_ties":   # float32 energies around 9: sigmoids within 1e-4 of 1
        ind, ood = (rng.standard_normal(n_ind) * 2 + 9).astype(np.float32), (rng.standard_normal(n_ood) * 2 + 7).astype(np.float32)
    else:
        ind, ood = rng.integers(0, 50, n_ind) / 49.0, rng.integers(0, 40, n_ood) / 49.0
    a, b = torch.from_numpy(np.ascontiguousarray(ind)).cuda(), torch.from_numpy(np.ascontiguousarray(ood)).cuda()
    fused = _hip.ood_metrics(a, b).cpu().numpy()
    chain = _hip.ood_clf_curve(a, b)[0].cpu().numpy()
    assert fused == pytest.approx(chain, abs=2e-7, nan_ok=True), (fused, chain)
    exp = oracle.auroc_fpr95_aupr(ind, ood)
    assert tuple(fused) == pytest.approx(exp, abs=2e-6, nan_ok=True)
    assert np.array_equal(_hip.ood_metrics(a, b).cpu().numpy(), fused, equal_nan=True)   # same bits from run to run


def test_subspace_basis_reconditioning_and_its_fallback():
    """device_fit._orthonormalise (round 6): y L^-T with the Cholesky factor of y^T y - columns orthonormal, same span; a Gram
    matrix without a usable factor (duplicated columns: rank-deficient) takes the symmetric inverse square root, which drops the
    directions below the numerical rank instead of dividing by a rounding-sized pivot."""
    from runia_core_amd.device_fit import _orthonormalise

    rng = np.random.default_rng(3)
    y = rng.standard_normal((512, 74)) * np.logspace(0, -4, 74)[None, :]  # columns over four orders of magnitude
    q = _orthonormalise(torch.from_numpy(y).cuda()).cpu().numpy()
    assert np.max(np.abs(q.T @ q - np.eye(74))) < 1e-7
    proj = q @ q.T
    assert np.max(np.abs(proj @ y - y)) < 1e-9 * np.abs(y).max()  # same column span
    yd = y.copy()
    yd[:, 10] = yd[:, 3]          # exact duplicate: y^T y is singular
    yd[:, 20] = 0.0               # and a zero column
    qd = _orthonormalise(torch.from_numpy(yd).cuda()).cpu().numpy()
    assert np.isfinite(qd).all()
    g = qd.T @ qd                 # a projector of rank 72, not the identity
    assert np.max(np.abs(g @ g - g)) < 1e-8 and abs(np.trace(g) - 72.0) < 1e-6


def test_pinvh_device_routes():
    """device_fit.pinvh_device (round 6): the Cholesky route for matrices that provably lose no direction to SciPy's cut-off, the
    eigen route for everything else - both equal scipy.linalg.pinvh."""
    from scipy.linalg import pinvh

    from runia_core_amd import _hip
    from runia_core_amd import device_fit

    rng = np.random.default_rng(9)
    n = 300

    def run(a):
        calls = {"eigh": 0}
        real = _hip.eigh

        def counting(*args, **kw):
            calls["eigh"] += 1
            return real(*args, **kw)

        _hip.eigh = counting
        try:
            out = device_fit.pinvh_device(torch.from_numpy(a).cuda()).cpu().numpy()
        finally:
            _hip.eigh = real
        return out, calls["eigh"]

    x = rng.standard_normal((n, 3 * n))
    good = x @ x.T / (3 * n) + 0.1 * np.eye(n)                      # condition ~ 30
    got, used = run(good)
    assert used == 0 and rel_err(got, pinvh(good)) < 1e-12 and np.array_equal(got, got.T)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    stiff = (q * np.logspace(0, -10, n)) @ q.T                       # positive definite, condition 1e10: beyond the bound
    stiff = (stiff + stiff.T) / 2
    got, used = run(stiff)
    assert used == 1 and rel_err(got @ stiff, np.eye(n)) < 1e-4
    low = x[:, :40] @ x[:, :40].T                                    # rank 40: SciPy drops 260 directions
    got, used = run(low)
    ref = pinvh(low)
    assert used == 1 and np.max(np.abs(got - ref)) < 1e-9 * np.abs(ref).max()
    small = good[:64, :64].copy()                                    # below the size threshold: the eigen route
    got, used = run(small)
    assert used == 1 and rel_err(got, pinvh(small)) < 1e-11
