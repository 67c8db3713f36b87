#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own hot-path source files.

Runs ONLY in the build container (needs /root/reference).  The reference package
cannot be imported as a package (``dropblock`` etc. are absent), so its hot-path
modules are imported *by path* under bare namespace packages (package
``__init__`` files never run) with inert placeholders for the absent optional
imports that those modules name at import time but never call on this path
(omegaconf.DictConfig as a type annotation, faiss, dropblock, pacmap).  See
SURVEY.md section 8c / Appendix B.

Only DATA is written: seeded inputs, fitted state as plain arrays, and the
scores the reference returned.  No reference source travels.

Usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tools/make_goldens.py
"""
from __future__ import annotations

import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _load_reference():
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    ns("runia_core", f"{REF}/runia_core")
    ns("runia_core.inference", f"{REF}/runia_core/inference")
    ns("runia_core.feature_extraction", f"{REF}/runia_core/feature_extraction")
    ns("runia_core.evaluation", f"{REF}/runia_core/evaluation")

    om = types.ModuleType("omegaconf")
    om.DictConfig = dict
    sys.modules["omegaconf"] = om
    sys.modules["faiss"] = types.ModuleType("faiss")
    db = types.ModuleType("dropblock")

    class DropBlock2D(torch.nn.Module):  # placeholder type, never called
        pass

    db.DropBlock2D = DropBlock2D
    sys.modules["dropblock"] = db
    pm = types.ModuleType("pacmap")
    pm.PaCMAP = type("PaCMAP", (), {})
    sys.modules["pacmap"] = pm

    import runia_core.inference.postprocessors as pp  # noqa: E402
    import runia_core.inference.funcs as funcs  # noqa: E402
    import runia_core.inference.abstract_classes as ac  # noqa: E402
    import runia_core.dimensionality_reduction as dr  # noqa: E402

    return pp, funcs, ac, dr


def generate_test_data(num_samples=10, feature_dim=32, num_classes=10, seed=42):
    """Same seeded recipe as /root/reference/tests/unit_test_postprocessors.py:66-100
    (re-derived here so that fixtures carry the exact inputs)."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    features = np.random.randn(num_samples, feature_dim).astype(np.float32)
    labels = np.random.randint(0, num_classes, num_samples)
    for i in range(num_classes):
        m = labels == i
        if np.any(m):
            features[m] += np.random.randn(feature_dim) * 0.5
    logits = np.random.randn(num_samples, num_classes).astype(np.float32)
    return features, labels, logits


def main():
    pp, funcs, ac, dr = _load_reference()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)

    # ---------------- MD / LaREM ------------------------------------------------
    cases = {}
    # (1) the reference's own unit-test inputs (10x32, singular covariance -> pinvh)
    tr, _, _ = generate_test_data(seed=42)
    te, _, _ = generate_test_data(seed=43)
    # (2) 200x20 uniform (unit_test_baselines.py:463-530 recipe)
    np.random.seed(1)
    tf = np.random.rand(200, 20)
    # (3) realistic LaREM size: PCA-256-like whitened features
    tr3 = rng.standard_normal((2000, 256))
    te3 = rng.standard_normal((300, 256)) * 1.3 + 0.2
    for name, (a, b) in {"unit": (tr, te), "baselines": (tf, tf), "d256": (tr3, te3)}.items():
        p = pp.MDLatentSpace()
        p.setup(a)
        s = p.postprocess(b)
        if a.size <= 8192:  # keep fixtures small: big cases carry fitted state only
            cases[f"{name}_train"] = a
        cases[f"{name}_test"] = b
        cases[f"{name}_mean"] = p.feats_mean
        cases[f"{name}_precision"] = p.precision
        cases[f"{name}_scores"] = s
    np.savez_compressed(os.path.join(OUT, "ref_md.npz"), **cases)

    # ---------------- KDE / LaRED ----------------------------------------------
    cases = {}
    for name, (a, b) in {"unit": (tr, te), "baselines": (tf, tf[:50])}.items():
        p = pp.KDELatentSpace()
        p.setup(a)
        cases[f"{name}_train"] = a
        cases[f"{name}_test"] = b
        cases[f"{name}_scores"] = p.postprocess(b)
    a = rng.standard_normal((700, 12))
    b = rng.standard_normal((90, 12)) * 1.2
    p = pp.KDELatentSpace()
    p.setup(a)
    cases["d12_train"], cases["d12_test"], cases["d12_scores"] = a, b, p.postprocess(b)
    # Reference quirk (recorded): sklearn's KD-tree KDE tracks its
    # bounds in log space and, for D >~ 20, returns the floating-point residue of
    # the root-node bound instead of the density (error grows to hundreds of nats).
    a = rng.standard_normal((300, 64))
    b = rng.standard_normal((40, 64)) * 1.2
    p = pp.KDELatentSpace()
    p.setup(a)
    cases["quirk64_train"], cases["quirk64_test"], cases["quirk64_scores"] = a, b, p.postprocess(b)
    np.savez_compressed(os.path.join(OUT, "ref_kde.npz"), **cases)

    # ---------------- Mahalanobis (class-conditional) ---------------------------
    cases = {}
    trf, trl, _ = generate_test_data(seed=42)
    vaf, _, _ = generate_test_data(seed=44)
    tef, _, _ = generate_test_data(seed=43)
    p = pp.Mahalanobis(flip_sign=True, num_classes=10)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p.setup(trf, train_labels=trl, valid_feats=vaf)
        cases["unit_train"], cases["unit_labels"], cases["unit_valid"], cases["unit_test"] = trf, trl, vaf, tef
        cases["unit_class_mean"], cases["unit_precision"] = p.class_mean, p.precision
        cases["unit_scores"] = p.postprocess(tef)
        cases["unit_threshold"] = np.array(p.threshold)
    # well-conditioned case, f32 features, 7 classes, D=96
    C, D = 7, 96
    centres = rng.standard_normal((C, D)).astype(np.float32)
    lab = rng.integers(0, C, 1500)
    f = (centres[lab] + rng.standard_normal((1500, D))).astype(np.float32)
    lab_t = rng.integers(0, C, 200)
    ft = (centres[lab_t] * 0.7 + 1.2 * rng.standard_normal((200, D))).astype(np.float32)
    p = pp.Mahalanobis(flip_sign=False, num_classes=C)
    p.setup(f, train_labels=lab, valid_feats=f[:100])
    cases["d96_train"], cases["d96_labels"], cases["d96_test"] = f, lab, ft
    cases["d96_class_mean"], cases["d96_precision"] = p.class_mean, p.precision
    cases["d96_scores"] = p.postprocess(ft)
    cases["d96_threshold"] = np.array(p.threshold)
    # f64 features path (diff stays f64)
    p = pp.Mahalanobis(flip_sign=False, num_classes=C)
    p.setup(f.astype(np.float64), train_labels=lab, valid_feats=f[:100].astype(np.float64))
    cases["d96_f64_class_mean"], cases["d96_f64_precision"] = p.class_mean, p.precision
    cases["d96_f64_scores"] = p.postprocess(ft.astype(np.float64))
    np.savez_compressed(os.path.join(OUT, "ref_mahalanobis.npz"), **cases)

    # ---------------- Energy / MSP ----------------------------------------------
    cases = {}
    _, _, trl_ = generate_test_data(seed=42)
    _, _, tel_ = generate_test_data(seed=43)
    for cls, nm in ((pp.Energy, "energy"), (pp.MSP, "msp")):
        p = cls(flip_sign=True)
        p.setup(trl_)
        cases[f"unit_{nm}_scores"] = p.postprocess(tel_)
        cases[f"unit_{nm}_threshold"] = np.array(p.threshold)
    cases["unit_train_logits"], cases["unit_test_logits"] = trl_, tel_
    big = (rng.standard_normal((257, 1000)) * 3.0).astype(np.float32)
    big[5, 17] = 80.0  # large-dynamic-range row
    big[6, :] = -50.0  # constant row
    small = rng.standard_normal((513, 10)).astype(np.float32)
    for nm, arr in (("c1000", big), ("c10", small)):
        for cls, sn in ((pp.Energy, "energy"), (pp.MSP, "msp")):
            p = cls(flip_sign=False)
            p.setup(arr[:64])
            cases[f"{nm}_{sn}_scores"] = p.postprocess(arr)
            cases[f"{nm}_{sn}_threshold"] = np.array(p.threshold)
        cases[f"{nm}_logits"] = arr
    np.savez_compressed(os.path.join(OUT, "ref_energy_msp.npz"), **cases)

    # ---------------- PCA (fit with sklearn through the reference helper) -------
    cases = {}
    np.random.seed(1)
    ind = 0.5 + np.random.randn(1000, 20)
    ood = -0.5 + np.random.randn(1000, 20)
    tr_t, pca = dr.apply_pca_ds_split(ind, 10)
    cases["unit_ind"], cases["unit_ood"] = ind, ood
    cases["unit_components"], cases["unit_mean"], cases["unit_var"] = (
        pca.components_,
        pca.mean_,
        pca.explained_variance_,
    )
    cases["unit_train_transformed"] = tr_t
    cases["unit_ood_transformed"] = dr.apply_pca_transform(ood, pca)
    np.random.seed(7)
    x = rng.standard_normal((1500, 512)) * (0.2 + rng.random(512)) + rng.standard_normal(512)
    xt = rng.standard_normal((100, 512)) * (0.2 + rng.random(512))
    tr_t, pca = dr.apply_pca_ds_split(x, 256)
    cases["d512_components"], cases["d512_mean"], cases["d512_var"] = (
        pca.components_,
        pca.mean_,
        pca.explained_variance_,
    )
    cases["d512_test"] = xt
    cases["d512_test_transformed"] = dr.apply_pca_transform(xt, pca)
    cases["d512_test_f32_transformed"] = dr.apply_pca_transform(xt.astype(np.float32), pca)
    np.savez_compressed(os.path.join(OUT, "ref_pca.npz"), **cases)

    # ---------------- f4: GEN / ASH / ReAct (DICE needs .cuda() in the reference: pinned by its test golden) ----
    cases = {}
    C, D = 37, 300
    w = (rng.standard_normal((C, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    tr = np.maximum(rng.standard_normal((400, D)), 0).astype(np.float32)
    va = np.maximum(rng.standard_normal((64, D)), 0).astype(np.float32)
    te = np.maximum(rng.standard_normal((150, D)) * 1.3, 0).astype(np.float32)
    fc = {"weight": w, "bias": b}
    cases.update(w=w, b=b, train=tr, valid=va, test=te)
    for pct in (90, 65):
        p = pp.ASH(flip_sign=False, ash_percentile=pct)
        p.setup(tr, valid_feats=va, final_linear_layer_params=fc)
        cases[f"ash{pct}_scores"], cases[f"ash{pct}_threshold"] = p.postprocess(te), np.array(p.threshold)
        cases[f"ash{pct}_transformed"] = funcs.ash_s_linear_layer(te.copy(), pct)
    p = pp.ReAct(flip_sign=False, react_percentile=90)
    p.setup(tr, valid_feats=va, final_linear_layer_params=fc)
    cases["react_scores"], cases["react_threshold"], cases["react_clip"] = p.postprocess(te), np.array(p.threshold), np.array(p.activation_threshold)
    logits_tr = (tr @ w.T + b).astype(np.float32)
    logits_te = (te @ w.T + b).astype(np.float32)
    cases.update(logits_train=logits_tr, logits_test=logits_te)
    for M in (37, 10, 100):
        p = pp.GEN(flip_sign=False, gamma=0.1, num_classes=M)
        p.setup(logits_tr)
        cases[f"gen{M}_scores"], cases[f"gen{M}_threshold"] = p.postprocess(logits_te), np.array(p.threshold)
    logits_va = (va @ w.T + b).astype(np.float32)
    cases["logits_valid"] = logits_va
    p = pp.ViM(flip_sign=False)
    p.setup(tr, final_linear_layer_params=fc, train_logits=logits_tr, valid_feats=va, valid_logits=logits_va)
    cases["vim_u"], cases["vim_NS"], cases["vim_alpha"] = p.u, p.NS, np.array(p.alpha)
    cases["vim_scores"], cases["vim_threshold"] = p.postprocess(te, logits=logits_te), np.array(p.threshold)
    # GMM / DDU on well-conditioned data (the reference's own unit goldens sit on singular covariances where the
    # float32 Cholesky + jitter search decides the value)
    Cg, Dg = 6, 24
    cen = rng.standard_normal((Cg, Dg)).astype(np.float32) * 2
    lab_g = rng.integers(0, Cg, 1800)
    fg = (cen[lab_g] + rng.standard_normal((1800, Dg)) * (0.5 + rng.random(Dg))).astype(np.float32)
    fg_te = (cen[rng.integers(0, Cg, 120)] * 0.8 + 1.1 * rng.standard_normal((120, Dg))).astype(np.float32)
    p = pp.GMMLatentSpace()
    p.setup(fg, ind_train_labels=lab_g)
    cases.update(gmm_train=fg, gmm_labels=lab_g, gmm_test=fg_te, gmm_scores=p.postprocess(fg_te),
                 gmm_loc=p.gmm.loc.numpy(), gmm_tril=p.gmm.scale_tril.numpy())
    p = pp.DDU(flip_sign=False, num_classes=Cg)
    p.device = "cpu"
    p.setup(fg, valid_feats=fg[:200], train_labels=lab_g)
    cases.update(ddu_scores=p.postprocess(fg_te), ddu_threshold=np.array(p.threshold))
    np.savez_compressed(os.path.join(OUT, "ref_f4.npz"), **cases)

    # ---------------- thresholds -------------------------------------------------
    sc = rng.standard_normal(1000) * 3 - 7
    np.savez_compressed(
        os.path.join(OUT, "ref_threshold.npz"),
        scores=sc,
        thr=np.array(ac.get_method_threshold(sc, 1.645)),
        thr1=np.array(ac.get_method_threshold(sc, 1.0)),
    )
    print("fixtures written to", os.path.abspath(OUT))
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn))} bytes")


if __name__ == "__main__":
    main()
