"""The baselines harness (``runia_core_amd.evaluation.baselines`` = reference ``runia_core/evaluation/baselines.py``):
host logic and the oracle's CPU form on the CPU, ``calculate_all_baselines`` itself on the GPU against the ten means the
reference's own test holds (/root/reference/tests/unit_test_baselines.py:209-268) and row by row against the oracle."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import harness  # noqa: E402
from test_oracle_goldens import _all_baselines_inputs, _fc_params  # noqa: E402

# the reference's list (unit_test_baselines.py:58-71) and configuration (:212-221)
BASELINES_NAMES = ["vim", "mdist", "msp", "knn", "energy", "ash", "dice", "react", "gen", "dice_react", "ddu", "raw"]
CFG = {"ood_datasets": ["test_ood"], "ash_percentile": 90, "react_percentile": 90, "dice_percentile": 90, "gen_gamma": 0.1,
       "k_neighbors": 10}
# golden index in tests/golden/reference_test_vectors.json["all_baselines_means"], tolerance on the mean (ash ~ 437 is a
# float32 value: one ulp is 3e-5).  The tenth number the reference's test holds, ddu = -863839.4375, is NOT reproduced by the
# reference itself in this environment: its own calculate_all_baselines, run here by path (tools/make_goldens_r6.py,
# tests/golden/ref_baselines.npz), returns -869670.8125 - twenty float32 Gaussians fitted on ~10 samples each in 20 dimensions
# sit behind a jitter ladder decided by round-off, and torch's float32 Cholesky differs between versions.  ddu is therefore
# pinned by that reference run, row by row, like the other eight baselines the reference can run here.
GOLDEN = {"msp": (0, 1e-6), "knn": (1, 1e-6), "energy": (2, 1e-6), "ash": (3, 1e-3), "gen": (4, 1e-5), "react": (5, 1e-5),
          "dice": (6, 1e-5), "dice_react": (7, 1e-5), "mdist": (8, 1e-6)}
REF_RUN = ["vim", "msp", "raw", "energy", "ash", "gen", "react", "mdist", "ddu"]  # what tools/make_goldens_r6.py could run
REF_CASES = {"unit": ["test_ood"], "wide": ["far", "near"]}


def _ref_case(tag):
    from test_oracle_goldens import load_npz

    g = load_npz("ref_baselines.npz")
    ind = {k: g[f"{tag}/in/{k}"] for k in ("train features", "train logits", "valid features", "valid logits")}
    ood = {f"{o} {kind}": g[f"{tag}/in/{o} {kind}"] for o in REF_CASES[tag] for kind in ("features", "logits")}
    return g, ind, ood, {"weight": g[f"{tag}/in/weight"], "bias": g[f"{tag}/in/bias"]}, int(g[f"{tag}/classes"])


def _close(got, exp, name, ill_posed=False):
    """1e-5 of max(|score|, 1).  ddu: float32 log-densities through a float32 Cholesky, 2e-4.  ``ill_posed`` - ddu on the
    reference's own test recipe: twenty Gaussians from 6-14 samples each in 20 dimensions, singular covariances lifted by a jitter
    of 1e-7, log-densities of -1e5 ... -3e6 that ARE the factorisation's round-off times 1e7 (the reference's held number is not
    reproduced by the reference here, see GOLDEN): only the order of magnitude is compared, and INTEGRATION.md lists the case under
    "Known divergences"."""
    e, g = np.asarray(exp, dtype=np.float64), np.asarray(got, dtype=np.float64)
    err = float(np.max(np.abs(g - e) / np.maximum(np.abs(e), 1.0)))
    if ill_posed and name == "ddu":
        return err < 0.5 and abs(g.mean() / e.mean() - 1.0) < 0.1
    return err < (2e-4 if name == "ddu" else 1e-5)


def _dicts():
    d = _all_baselines_inputs()
    ind = {"train features": d["tr_f"], "train logits": d["tr_l"], "valid features": d["va_f"], "valid logits": d["va_l"]}
    ood = {"test_ood features": d["ood_f"], "test_ood logits": d["ood_l"]}
    w, b = _fc_params()
    return ind, ood, {"weight": w, "bias": b}


@pytest.fixture(scope="module")
def golden_means():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_test_vectors.json")) as f:
        return [s["value"] for s in json.load(f)["all_baselines_means"]["scalars"]]


# ---------------------------------------------------------------- CPU: host logic + the oracle's form
def test_oracle_all_baselines_reference_means(golden_means):
    ind, ood, fc = _dicts()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = harness.all_baselines(harness.BASELINES, ind, ood, ["test_ood"], fc["weight"], fc["bias"], 20, 10, 90, 90, 90, 0.1)
    for name, (idx, tol) in GOLDEN.items():
        mean = float(np.asarray(got[name]["test_ood"], dtype=np.float64).mean())
        assert abs(mean - golden_means[idx]) < tol, (name, mean, golden_means[idx])
    assert np.array_equal(got["raw"]["valid"], got["msp"]["valid"]) and got["vim"]["test_ood"].shape == (200,)


@pytest.mark.parametrize("tag", ["unit", "wide"])
def test_oracle_all_baselines_vs_the_reference_run(tag):
    """The oracle's loop against what the reference's own calculate_all_baselines returned here (by-path run)."""
    g, ind, ood, fc, classes = _ref_case(tag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = harness.all_baselines(REF_RUN, ind, ood, REF_CASES[tag], fc["weight"], fc["bias"], classes, 10, 90, 90, 90, 0.1)
    for name in REF_RUN:
        for split in ["valid"] + REF_CASES[tag]:
            if name == "ash":  # upstream's scatter quirk (oracle.ash_s_linear_layer) permutes kept values in some rows:
                e = harness.H.linear_energy(harness.H.ash_s_linear_layer(np.array(  # the reference form is compared instead
                    ind["valid features"] if split == "valid" else ood[f"{split} features"], copy=True), 90), fc["weight"], fc["bias"])
                assert _close(e, g[f"{tag}/{split}/{name}"], name), (name, split)
                continue
            assert _close(got[name][split], g[f"{tag}/{split}/{name}"], name), (name, split)


def test_get_labels_from_logits_branches():
    # /root/reference/tests/unit_test_baselines.py:116-150
    from runia_core_amd.evaluation.baselines import get_labels_from_logits

    id_data = {"train logits": np.array([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1]]), "valid logits": np.array([[0.4, 0.5, 0.1]])}
    ood_data = {"ood1 logits": np.array([[0.2, 0.7, 0.1]])}
    id_res, ood_res = get_labels_from_logits(id_data.copy(), ood_data.copy(), ["ood1"])
    assert list(id_res["train labels"]) == [1, 0] and list(id_res["valid labels"]) == [1] and list(ood_res["ood1 labels"]) == [1]
    assert "train logits" not in id_res and "valid logits" not in id_res and "ood1 logits" not in ood_res  # popped, as upstream
    id2, ood2 = get_labels_from_logits({"train logits": [], "valid logits": []}, {"ood1 logits": []}, ["ood1"])
    assert len(id2["train labels"]) == 0 and len(id2["valid labels"]) == 0 and len(ood2["ood1 labels"]) == 0
    with pytest.raises(NotImplementedError):
        get_labels_from_logits({"train logits": [1, 2, 3], "valid logits": [4, 5, 6]}, {"ood1 logits": [1, 2]}, ["ood1"])
    with pytest.raises(NotImplementedError):
        get_labels_from_logits({"train logits": np.zeros((2, 3)), "valid logits": np.zeros((2, 3))}, {"ood1 logits": [1, 2]}, ["ood1"])
    # a background column (detectors: 21 or 11 columns) never wins the argmax; only the valid logits present
    lg = np.zeros((3, 11))
    lg[:, -1] = 5.0
    lg[1, 4] = 1.0
    id3, _ = get_labels_from_logits({"valid logits": lg}, {}, [])
    assert list(id3["valid labels"]) == [0, 4, 0] and len(id3["train labels"]) == 0


def test_remove_latent_features_and_names():
    # /root/reference/tests/unit_test_baselines.py:152-159
    from runia_core_amd.evaluation import baselines as B

    id_out, ood_out = B.remove_latent_features({"train features": np.ones((2, 3)), "valid features": np.zeros((1, 3)), "x": 1},
                                               {"oodA features": np.full((1, 3), 2.0)}, ["oodA", "absent"])
    assert id_out == {"x": 1} and ood_out == {}
    assert set(B.baseline_name_dict) == {"pred_h", "mi", "msp", "energy", "mdist", "knn", "ash", "dice", "react", "dice_react", "vim",
                                         "gen", "ddu", "raw"}
    assert all(set(v) == {"plot_title", "x_axis", "plot_name"} for v in B.baseline_name_dict.values())
    assert B.baseline_name_dict["mi"]["plot_name"] == "pred_mi" and B.baseline_name_dict["raw"]["plot_name"] == "raw_predictions"
    import runia_core_amd.evaluation as E

    assert E.calculate_all_baselines is B.calculate_all_baselines and set(B.__all__) <= set(dir(E))


def test_gen_refuses_more_than_21_classes_before_any_work():
    from runia_core_amd.evaluation.baselines import calculate_all_baselines

    with pytest.raises(ValueError, match="does not yet support num_classes greater than 21"):
        calculate_all_baselines(["gen"], {}, {}, None, CFG, 22)


# ---------------------------------------------------------------- GPU: the harness itself
@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [False, True])
def test_calculate_all_baselines_reference_goldens(golden_means, device_resident):
    from types import SimpleNamespace

    from runia_core_amd.evaluation.baselines import calculate_all_baselines

    ind, ood, fc = _dicts()
    ref_ind = {k: v.copy() for k, v in ind.items()}
    ref_ood = {k: v.copy() for k, v in ood.items()}
    cfg = SimpleNamespace(**CFG) if device_resident else dict(CFG)  # attribute access (OmegaConf) and a plain dict
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ind_out, ood_out, scores = calculate_all_baselines(BASELINES_NAMES, ind, ood, fc, cfg, 20, device_resident=device_resident)
        exp = harness.all_baselines(harness.BASELINES, ref_ind, ref_ood, ["test_ood"], fc["weight"], fc["bias"], 20, 10, 90, 90, 90, 0.1)
    assert ind_out is ind and ood_out is ood
    assert set(scores) == {f"test_ood {b}" for b in BASELINES_NAMES}
    assert all(isinstance(v, np.ndarray) and v.shape == (200,) for v in scores.values())
    for b in BASELINES_NAMES:  # the InD valid scores live in the InD dictionary under the baseline's name
        assert isinstance(ind[b], np.ndarray) and ind[b].shape == (200,)
    # the logits left the dictionaries, the argmax labels came in (get_labels_from_logits)
    assert "train logits" not in ind and "valid logits" not in ind and "test_ood logits" not in ood
    assert np.array_equal(ind["train labels"], np.argmax(ref_ind["train logits"], axis=-1))
    assert np.array_equal(ood["test_ood labels"], np.argmax(ref_ood["test_ood logits"], axis=-1))
    # the reference's own numbers
    for name, (idx, tol) in GOLDEN.items():
        mean = float(np.asarray(scores[f"test_ood {name}"], dtype=np.float64).mean())
        assert abs(mean - golden_means[idx]) < tol, (name, mean, golden_means[idx])
    # row by row against the oracle's CPU form of the loop, both splits
    for name in BASELINES_NAMES:
        for split, got in (("valid", ind[name]), ("test_ood", scores[f"test_ood {name}"])):
            assert _close(got, exp[name][split], name, ill_posed=True), (name, split)
    # dtypes as upstream (ref_baselines.npz): float32 in -> float32 scores, except the float64 Mahalanobis
    for name in ("msp", "raw", "energy", "knn", "gen", "ash", "react", "dice", "dice_react", "ddu", "vim"):
        assert scores[f"test_ood {name}"].dtype == np.float32, name
    assert scores["test_ood mdist"].dtype == np.float64


@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [False, True])
@pytest.mark.parametrize("tag", ["unit", "wide"])
def test_calculate_all_baselines_vs_the_reference_run(tag, device_resident):
    """Row by row against what the reference's own calculate_all_baselines returned (tools/make_goldens_r6.py)."""
    from runia_core_amd.evaluation.baselines import calculate_all_baselines

    g, ind, ood, fc, classes = _ref_case(tag)
    cfg = {"ood_datasets": REF_CASES[tag], "ash_percentile": 90, "react_percentile": 90, "dice_percentile": 90, "gen_gamma": 0.1,
           "k_neighbors": 10}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ind, ood, scores = calculate_all_baselines(REF_RUN, ind, ood, fc, cfg, classes, device_resident=device_resident)
    assert np.array_equal(ind["train labels"], g[f"{tag}/train labels"])
    ash_ok = {}
    for name in REF_RUN:
        for split in ["valid"] + REF_CASES[tag]:
            got = ind[name] if split == "valid" else scores[f"{split} {name}"]
            exp = g[f"{tag}/{split}/{name}"]
            assert got.dtype == exp.dtype and got.shape == exp.shape, (name, split, got.dtype, exp.dtype)
            if name == "ash":  # rows where upstream's scatter is self-consistent (see test_f4_kernels_and_classes)
                x = g[f"{tag}/in/valid features"] if split == "valid" else g[f"{tag}/in/{split} features"]
                ok = np.all(np.isclose(harness.H.ash_s_linear_layer(x.copy(), 90), harness.H.ash_s_defined(x.copy(), 90), rtol=1e-6), axis=1)
                ash_ok[split] = int(ok.sum())
                assert _close(got[ok], exp[ok], name), (name, split)
                continue
            assert _close(got, exp, name, ill_posed=(tag == "unit")), (name, split)
    assert all(v > 0 for v in ash_ok.values())


@pytest.mark.gpu
def test_calculate_all_baselines_device_resident_same_bits():
    """Scores from splits uploaded once equal the upstream-style call's (one upload per postprocess) bit for bit, on rows wide
    enough for the matrix-core kernels (kNN 3 000 x 600 x 256, Mahalanobis 256-d, heads of 10 classes)."""
    from runia_core_amd.evaluation.baselines import calculate_all_baselines

    rng = np.random.default_rng(77)
    n_tr, n_te, d, c = 3000, 600, 256, 10
    centres = rng.standard_normal((c, d)).astype(np.float32)
    w = (rng.standard_normal((c, d)) / np.sqrt(d)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32) * 0.1

    def split(n, shift):
        lab = rng.integers(0, c, n)
        f = np.maximum(centres[lab] + rng.standard_normal((n, d)).astype(np.float32) + np.float32(shift), 0).astype(np.float32)
        return f, (f @ w.T + b).astype(np.float32)

    def dicts():
        r = np.random.default_rng(5)
        nonlocal rng
        rng = r
        trf, trl = split(n_tr, 0.0)
        vaf, val = split(n_te, 0.0)
        o1f, o1l = split(n_te, 0.6)
        o2f, o2l = split(n_te, -0.4)
        return ({"train features": trf, "train logits": trl, "valid features": vaf, "valid logits": val},
                {"far features": o1f, "far logits": o1l, "near features": o2f, "near logits": o2l})

    cfg = {"ood_datasets": ["far", "near"], "ash_percentile": 90, "react_percentile": 90, "dice_percentile": 90, "gen_gamma": 0.1,
           "k_neighbors": 50}
    names = ["msp", "raw", "knn", "energy", "ash", "gen", "react", "dice", "dice_react", "mdist", "ddu", "vim"]
    out = []
    for resident in (False, True):
        ind, ood = dicts()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ind, ood, sc = calculate_all_baselines(names, ind, ood, {"weight": w, "bias": b}, cfg, c, device_resident=resident)
        out.append((ind, sc))
    (ind_a, sc_a), (ind_b, sc_b) = out
    assert set(sc_a) == set(sc_b) == {f"{o} {n}" for o in ("far", "near") for n in names}
    for key in sc_a:
        assert sc_a[key].dtype == sc_b[key].dtype and np.array_equal(sc_a[key], sc_b[key], equal_nan=True), key
    for n in names:
        assert np.array_equal(ind_a[n], ind_b[n], equal_nan=True), n
    assert all(np.isfinite(v).all() for v in sc_a.values())
