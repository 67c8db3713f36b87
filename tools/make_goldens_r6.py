#!/usr/bin/env python3
"""Round-6 fixture: the reference's own baselines harness - ``calculate_all_baselines`` of
/root/reference/runia_core/evaluation/baselines.py, imported by path (recipe of tools/make_goldens.py) - run on the inputs of
its own test (/root/reference/tests/unit_test_baselines.py:209-246) and on a second, wider case.

Run here: vim, msp, raw, energy, ash, gen, react, mdist, ddu.  Not run here: knn (faiss is absent), dice / dice_react
(RouteDICE calls .cuda(); no GPU in the build container) - those three stay pinned by the means the reference's test holds.

Writes tests/golden/ref_baselines.npz: the inputs, and per baseline the InD valid scores and the OoD scores the reference
returned (plus the labels it derived).  Only data travels.

Usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tools/make_goldens_r6.py
"""
from __future__ import annotations

import os
import sys
import warnings

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch

from make_goldens import OUT, _load_reference  # noqa: E402

NAMES = ["vim", "msp", "raw", "energy", "ash", "gen", "react", "mdist", "ddu"]


class Cfg(dict):
    __getattr__ = dict.__getitem__


def unit_case():
    """unit_test_baselines.py:209-246 (SEED = 1, 200 samples, 20 dimensions = 20 classes)."""
    torch.manual_seed(1)
    np.random.seed(1)
    fc = {"weight": np.random.rand(20, 20).astype(np.float32), "bias": np.random.rand(20).astype(np.float32)}
    f32 = lambda: np.float32(np.random.random((200, 20)))  # noqa: E731
    ind = {"train features": f32(), "train logits": f32(), "valid features": f32(), "valid logits": f32()}
    ood = {"test_ood features": f32(), "test_ood logits": f32()}
    return ind, ood, fc, ["test_ood"], 20


def wide_case():
    """Class-structured features (96-d, 10 classes, logits from a linear head), two OoD sets."""
    rng = np.random.default_rng(606)
    d, c = 96, 10
    centres = rng.standard_normal((c, d)).astype(np.float32)
    w = (rng.standard_normal((c, d)) / np.sqrt(d)).astype(np.float32)
    b = (rng.standard_normal(c) * 0.1).astype(np.float32)

    def split(n, shift):
        lab = rng.integers(0, c, n)
        f = np.maximum(centres[lab] + rng.standard_normal((n, d)).astype(np.float32) + np.float32(shift), 0).astype(np.float32)
        return f, (f @ w.T + b).astype(np.float32)

    trf, trl = split(1500, 0.0)
    vaf, val = split(300, 0.0)
    of, ol = split(300, 0.7)
    nf, nl = split(300, -0.3)
    ind = {"train features": trf, "train logits": trl, "valid features": vaf, "valid logits": val}
    ood = {"far features": of, "far logits": ol, "near features": nf, "near logits": nl}
    return ind, ood, {"weight": w, "bias": b}, ["far", "near"], c


def main():
    _load_reference()
    import runia_core.evaluation.baselines as B  # by path: needs inference.postprocessors only

    out = {}
    for tag, case in (("unit", unit_case), ("wide", wide_case)):
        ind, ood, fc, ood_names, classes = case()
        for k, v in {**ind, **ood}.items():
            out[f"{tag}/in/{k}"] = v.copy()
        out[f"{tag}/in/weight"], out[f"{tag}/in/bias"] = fc["weight"].copy(), fc["bias"].copy()
        out[f"{tag}/classes"] = np.int64(classes)
        cfg = Cfg(ood_datasets=ood_names, ash_percentile=90, react_percentile=90, dice_percentile=90, gen_gamma=0.1, k_neighbors=10)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ind2, ood2, scores = B.calculate_all_baselines(baselines_names=NAMES, ind_data_dict=ind, ood_data_dict=ood, fc_params=fc,
                                                           cfg=cfg, num_classes=classes)
        for name in NAMES:
            out[f"{tag}/valid/{name}"] = np.asarray(ind2[name])
            for o in ood_names:
                out[f"{tag}/{o}/{name}"] = np.asarray(scores[f"{o} {name}"])
        out[f"{tag}/train labels"] = np.asarray(ind2["train labels"])
        for o in ood_names:
            out[f"{tag}/{o} labels"] = np.asarray(ood2[f"{o} labels"])
        assert "train logits" not in ind2 and f"{ood_names[0]} logits" not in ood2
        print(tag, {n: float(np.mean(scores[f"{ood_names[0]} {n}"])) for n in NAMES}, file=sys.stderr)
    np.savez_compressed(os.path.join(OUT, "ref_baselines.npz"), **out)
    print("wrote ref_baselines.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
