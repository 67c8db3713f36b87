#!/usr/bin/env python3
"""Round-4 fixtures: the free functions of the reference's inference/funcs.py that round 3 had not mirrored, run from
the reference's own file (imported by path, recipe of tools/make_goldens.py), plus the `__all__` lists of every mirrored
reference module (names only, read with `ast`).

Writes tests/golden/ref_funcs_r4.npz (inputs + what the reference returned) and tests/golden/reference_all_names.json.
Only data travels.  Not run here: RouteDICE.forward / calculate_mask_weight (they call .cuda(), no GPU in the build
container) - the DICE arithmetic stays pinned by the all-baselines golden of round 2.

Usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tools/make_goldens_r4.py
"""
from __future__ import annotations

import ast
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch

from make_goldens import OUT, REF, _load_reference  # noqa: E402


def all_names(path):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "__all__" for t in node.targets):
            return [ast.literal_eval(e) for e in node.value.elts]
    return None


def main():
    _, funcs, _, _ = _load_reference()
    out = {}
    rng = np.random.default_rng(404)

    # ash_s_linear_layer (funcs.py:230-261): rows without ties, D = 64 and 300, the reference's default percentile and 65
    for tag, d, pct in (("a", 64, 85), ("b", 300, 65)):
        x = np.abs(rng.standard_normal((24, d))).astype(np.float32) + 0.01
        out[f"ashl_{tag}_x"] = x
        out[f"ashl_{tag}_pct"] = np.int64(pct)
        out[f"ashl_{tag}_y"] = funcs.ash_s_linear_layer(x.copy(), pct)

    # ash_s_conv_layer (funcs.py:194-227): 4-D maps, (5, 16, 4, 4) and (3, 96, 7, 7) (4 704 elements per sample)
    for tag, shape, pct in (("a", (5, 16, 4, 4), 65), ("b", (3, 96, 7, 7), 90)):
        x = torch.from_numpy(np.abs(rng.standard_normal(shape)).astype(np.float32) + 0.01)
        out[f"ashc_{tag}_x"] = x.numpy().copy()
        out[f"ashc_{tag}_pct"] = np.int64(pct)
        xin = x.clone()
        y = funcs.ash_s_conv_layer(xin, pct)
        out[f"ashc_{tag}_y"] = y.numpy()
        out[f"ashc_{tag}_x_after"] = xin.numpy()  # the reference prunes its argument in place (view + scatter_)

    # generalized_entropy (funcs.py:347-375): f64 and f32 probabilities, M below / at / above the class count
    for tag, c, m, gamma, dt in (("a", 10, 10, 0.1, np.float64), ("b", 100, 10, 0.1, np.float32), ("c", 1000, 100, 0.5, np.float32)):
        logits = rng.standard_normal((40, c)) * 3.0
        p = np.exp(logits - logits.max(1, keepdims=True))
        p = (p / p.sum(1, keepdims=True)).astype(dt)
        out[f"gen_{tag}_p"] = p
        out[f"gen_{tag}_gm"] = np.array([gamma, m], dtype=np.float64)
        out[f"gen_{tag}_s"] = funcs.generalized_entropy(p, gamma, m)

    # get_predictive_uncertainty_score (funcs.py:430-465): logits (N * n_mc, C) -> predictive entropy, mutual information
    for tag, n, n_mc, c in (("a", 30, 16, 10), ("b", 12, 5, 43), ("c", 7, 2, 1000), ("d", 9, 32, 100)):
        logits = torch.from_numpy((rng.standard_normal((n * n_mc, c)) * 2.0).astype(np.float32))
        ph, mi = funcs.get_predictive_uncertainty_score(logits, n_mc)
        out[f"pu_{tag}_logits"] = logits.numpy()
        out[f"pu_{tag}_nmc"] = np.int64(n_mc)
        out[f"pu_{tag}_pred_h"] = ph.numpy()
        out[f"pu_{tag}_mi"] = mi.numpy()

    # get_mcd_pred_uncertainty_score (funcs.py:378-427) with a small dropout classifier and a list of batches as the
    # loader; the reference draws its dropout masks from torch's global CPU generator: the masks (= the logits it saw)
    # are recorded through a forward hook so that the mirror can be fed the same MC outputs
    torch.manual_seed(7)
    model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(48, 32), torch.nn.ReLU(), torch.nn.Dropout(0.3),
                                torch.nn.Linear(32, 10))
    model.train()
    loader = [(torch.randn(1, 3, 4, 4), torch.zeros(1)) for _ in range(6)]
    seen = []
    hook = model.register_forward_hook(lambda m, i, o: seen.append(o.detach().clone()))
    torch.manual_seed(11)
    samples, ph, mi = funcs.get_mcd_pred_uncertainty_score(model, loader, 4)
    hook.remove()
    out["mcd_logits"] = torch.cat(seen, 0).numpy()
    out["mcd_nmc"] = np.int64(4)
    out["mcd_samples"] = samples.numpy()
    out["mcd_pred_h"] = ph.numpy()
    out["mcd_mi"] = mi.numpy()

    # get_dice_feat_mean_react_percentile (funcs.py:468-495): a "model" with the attribute the function asserts
    class Feat(torch.nn.Module):
        dice_precompute = True

        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 12, 3, padding=1)

        def forward(self, x):
            return torch.relu(self.conv(x))

    torch.manual_seed(3)
    fm = Feat()
    batches = [(torch.randn(1, 3, 6, 6), torch.zeros(1, dtype=torch.long)) for _ in range(9)]
    with torch.no_grad():
        mean, thr = funcs.get_dice_feat_mean_react_percentile(fm, batches, 90)
    out["dice_w"] = fm.conv.weight.detach().numpy()
    out["dice_b"] = fm.conv.bias.detach().numpy()
    out["dice_inputs"] = torch.cat([b[0] for b in batches], 0).numpy()
    out["dice_mean"] = np.asarray(mean)
    out["dice_thr"] = np.float64(thr)

    np.savez_compressed(os.path.join(OUT, "ref_funcs_r4.npz"), **out)

    names = {}
    for rel in ("inference/funcs.py", "inference/postprocessors.py", "inference/abstract_classes.py", "inference/image_level.py",
                "inference/__init__.py", "evaluation/entropy.py", "evaluation/metrics.py", "evaluation/latent_space.py", "evaluation/baselines.py",
                "dimensionality_reduction.py",
                "feature_extraction/abstract_classes.py", "feature_extraction/utils.py", "feature_extraction/image_level.py",
                "feature_extraction/object_level.py", "feature_extraction/__init__.py", "llm_uncertainty/scores.py"):
        names[rel] = all_names(f"{REF}/runia_core/{rel}")
    json.dump(names, open(os.path.join(OUT, "reference_all_names.json"), "w"), indent=1, sort_keys=True)
    print("wrote ref_funcs_r4.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim}, file=sys.stderr)
    print({k: (len(v) if v else None) for k, v in names.items()})


if __name__ == "__main__":
    main()
