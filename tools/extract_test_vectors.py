#!/usr/bin/env python3
"""Collect the golden NUMBERS hard-coded in the reference's unit tests into
tests/golden/reference_test_vectors.json (data only; parsed with ``ast`` so that
no test code is copied).  Build-container only (reads /root/reference/tests).
"""
from __future__ import annotations

import ast
import json
import os

REF_TESTS = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_test_vectors.json")

# (file, test function) -> key
WANTED = {
    ("unit_test_feature_extraction.py", "test_single_image_entropy_calculation"): "entropy_single_image",
    ("unit_test_feature_extraction.py", "test_get_dl_h_z"): "entropy_get_dl_h_z",
    ("unit_test_dim_reduction.py", "test_pca_ds_split"): "pca_ds_split",
    ("unit_test_dim_reduction.py", "test_apply_pca_transform"): "pca_transform",
    ("unit_test_postprocessors.py", "test_md_postprocess"): "md_unit",
    ("unit_test_postprocessors.py", "test_kde_postprocess"): "kde_unit",
    ("unit_test_postprocessors.py", "test_mahalanobis_postprocess"): "mahalanobis_unit",
    ("unit_test_postprocessors.py", "test_energy_postprocess"): "energy_unit",
    ("unit_test_postprocessors.py", "test_cmd_postprocess"): "cmd_unit",
    ("unit_test_postprocessors.py", "test_knn_postprocess"): "knn_latent_unit",
    ("unit_test_baselines.py", "test_larem_postprocessor"): "larem_baselines",
    ("unit_test_baselines.py", "test_lared_postprocessor"): "lared_baselines",
    ("unit_test_baselines.py", "test_all_baselines_postp"): "all_baselines_means",
    ("unit_test_metrics.py", "test_hz_detector_results"): "metrics_hz",
    ("unit_test_metrics.py", "test_evaluate_postprocessors"): "metrics_postprocessors",
    ("integration_tests.py", "test_extract_entropy_larex"): "entropy_degenerate",
}


def _num(node):
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return node.value
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        v = _num(node.operand)
        return None if v is None else -v
    return None


def collect(fn_node):
    lists, scalars = [], []
    for node in ast.walk(fn_node):
        if isinstance(node, ast.List) and node.elts:
            vals = [_num(e) for e in node.elts]
            if all(v is not None for v in vals):
                lists.append({"line": node.lineno, "values": vals})
        elif isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "assertAlmostEqual":
            for a in node.args[:2]:
                v = _num(a)
                if isinstance(v, float):
                    scalars.append({"line": node.lineno, "value": v})
    lists.sort(key=lambda d: d["line"])
    scalars.sort(key=lambda d: d["line"])
    return lists, scalars


def main():
    out = {}
    for (fname, func), key in WANTED.items():
        path = os.path.join(REF_TESTS, fname)
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name == func:
                lists, scalars = collect(node)
                out[key] = {
                    "source": f"/root/reference/tests/{fname}:{node.lineno} ({func})",
                    "lists": lists,
                    "scalars": scalars,
                }
                break
        else:
            raise SystemExit(f"{func} not found in {fname}")
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(k, [len(l["values"]) for l in v["lists"]], [s["value"] for s in v["scalars"]][:6])


if __name__ == "__main__":
    main()
