import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # librunia_hip.so is a build artefact (git-ignored): make sure it exists and is not older than its sources.
    # `make` is incremental; hipcc cross-compiles gfx950 without a GPU.
    import subprocess

    csrc = os.path.join(ROOT, "runia_core_amd", "csrc")
    try:
        subprocess.run(["make", "-C", csrc, "-j8", "-s"], check=True, stdout=subprocess.DEVNULL)
    except Exception as e:  # pragma: no cover - surfaces as loud failures of every test that needs the library
        print(f"WARNING: could not (re)build librunia_hip.so: {e}", file=sys.stderr)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def ref_vectors():
    """Golden numbers hard-coded in the reference's own unit tests
    (tools/extract_test_vectors.py)."""
    with open(os.path.join(GOLDEN, "reference_test_vectors.json")) as f:
        return json.load(f)


def generate_test_data(num_samples=10, feature_dim=32, num_classes=10, seed=42):
    """Seeded input recipe of /root/reference/tests/unit_test_postprocessors.py:66-100
    (data recipe only: the reference goldens are defined on these inputs)."""
    import torch

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


def rel_err(got, ref):
    """Parity criterion of BASELINE.md section 5: |d| <= tol * max(1, |ref|)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))) if ref.size else 0.0


def kde_hd_inputs(d, seed):
    """Seeded inputs of tests/golden/ref_kde_hd.npz (same recipe as tools/make_goldens_r2.py::kde_hd_inputs; only the
    reference's scores are stored, the embeddings are regenerated and verified by checksum)."""
    g = np.random.default_rng(seed)
    scale = 0.8 + 0.4 * g.random(d)
    train = g.standard_normal((3000, d)) * scale
    ind = g.standard_normal((600, d)) * scale
    ood = g.standard_normal((600, d)) * scale * 1.15 + 0.35
    return train, ind, ood


# What the reference's LaRED (KDELatentSpace -> sklearn KernelDensity tree) returned on those inputs, against the exact
# log-density (measured in the build container, tools/make_goldens_r2.py --only kde_hd; quoted in INTEGRATION.md):
#   D    max |ref - exact|   AUROC ref / exact      FPR@95 ref / exact
KDE_HD_MEASURED = {
    16: dict(max_abs=2.9e-7, auroc=(0.78705835, 0.78705835), fpr95=(0.66166669, 0.66166669)),
    64: dict(max_abs=34.3, auroc=(0.63099998, 0.91174167), fpr95=(0.63166666, 0.36333334)),
    256: dict(max_abs=260.2, auroc=(0.64546669, 0.99583334), fpr95=(0.54166669, 0.01333333)),
}
