"""The refit loop of the reference's evaluation harness (evaluation/latent_space.py:135-140: one PCA fit + one LaREM fit per
n_pca_components) on 50 000 x 512 entropy rows: the reference's host calls against runia_core_amd.config.device_fit."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import runia_core_amd as rc
from runia_core_amd import config
from runia_core_amd.inference import MDLatentSpace

rng = np.random.default_rng(0)
h = (rng.standard_normal((50_000, 512)) * (0.2 + rng.random(512)) + rng.standard_normal(512)).astype(np.float64)
te = h[:10_000]
print(f"{'n_comp':>6s} {'mode':>8s} {'pca fit s':>10s} {'md fit s':>9s} {'total s':>8s}   max |score - host| / max(1, |host|)")
for n_comp in (16, 32, 64, 128, 256):
    ref = None
    for mode in ("host", "device"):
        config.device_fit = mode == "device"
        for rep in range(2):  # second repetition is the steady state (first: code objects, allocator)
            np.random.seed(7)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            red, pca = rc.apply_pca_ds_split(h, n_comp)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            md = MDLatentSpace(); md.setup(red)
            torch.cuda.synchronize(); t2 = time.perf_counter()
        s = md.postprocess(rc.apply_pca_transform(te, pca))
        if ref is None:
            ref, err = s, 0.0
        else:
            err = float(np.max(np.abs(s - ref) / np.maximum(1.0, np.abs(ref))))
        print(f"{n_comp:6d} {mode:>8s} {t1 - t0:10.3f} {t2 - t1:9.3f} {t2 - t0:8.3f}   {err:.1e}", flush=True)
config.device_fit = None
