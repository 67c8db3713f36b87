"""Latency of ONE call of the public postprocessors on few rows - NumPy in, NumPy out, as the reference's API is used from
an evaluation loop or a service handling one image / one request at a time: host wall time per call.
    gpurun -- python tools/ablate/run_api_latency.py"""
import gc, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd.inference.postprocessors import KNN, Energy, Mahalanobis, MSP
gc.disable()
rng = np.random.default_rng(0)

def t(fn, reps=100):
    for _ in range(10): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6

D, C, M = 2048, 10, 20000
centres = rng.standard_normal((C, D)).astype(np.float32)
labels = rng.integers(0, C, M)
feats = np.maximum(centres[labels] + rng.standard_normal((M, D)).astype(np.float32), 0).astype(np.float32)
logits = rng.standard_normal((M, 1000)).astype(np.float32)
maha = Mahalanobis(flip_sign=False, num_classes=C); maha.setup(feats[:6000], train_labels=labels[:6000], valid_feats=feats[6000:7000])
knn = KNN(flip_sign=False, k_neighbors=50); knn.setup(feats, valid_feats=feats[:1000])
en = Energy(flip_sign=False); en.setup(logits[:1000])
for n in (1, 8, 100, 1000):
    f, lg = feats[:n].copy(), logits[:n].copy()
    print(f"rows {n:5d}:  Mahalanobis.postprocess {t(lambda: maha.postprocess(f)):8.1f} us   KNN.postprocess (bank {M}) "
          f"{t(lambda: knn.postprocess(f)):8.1f} us   Energy.postprocess {t(lambda: en.postprocess(lg)):7.1f} us", flush=True)
