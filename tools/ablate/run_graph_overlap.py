"""Headline step (K0 + K1 + K2') as a HIP graph: sequential, row blocks over two streams, and K2'(i) beside K1(i + 1).
Does the matrix-core launch hide under the vector-ALU launch when the host's launch cost is out of the way?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
from runia_core_amd.dimensionality_reduction import DevicePCA
from runia_core_amd.inference import LaREMPipeline, MDLatentSpace

n, c, n_pca, n_mc = 10_000, 512, 256, 16
rng = np.random.default_rng(0)
comp = np.linalg.qr(rng.standard_normal((c, n_pca)))[0].T
a = rng.standard_normal((n_pca, n_pca))
md = MDLatentSpace()
md.feats_mean, md.precision, md._setup_flag = rng.standard_normal((1, n_pca)) * 0.1, a @ a.T / n_pca + np.eye(n_pca), True
pipe = LaREMPipeline(md, DevicePCA(comp, rng.standard_normal(c), rng.random(n_pca) + 0.05, True), n_mc, 0.5, 2)
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.relu(torch.randn(n, c, 4, 4, device="cuda", generator=g)) for _ in range(3)]
S = 8  # steps per graph

def run_steps(chunks):
    out = []
    for i in range(S):
        out.append(pipe.score_latents(xs[i % 3], _hip.CounterDraws(7, i * n), chunks=chunks))
    return out

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / S * 1e3

ref = [s.clone() for s in run_steps(1)]
print(f"eager, one launch chain per step: {timed(lambda: run_steps(1), 40):.4f} ms/step", flush=True)
for chunks in (1, 2, 4):
    run_steps(chunks); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        outs = run_steps(chunks)
    gr.replay(); torch.cuda.synchronize()
    same = all(torch.equal(o, r) for o, r in zip(outs, ref))
    print(f"graph of {S} steps, {chunks} row block(s) per step: {timed(gr.replay, 40):.4f} ms/step  same bits {same}", flush=True)
