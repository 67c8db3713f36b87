"""Do K1 (vector ALUs) and K2' (f64 matrix cores) share the chip when they run at the same time?  Independent data, two
streams, each a long back-to-back chain of its launch (captured as a graph: no host in the way), then both graphs at once."""
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
x = torch.relu(torch.randn(n, c, 4, 4, device="cuda", generator=g))
folded = pipe._folded_state()
h_a = pipe.entropy_from_latents(x, _hip.CounterDraws(7, 0))
h_b = torch.nan_to_num(h_a.clone())
table = _hip.mc_mask_table(_hip.CounterDraws(7, 0), n, 4, 4, n_mc, 0.5, 2)
R1, R2 = 40, 100
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
out_h = torch.empty_like(h_a)
out_s = torch.empty(n, dtype=torch.float64, device="cuda")

def chain_k1():
    for _ in range(R1): _hip.mc_entropy(x, None, n_mc, 0.5, 2, 5, table=table, out=out_h)
def chain_k2():
    for _ in range(R2): _hip.proj_sq_score(h_b, *folded, out=out_s)

def graph_of(fn, stream):
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            fn()
    return gr
g1, g2 = graph_of(chain_k1, s1), graph_of(chain_k2, s2)

def wall(fns, reps=5):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def on(stream, gr):
    def f():
        with torch.cuda.stream(stream): gr.replay()
    return f
t1, t2 = wall([on(s1, g1)]), wall([on(s2, g2)])
tb = wall([on(s1, g1), on(s2, g2)])
print(f"K1 x {R1}: {t1:.3f} ms ({t1 / R1 * 1e3:.1f} us each)   K2' x {R2}: {t2:.3f} ms ({t2 / R2 * 1e3:.1f} us each)   both at once: {tb:.3f} ms "
      f"(sum {t1 + t2:.3f}, max {max(t1, t2):.3f})")
