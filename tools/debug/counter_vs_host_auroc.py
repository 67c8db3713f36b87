"""Counter-draw mode vs host draws: AUROC over many seeds on fixed InD / OOD latents (bench.py's synthetic sets)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, oracle
import runia_core_amd as rc
from runia_core_amd import _hip
from runia_core_amd.inference import LaREMPipeline, MDLatentSpace
dev = torch.device("cuda", 0)
N = 10000
probe = LaREMPipeline(None, None, bench.N_MC, bench.DROP_PROB, bench.BLOCK)
xtr, rtr = bench.synth_latents(4096, 1234, 0.0, dev)
h_train = probe.entropy(probe.stack(xtr, rtr)).cpu().numpy()
np.random.seed(1234)
red, pca = rc.apply_pca_ds_split(h_train, bench.N_PCA)
md = MDLatentSpace(); md.setup(red)
pipe = LaREMPipeline(md, pca, bench.N_MC, bench.DROP_PROB, bench.BLOCK)
x, _ = bench.synth_latents(N, 1235, 0.0, dev)
xo, _ = bench.synth_latents(N, 998, 0.0, dev, corr=0.25)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 40
res = {"counter": [], "host": [], "counter_as_explicit": []}
for sd in range(S):
    ic = pipe.score_latents(x, _hip.CounterDraws(99 + sd, 0, True)).cpu().numpy()
    oc = pipe.score_latents(xo, _hip.CounterDraws(99 + sd, N, True)).cpu().numpy()
    res["counter"].append(oracle.auroc_fpr95_aupr(ic, oc)[0])
    _, ri = bench.synth_latents(N, 5000 + sd, 0.0, dev, dead="redraw")
    _, ro = bench.synth_latents(N, 6000 + sd, 0.0, dev, dead="redraw")
    ih = pipe.score_latents(x, ri).cpu().numpy(); oh = pipe.score_latents(xo, ro).cpu().numpy()
    res["host"].append(oracle.auroc_fpr95_aupr(ih, oh)[0])
for k, v in res.items():
    if v:
        v = np.asarray(v); print(f"{k:10s} mean {v.mean():.5f} sd {v.std(ddof=1):.5f} se {v.std(ddof=1)/np.sqrt(len(v)):.5f} n {len(v)}")
c, h = np.asarray(res["counter"]), np.asarray(res["host"])
print("gap", c.mean() - h.mean(), "+-", np.sqrt(c.var(ddof=1)/len(c) + h.var(ddof=1)/len(h)))
# draw statistics: fraction below gamma, mean, per-position fractions
g = bench.DROP_PROB / bench.BLOCK**2
dc = _hip.mc_draws(N, 16, 4, 4, 99, 0); _, dh = bench.synth_latents(N, 5000, 0.0, dev, dead="keep")
print("P(u<gamma): counter", float((dc < g).float().mean()), "host", float((dh < g).float().mean()), "gamma", g)
print("mean: counter", float(dc.mean()), "host", float(dh.mean()))
