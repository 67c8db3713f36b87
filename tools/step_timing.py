"""Debug helper: per-step and per-stage HIP-event timings of the bench workload."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import runia_core_amd as rc
from runia_core_amd.inference import LaREMPipeline, MDLatentSpace

dev = torch.device("cuda", 0)
probe = LaREMPipeline(None, None, 16, 0.5, 2)
xtr, rtr = bench.synth_latents(4096, 1234, 0.0, dev)
h_train = probe.entropy(probe.stack(xtr, rtr)).cpu().numpy()
np.random.seed(1234)
red, pca = rc.apply_pca_ds_split(h_train, 256)
md = MDLatentSpace(); md.setup(red)
pipe = LaREMPipeline(md, pca, 16, 0.5, 2)
x, rand = bench.synth_latents(10000, 1235, 0.0, dev)
def E(): return torch.cuda.Event(enable_timing=True)
for rep in range(3):
    rows = []
    for i in range(12):
        e = [E() for _ in range(5)]
        e[0].record(); z = pipe.stack(x, rand)
        e[1].record(); h = pipe.entropy(z)
        e[2].record(); y = pipe.pca.transform_device(h)
        e[3].record(); s = md.postprocess_device(y)
        e[4].record()
        torch.cuda.synchronize()
        rows.append([e[j].elapsed_time(e[j+1]) for j in range(4)])
    r = np.array(rows)
    print("rep", rep, "median ms [stack, entropy, pca, md]:", np.round(np.median(r, 0), 4), "max:", np.round(r.max(0), 3), flush=True)
    t0 = time.perf_counter()
    for i in range(20):
        s = pipe.score_latents(x, rand)
    torch.cuda.synchronize()
    print("   20 steps back-to-back: %.4f ms/step" % ((time.perf_counter() - t0) * 50), "finite:", bool(torch.isfinite(s).all()), float(s.mean()), flush=True)
