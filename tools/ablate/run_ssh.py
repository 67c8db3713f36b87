import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
import runia_core_amd as rc
from runia_core_amd import _hip
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
z_host = pipe.stack(x, rand).cpu().numpy()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("to_device(z_host): %.2f ms" % t(lambda: _hip.to_device(z_host, torch.float32)), z_host.flags['C_CONTIGUOUS'], z_host.dtype, z_host.shape)
zd = _hip.to_device(z_host, torch.float32)
print("entropy(zd): %.3f ms" % t(lambda: pipe.entropy(zd)))
h = pipe.entropy(zd)
print("score_entropies(h): %.3f ms" % t(lambda: pipe.score_entropies(h)))
print("score_samples(zd): %.3f ms" % t(lambda: pipe.score_samples(zd)))
print("score_samples_host(z_host): %.2f ms" % t(lambda: pipe.score_samples_host(z_host)))
