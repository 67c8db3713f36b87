"""PCIe-inclusive rate of the host-buffer boundary (ndarray in, ndarray out) for the bench workload."""
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
z_host = pipe.stack(x, rand).cpu().numpy()          # (160000, 512) f32 MC samples on the host
x_host, r_host = x.cpu().numpy(), rand.cpu().numpy()
for name, fn in (
    ("host MC samples -> get_dl_h_z -> apply_pca_transform -> MD.postprocess (3 host round trips, as the reference API)",
     lambda: md.postprocess(rc.apply_pca_transform(rc.get_dl_h_z(z_host, 16)[1], pca))),
    ("host MC samples -> LaREMPipeline.score_samples_host (one H2D, one D2H)", lambda: pipe.score_samples_host(z_host)),
    ("host latents+draws -> device chain -> host scores",
     lambda: pipe.score_latents(torch.from_numpy(x_host).cuda(), torch.from_numpy(r_host).cuda()).cpu().numpy()),
):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt*1e3:.2f} ms per 10000 images = {10000/dt:,.0f} images/s")
