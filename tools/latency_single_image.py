"""Latency of the per-image API (LaRExInference.get_score tail: hooked latent -> host score), batch 1."""
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


class Backbone(torch.nn.Module):  # stands in for the frozen model: returns the hooked (1,512,4,4) map
    def __init__(self):
        super().__init__()
        self.layer4 = torch.nn.Identity()
    def forward(self, x):
        return self.layer4(x).mean(dim=(2, 3))

model = Backbone().eval()
hook = rc.Hook(model.layer4)
inf = rc.LaRExInference(model=model, postprocessor=md, mcd_sampler=rc.MCSamplerModule, pca_transform=pca,
                        mcd_samples_nro=16, drop_block_prob=0.5, drop_block_size=2, layer_type="Conv")
x, _ = bench.synth_latents(64, 5, 0.0, dev)
for i in range(5): inf.get_score(x[i:i+1], hook)
torch.cuda.synchronize()
ts = []
for i in range(64):
    t0 = time.perf_counter(); out, s = inf.get_score(x[i:i+1], hook); ts.append(time.perf_counter() - t0)
print(f"LaRExInference.get_score (batch 1, incl. CPU-generator draws + H2D of 256 floats + D2H of the score): "
      f"median {np.median(ts)*1e6:.0f} us, p90 {np.percentile(ts,90)*1e6:.0f} us, score {s}")
