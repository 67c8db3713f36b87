"""cfg4 per-ROI path: roi_align + sampler + entropy, two calls against the fused launch (60 000 ROIs x 256 ch x 7x7)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B, C, H, W = 60, 256, 50, 80
g = torch.Generator(device="cuda").manual_seed(3)
fm = torch.relu(torch.randn(B, C, H, W, device="cuda", generator=g))
cg = torch.Generator().manual_seed(1)
xy = torch.rand(K, 2, generator=cg) * torch.tensor([400.0, 250.0])
wh = 30 + torch.rand(K, 2, generator=cg) * torch.tensor([200.0, 120.0])
boxes = torch.cat([xy, xy + wh], dim=1).cuda()
bidx = (torch.arange(K) % B).to(torch.int32).cuda()
rand = torch.rand(K, n_mc, 7, 7, device="cuda", generator=g)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
def two_calls():
    hs = []
    for k0 in range(0, K, 60000):
        rois = _hip.roi_align(fm, boxes[k0:k0 + 60000], 7, W / 640.0, 2, True, bidx[k0:k0 + 60000])
        hs.append(_hip.mc_entropy(rois, rand[k0:k0 + 60000], n_mc, 0.4, 3, 5))
    return torch.cat(hs)
nhwc = _hip.nchw_to_nhwc(fm)
def fused():
    return _hip.roi_mc_entropy(nhwc, boxes, 7, W / 640.0, 2, True, rand, n_mc, 0.4, 3, 5, batch_idx=bidx)
t_tr, _ = timed(lambda: _hip.nchw_to_nhwc(fm))
t2, h2 = timed(two_calls)
t1, h1 = timed(fused)
same = torch.equal(torch.nan_to_num(h1, nan=-7.0), torch.nan_to_num(h2, nan=-7.0))
byt = K * C * 49 * 4
print(f"K {K} n_mc {n_mc}: two calls {t2:.3f} ms, fused {t1:.3f} ms (+ {t_tr:.3f} ms channels-last copy of {B} maps), same bits {same}; "
      f"fused = {K / t1 * 1e3 / 1e6:.2f} M ROIs/s, {byt / t1 / 1e6:.0f} GB/s of the (K, C, 7, 7) tensor never written")
