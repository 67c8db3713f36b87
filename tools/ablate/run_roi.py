import os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
fm = torch.relu(torch.randn(1, 256, 50, 80, device=dev, generator=g))
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for kb in (1000, 10000, 60000):
    xy = torch.rand(kb, 2, device=dev, generator=g) * torch.tensor([400.0, 250.0], device=dev)
    wh = 30 + torch.rand(kb, 2, device=dev, generator=g) * torch.tensor([200.0, 120.0], device=dev)
    boxes = torch.cat([xy, xy + wh], dim=1)
    for sr in (2, 0):
        ms = t(lambda: _hip.roi_align(fm, boxes, 7, 80 / 640, sr, True), reps=5 if kb > 10000 else 20)
        print(f"roi_align {kb} boxes x 256 ch x 7x7, sampling_ratio {sr}: {ms*1e3:.1f} us = {kb/ms/1e3:.2f} M boxes/s, output {kb*256*49*4/ms/1e6:.0f} GB/s", flush=True)
