"""Per-dimension entropy of MC samples at the cfg4_lared shape: 100 000 x 16 x 1024 f32 -> [100 000, 1024] f64."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
g = torch.Generator(device="cuda").manual_seed(0)
for n, n_mc, d in ((100_000, 16, 1024), (100_000, 8, 1024), (20_000, 32, 1024)):
    z = torch.randn(n * n_mc, d, device="cuda", generator=g)
    for _ in range(3): h = _hip.kl_entropy_per_dim(z, n_mc, 5 if n_mc > 8 else 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): h = _hip.kl_entropy_per_dim(z, n_mc, 5 if n_mc > 8 else 4)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = z.numel() * 4 + h.numel() * 8
    print(f"{n} x {n_mc} x {d}: {ms:.3f} ms  {byt / ms / 1e9:.2f} TB/s ({byt / ms / 1e9 / 8:.3f} of 8)  checksum {float(h.sum()):.9e}", flush=True)
