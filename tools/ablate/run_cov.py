"""Covariance (Gram matrix on the f64 matrix cores) at the fit shapes: cfg3 (50 000 x 2048 f32), cfg2 (50 000 x 512 f64),
cfg4 (100 000 x 1024 f64).   gpurun -- python tools/ablate/run_cov.py"""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
def t(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n, d, dt in ((50000, 2048, torch.float32), (50000, 512, torch.float64), (100000, 1024, torch.float64), (50000, 256, torch.float64)):
    x = torch.randn(n, d, device="cuda", dtype=dt)
    ms = t(lambda: _hip.covariance(x))
    print(f"covariance {n} x {d} {str(dt)[6:]}: {ms:.3f} ms   {2.0 * n * d * d / ms * 1e-9:.1f} TFLOP/s of the full square "
          f"({n * d * (d + 128) / ms * 1e-9:.1f} executed on the upper triangle of 128-tiles)", flush=True)
