import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
for (N, M, D, BW) in ((10000, 50000, 256, 8.0), (10000, 50000, 64, 8.0), (2000, 50000, 256, 8.0), (100000, 50000, 64, 8.0), (10000, 10000, 256, 8.0),
                      # sklearn's default bandwidth on whitened rows: the kernel sum is carried by a few neighbours
                      (10000, 50000, 256, 1.0), (100000, 4000, 256, 1.0), (10000, 50000, 64, 1.0)):
    tr = torch.randn(M, D, dtype=torch.float64, device="cuda"); x = torch.randn(N, D, dtype=torch.float64, device="cuda")
    st = _hip.kde_pack_train(tr)
    for _ in range(2): got = _hip.kde_score_packed(st, x, BW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): got = _hip.kde_score_packed(st, x, BW)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"N {N} M {M} D {D} bw {BW}: {ms:.3f} ms  {2.0 * N * M * D / ms * 1e-9:.1f} TFLOP/s  checksum {float(got.sum()):.9e}", flush=True)
