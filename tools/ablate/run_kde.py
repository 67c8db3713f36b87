import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
for (N, M, D) in ((8192, 10000, 16), (8192, 10000, 32), (8192, 10000, 64), (65536, 10000, 16), (65536, 10000, 32)):
    tr = torch.randn(M, D, dtype=torch.float64, device="cuda"); x = torch.randn(N, D, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        _hip.kde_score(tr, x, 1.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): _hip.kde_score(tr, x, 1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"N {N} M {M} D {D}: {ms:.3f} ms  {N * M / ms * 1e-6:.1f} G pairs/s")
    if D >= 16:
        st = _hip.kde_pack_train(tr)
        ref = _hip.kde_score(tr, x, 1.0)
        for _ in range(3): got = _hip.kde_score_packed(st, x, 1.0)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10): got = _hip.kde_score_packed(st, x, 1.0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"   matrix-core path: {ms:.3f} ms  {N * M / ms * 1e-6:.1f} G pairs/s  {2.0 * N * M * D / ms * 1e-9:.1f} TFLOP/s  max |diff| {float((got - ref).abs().max()):.2e}")
