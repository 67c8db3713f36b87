import os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
dev = torch.device("cuda")
def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps): fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for d in (2, 8, 11, 16, 23):
    tr = torch.randn(10000, d, dtype=torch.float64, device=dev)
    st = _hip.kde_pack_train(tr)
    for n in (1, 64, 512):
        x = torch.randn(n, d, dtype=torch.float64, device=dev)
        a, b = _hip.kde_score(tr, x, 1.0), _hip.kde_score_packed(st, x, 1.0)
        print(f"D {d:3d} rows {n:4d}: direct {t(lambda: _hip.kde_score(tr, x, 1.0)):7.1f} us  packed {t(lambda: _hip.kde_score_packed(st, x, 1.0)):7.1f} us  max |diff| {float((a - b).abs().max()):.2e}")
