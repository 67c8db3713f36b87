"""Row GEMM stages at cfg4's shapes over batch sizes around whole rounds of 32-row tiles (last-round balancing).
PCA 1024 -> 256 and KDE of the reduced rows against 4 000 training rows; TFLOP/s of f64 MFMA (peak 78.6)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
dev = torch.device("cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
g = torch.Generator(device="cuda").manual_seed(0)
D, n, M = 1024, 256, 4000
comp = torch.linalg.qr(torch.randn(D, n, dtype=torch.float64, device=dev, generator=g))[0].contiguous()
pct = _hip.pack_weights(comp)
bias = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
scale = torch.rand(n, dtype=torch.float64, device=dev, generator=g) + 0.5
tr = torch.randn(M, n, dtype=torch.float64, device=dev, generator=g)
st = _hip.kde_pack_train(tr)
SIZES = [int(a) for a in sys.argv[1:]] or [16_384, 20_000, 32_768, 40_000, 65_536, 80_000, 98_304, 100_000, 114_688, 131_072, 150_000]
for N in SIZES:
    h = torch.randn(N, D, dtype=torch.float64, device=dev, generator=g)
    y = _hip.pca_transform(h, pct, bias, scale, n)
    tp = t(lambda: _hip.pca_transform(h, pct, bias, scale, n))
    tk = t(lambda: _hip.kde_score_packed(st, y, 16.0), reps=8)
    tk1 = t(lambda: _hip.kde_score_packed(st, y, 1.0), reps=8)
    print(f"N {N:7d}: PCA {tp * 1e6:8.1f} us {2 * N * D * n / tp / 1e12:6.1f} TF/s ({2 * N * D * n / tp / 78.6e12:.3f})   "
          f"KDE h=16 {tk * 1e6:8.1f} us ({2 * N * n * M / tk / 78.6e12:.3f})  h=1 {tk1 * 1e6:8.1f} us ({2 * N * n * M / tk1 / 78.6e12:.3f})", flush=True)
