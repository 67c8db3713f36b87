import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
torch.manual_seed(0)
for n in (10, 64, 256, 512, 1024):
    a = torch.randn(n, n, dtype=torch.float64, device="cuda"); a = (a + a.T).contiguous()
    for _ in range(2): w, v = _hip.eigh(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5 if n <= 256 else 2
    for _ in range(reps): w, v = _hip.eigh(a)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ref = torch.linalg.eigvalsh(a.cpu())
    print(f"n={n:5d}: {ms:8.2f} ms   max |w - lapack| {float((w.cpu().sort().values - ref).abs().max()):.1e}", flush=True)
