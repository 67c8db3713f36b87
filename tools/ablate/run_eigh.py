"""Timing + accuracy of the device eigen-solver: blocked Jacobi (default) against the scalar-rotation form and LAPACK."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
torch.manual_seed(0)
for n in (10, 64, 256, 512, 1024, 2048):
    g = torch.randn(n, n, dtype=torch.float64, device="cuda")
    a = (g @ g.T / n + torch.eye(n, dtype=torch.float64, device="cuda")).contiguous()
    out = []
    for blocked in (True, False):
        if not blocked and n > 1024:
            out.append("scalar form: skipped")
            continue
        sweeps = [0]
        for _ in range(2): w, v = _hip.eigh(a, blocked=blocked)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5 if n <= 512 else 2
        for _ in range(reps): w, v = _hip.eigh(a, blocked=blocked)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        ref = torch.linalg.eigvalsh(a.cpu())
        res = float((a @ v - v * w).abs().max()); orth = float((v.T @ v - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max())
        out.append(f"{'blocked' if blocked else 'scalar '} {ms:8.2f} ms  |w-lapack| {float((w.cpu() - ref).abs().max()):.1e} |Av-vw| {res:.1e} |V'V-I| {orth:.1e}")
    t0 = time.perf_counter(); np.linalg.eigh(a.cpu().numpy()); host = (time.perf_counter() - t0) * 1e3
    print(f"n={n:5d}: " + " || ".join(out) + f" || host LAPACK {host:.1f} ms", flush=True)
