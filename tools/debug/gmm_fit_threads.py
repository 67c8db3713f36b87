"""Host-side gmm_fit (float32 torch on the CPU, as the reference) against the intra-op thread count on the GPU box."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd.inference.funcs import gmm_fit
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((50000, 512)).astype(np.float32)); lab = torch.from_numpy(rng.integers(0, 10, 50000))
for nt in (torch.get_num_threads(), 64, 32, 16, 8, 4):
    torch.set_num_threads(nt)
    gmm_fit(x, lab, 10)
    t0 = time.perf_counter()
    for _ in range(3): gmm_fit(x, lab, 10)
    print(f"threads {nt:4d}: {(time.perf_counter() - t0) / 3 * 1e3:8.1f} ms per fit", flush=True)
