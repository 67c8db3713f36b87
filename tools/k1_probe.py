"""K0 + K1 of the bench workload on their own (10 000 images x 512 channels x 4x4, 16 drop layers), three rotating input
sets: HIP-event time of K1 alone and of K0 + K1; the profiling target of the `rocprofv3 --pmc` passes in profiles/.
  python3 tools/k1_probe.py [--iters 300] [--counter]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):  # an ablation build of the library (tools/ablate)
    _hip._LIB_PATH = os.environ['RUNIA_LIB']

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--counter", action="store_true")
ap.add_argument("--warm", type=float, default=0.5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, n_mc = 10000, 16
sets = [bench.synth_latents(n, 1235 + 1000 * j, 0.0, dev) for j in range(3)]
lib = _hip.load_library()
wsb = int(lib.runia_mc_entropy_workspace_bytes(n, 4, 4, n_mc))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
h = torch.empty(n, 512, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream

def k0(j, i):
    x, r = sets[j]
    if a.counter:
        rc = lib.runia_mc_mask_table_counter_f32(7, i * n, ws.data_ptr(), wsb, n, 4, 4, n_mc, 0.5, 2, st)
    else:
        rc = lib.runia_mc_mask_table_f32(r.data_ptr(), n_mc * 16, ws.data_ptr(), wsb, n, 4, 4, n_mc, 0.5, 2, st)
    assert rc == 0

def k1(j):
    x, _ = sets[j]
    rc = lib.runia_mc_entropy_from_table_f32(x.data_ptr(), ws.data_ptr(), wsb, h.data_ptr(), None, None, n, 512, 4, 4, n_mc, 5, 1e-5, st)
    assert rc == 0

t0 = time.perf_counter()
while time.perf_counter() - t0 < a.warm:
    for i in range(30):
        k0(i % 3, i); k1(i % 3)
    torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
for i in range(a.iters):
    e0, e1, e2 = ev[i]
    e0.record(); k0(i % 3, i); e1.record(); k1(i % 3); e2.record()
torch.cuda.synchronize()
import numpy as np
t_k0 = np.array([e0.elapsed_time(e1) for e0, e1, _ in ev]); t_k1 = np.array([e1.elapsed_time(e2) for _, e1, e2 in ev])
print(f"K0 median {np.median(t_k0) * 1e3:.1f} us   K1 median {np.median(t_k1) * 1e3:.1f} us  (min {t_k1.min() * 1e3:.1f}, p90 {np.percentile(t_k1, 90) * 1e3:.1f})  "
      f"checksum {float(torch.nan_to_num(h).sum()):.6f}")
