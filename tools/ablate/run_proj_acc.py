"""K2' accumulate form (runia_proj_sq_accumulate_f64, what bench.py's step uses) for the library named by RUNIA_LIB."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
torch.manual_seed(0)
gc.disable()
lib = _hip.load_library()
for N in ([int(v) for v in sys.argv[1:]] or [10000, 4000, 20000]):
    D, r = 512, 256
    h = torch.randn(N, D, dtype=torch.float64, device="cuda")
    M = torch.randn(D, r, dtype=torch.float64, device="cuda") * 0.05
    c = torch.randn(r, dtype=torch.float64, device="cuda")
    pm = _hip.pack_weights(M)
    out = torch.zeros(N, dtype=torch.float64, device="cuda")
    ref = -((h @ M + c) ** 2).sum(1)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = lib.runia_proj_sq_accumulate_f64(h.data_ptr(), pm.data_ptr(), c.data_ptr(), out.data_ptr(), N, D, r, st)
        assert rc == 0
    call(); torch.cuda.synchronize()
    err = float(((out - ref).abs() / ref.abs()).max())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(50): call()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 5
    print(f"{os.path.basename(os.environ.get('RUNIA_LIB', 'shipped')):18s} N {N:6d} {us:.1f} us  {2.0 * N * D * r / us * 1e-6:.1f} TFLOP/s  err {err:.1e}", flush=True)
