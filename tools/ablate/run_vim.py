"""ViM residual norm at the f4 leg's shape (1 M x 2048 f32 rows, NS 2048 x 1048): runia_proj_norm_f32."""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
N, D = 1_000_000, 2048
x = torch.relu(torch.randn(N, D, device="cuda"))
u = torch.randn(D, device="cuda") * 0.1
for n in (1048, 1024, 1280):
    ns = torch.linalg.qr(torch.randn(D, n, dtype=torch.float64, device="cuda"))[0].contiguous()
    pk = _hip.pack_weights(ns)
    for _ in range(2): out = _hip.proj_norm(x, u, pk, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = _hip.proj_norm(x, u, pk, n)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"proj_norm {N} x {D} -> {n}: {ms:.2f} ms  {2.0*N*D*n/ms*1e-9:.1f} TFLOP/s of 2 D n   checksum {float(out.sum()):.9e}")
