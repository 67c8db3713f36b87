"""kNN on narrow features (PCA-reduced latents): f32 kernel + dense distance matrix vs bf16 pieces + candidate filter."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
lib = _hip.load_library()
torch.manual_seed(0)
k = 50
SIZES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or None
for (n, m, d) in SIZES or ((65536, 50000, 8), (65536, 50000, 16), (16384, 20000, 16), (65536, 50000, 32), (65536, 50000, 64), (65536, 50000, 128), (65536, 50000, 256), (8192, 20000, 64), (8192, 20000, 128), (2048, 8192, 128)):
    q = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
    b = torch.nn.functional.normalize(torch.randn(m, d, device="cuda"), dim=1)
    ws_bytes = lib.runia_knn_workspace_bytes(n, m, d, k)
    ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device="cuda")
    out = torch.empty(n, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.runia_knn_kth_f32(q.data_ptr(), b.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, n, m, d, k, st)
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): call()
    e1.record(); torch.cuda.synchronize()
    print(f"N {n:6d} M {m:6d} D {d:4d}: pieces {lib.runia_knn_piece_products(n, m, d)}  {e0.elapsed_time(e1) / 5:8.3f} ms  checksum {float(out.double().sum()):.6f}", flush=True)
