import os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
lib = _hip.load_library()
torch.manual_seed(0)
n, m, d, k = 65536, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 50
q = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
b = torch.nn.functional.normalize(torch.randn(m, d, device="cuda"), dim=1)
ws_bytes = lib.runia_knn_workspace_bytes(n, m, d, k)
ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device="cuda")
out = torch.empty(n, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5): assert lib.runia_knn_kth_f32(q.data_ptr(), b.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, n, m, d, k, st) == 0
torch.cuda.synchronize()
