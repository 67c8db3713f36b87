"""Which rows of the candidate-filter path differ from the f32 kernel (debug aid)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip as hip
if os.environ.get("RUNIA_LIB"): hip._LIB_PATH = os.environ["RUNIA_LIB"]
lib = hip.load_library()
n, m, d, k = 9000, 8192, 256, 50
rng = np.random.default_rng(7)
bank = rng.standard_normal((m, d)).astype(np.float32)
bank /= np.linalg.norm(bank, axis=1, keepdims=True)
bank[1000:4000] = bank[17]
q = rng.standard_normal((n, d)).astype(np.float32)
q[:4500] = bank[17] + 0.05 * q[:4500] / np.sqrt(d)
q /= np.linalg.norm(q, axis=1, keepdims=True)
qd, bd = torch.from_numpy(q).cuda(), torch.from_numpy(bank).cuda()
def run(big, nn=n):
    full = lib.runia_knn_workspace_bytes(nn, m, d, k)
    f32_only = (min(nn, 8192) * m + min(nn, 8192) + m + 4) * 4
    ws_bytes = full if big else f32_only
    ws = torch.zeros(ws_bytes // 4 + 1, dtype=torch.float32, device="cuda")
    out = torch.full((nn,), 123.0, device="cuda")
    rc = lib.runia_knn_kth_f32(qd.data_ptr(), bd.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, nn, m, d, k,
                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out.cpu().numpy()
for nn in (9000, 8192, 4500, 2000):
    a, b = run(True, nn), run(False, nn)
    bad = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
    print("n", nn, "mismatches", bad.size, "first", bad[:12], "last", bad[-5:])
    for i in bad[:6]:
        print("   row", i, "filter", a[i], "f32", b[i])
