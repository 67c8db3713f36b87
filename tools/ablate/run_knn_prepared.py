"""kNN against a prepared bank on some tens to hundreds of queries: bf16 kernel (padded 256-row query tile) vs f32 kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip, config
torch.manual_seed(0)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (m, d) in ((50000, 2048), (20000, 2048), (20000, 512), (8192, 512)):
    bank = torch.nn.functional.normalize(torch.randn(m, d, device="cuda"), dim=1)
    st = _hip.knn_prepare_bank(bank)
    for n in (16, 64, 100, 256, 511, 1000):
        q = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
        a = _hip.knn_kth(q, bank, 50, state=st); us16 = t(lambda: _hip.knn_kth(q, bank, 50, state=st))
        config.knn_bf16_candidates = False
        b = _hip.knn_kth(q, bank, 50, state=st); us32 = t(lambda: _hip.knn_kth(q, bank, 50, state=st))
        config.knn_bf16_candidates = True
        print(f"bank {m} x {d}, {n:5d} queries: default {us16:8.1f} us   f32 kernel {us32:8.1f} us   same bits {bool(torch.equal(a, b))}", flush=True)
