import gc, os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
gc.disable(); torch.manual_seed(0)
M, D, k = 50000, 2048, 50
bank = torch.nn.functional.normalize(torch.randn(M, D, device="cuda"), dim=1)
for Q in (1, 2, 4, 8, 9, 100):
    q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=1)
    for _ in range(20): s = _hip.knn_kth(q, bank, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): s = _hip.knn_kth(q, bank, k)
    e1.record(); torch.cuda.synchronize()
    print(f"{Q:4d} queries x bank {M} x {D}, k = {k}: {e0.elapsed_time(e1) / 50:.3f} ms per call", flush=True)
