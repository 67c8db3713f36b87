import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
for (Q, M, D, k) in ((10000, 50000, 16, 50), (10000, 50000, 64, 50), (10000, 50000, 256, 50), (100, 50000, 2048, 50), (10000, 1000, 512, 5), (1, 50000, 2048, 50)):
    bank = torch.nn.functional.normalize(torch.randn(M, D, device="cuda"), dim=1)
    q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=1)
    for _ in range(3): s = _hip.knn_kth(q, bank, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): s = _hip.knn_kth(q, bank, k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"kNN Q {Q} M {M} D {D} k {k}: {ms:.3f} ms  {2.0*Q*M*D/ms*1e-9:.1f} TFLOP/s  dist matrix {Q*M*4/ms*1e-6:.0f} GB/s-equivalent", flush=True)
