import sys, os, time, gc
sys.path.insert(0, os.getcwd())
import torch
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable()
torch.manual_seed(0)
Q, M, D, k = 32768, 50000, 2048, 50
bank = torch.nn.functional.normalize(torch.randn(M, D, device="cuda"), dim=1)
q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=1)
for _ in range(3): s = _hip.knn_kth(q, bank, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): s = _hip.knn_kth(q, bank, k)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("kNN %.2f ms  %.1f TFLOP/s" % (ms, 2.0 * Q * M * D / ms * 1e-9))
