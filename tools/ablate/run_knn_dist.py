"""Candidate-distance kernel of one 8 192-query chunk alone (bf16 piece products), for the library named by RUNIA_LIB."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
lib = _hip.load_library()
Q, M, D, k = 8192, 50000, 2048, 50
torch.manual_seed(0)
bank = torch.nn.functional.normalize(torch.randn(M, D, device="cuda"), dim=1)
q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=1)
for _ in range(2): _hip.knn_kth(q, bank, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): _hip.knn_kth(q, bank, k)
e1.record(); torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('RUNIA_LIB', 'shipped')):22s} kNN call {e0.elapsed_time(e1) / 10:.3f} ms per 8192 queries")
