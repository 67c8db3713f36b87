import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
torch.manual_seed(0)
m, d, n = 50000, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 64
bank = torch.nn.functional.normalize(torch.randn(m, d, device="cuda"), dim=1)
st = _hip.knn_prepare_bank(bank)
q = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
for _ in range(30): _hip.knn_kth(q, bank, 50, state=st)
torch.cuda.synchronize()
