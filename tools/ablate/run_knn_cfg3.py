"""kNN stage at cfg3's shape through the Python entry (normaliser excluded): N queries against 50 000 x 2048, k = 50."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
lib = _hip.load_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
M, D, k = 50000, 2048, 50
torch.manual_seed(0)
g = torch.Generator(device="cuda").manual_seed(1)
mu = torch.randn(10, D, device="cuda", generator=g) * 0.5
def feats(n):
    lab = torch.randint(0, 10, (n,), device="cuda", generator=g)
    return torch.relu(mu[lab] + torch.randn(n, D, device="cuda", generator=g))
bank = torch.nn.functional.normalize(feats(M), dim=1)
q = torch.nn.functional.normalize(feats(N), dim=1)
state = _hip.knn_prepare_bank(bank)
for _ in range(2): r = _hip.knn_kth(q, bank, k, state=state)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 3
e0.record()
for _ in range(reps): r = _hip.knn_kth(q, bank, k, state=state)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{os.path.basename(os.environ.get('RUNIA_LIB', 'shipped')):22s} N {N}: {ms:.3f} ms = {ms / (N / 8192):.3f} ms per 8192 queries, "
      f"{N / ms * 1e3 / 1e6:.3f} M queries/s, {3 * 2 * N * M * D / ms / 1e12:.1f} TFLOP/s executed; checksum {float(r.double().sum()):.6f}")
