import sys, os
sys.path.insert(0, os.getcwd())
import torch
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
torch.manual_seed(0)
for (N, n_mc, D) in ((10000, 16, 512), (10000, 16, 256), (2000, 32, 512), (4000, 8, 2048)):
    z = torch.randn(N * n_mc, D, device="cuda")
    for _ in range(2): _hip.kl_entropy_joint(z, n_mc, min(5, n_mc - 1))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): _hip.kl_entropy_joint(z, n_mc, min(5, n_mc - 1))
    e1.record(); torch.cuda.synchronize()
    print(N, n_mc, D, "%.3f ms" % (e0.elapsed_time(e1) / 10))
