import gc, os, sys, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
a = torch.randn(1_000_000, dtype=torch.float64, device="cuda") + 1.0
b = torch.randn(1_000_000, dtype=torch.float64, device="cuda")
for _ in range(10): r = _hip.ood_metrics(a, b)
torch.cuda.synchronize()
import time
a32, b32 = a.float(), b.float()
for x, y, name in ((a, b, "f64"), (a32, b32, "f32")):
    for _ in range(5): _hip.ood_metrics(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): r = _hip.ood_metrics(x, y)
    e1.record(); torch.cuda.synchronize()
    print(f"metrics 2M {name}: {e0.elapsed_time(e1) / 20:.3f} ms", r.cpu().numpy())
