import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from runia_core_amd import _hip
z = np.random.default_rng(0).standard_normal((160000, 512)).astype(np.float32)   # 327 MB, pageable
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("torch.from_numpy(z).cuda():        %.2f ms" % t(lambda: torch.from_numpy(z).cuda()))
print("_hip.to_device(z, f32):            %.2f ms" % t(lambda: _hip.to_device(z, torch.float32)))
pin = torch.empty(z.shape, dtype=torch.float32).pin_memory()
print("copy into pinned + async H2D:      %.2f ms" % t(lambda: (pin.copy_(torch.from_numpy(z)), pin.cuda(non_blocking=True))[1]))
print("pinned -> device only:             %.2f ms" % t(lambda: pin.cuda(non_blocking=True)))
d = torch.from_numpy(z).cuda()
h = _hip.kl_entropy_per_dim(d, 16, 5)
print("entropy per dim (device):          %.3f ms" % t(lambda: _hip.kl_entropy_per_dim(d, 16, 5)))
print("h (10000x512 f64) .cpu().numpy():  %.2f ms" % t(lambda: h.cpu().numpy()))
import runia_core_amd as rc
print("get_dl_h_z(z numpy, 16):           %.2f ms" % t(lambda: rc.get_dl_h_z(z, 16)))
print("get_dl_h_z(device tensor, 16):     %.2f ms" % t(lambda: rc.get_dl_h_z(d, 16)))
