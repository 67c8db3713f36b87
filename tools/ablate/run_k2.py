"""Time K2 (pca_md) variants built with -DRUNIA_ABLATE_K2=<n>."""
import ctypes, glob, os, sys, torch
from ctypes import c_void_p, c_int64, c_int
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from runia_core_amd import _hip
torch.manual_seed(0)
n, D, nc = int(os.environ.get("N", 10000)), 512, 256
h = torch.randn(n, D, dtype=torch.float64, device="cuda")
ct = _hip.pack_weights(torch.randn(D, nc, dtype=torch.float64, device="cuda") / 22)
pp = _hip.pack_weights(torch.eye(nc, dtype=torch.float64, device="cuda") + 0.01)
bias = torch.randn(nc, dtype=torch.float64, device="cuda"); scale = torch.rand(nc, dtype=torch.float64, device="cuda") + 0.5
mean = torch.randn(nc, dtype=torch.float64, device="cuda")
s = torch.empty(n, dtype=torch.float64, device="cuda")
for so in sorted(glob.glob(os.path.join(here, "libk2_*.so"))):
    lib = ctypes.CDLL(so)
    f = lib.runia_pca_md_score_f64
    f.restype = c_int
    f.argtypes = [c_void_p] * 8 + [c_int64, c_int64, c_int64, c_void_p]
    def call():
        rc = f(h.data_ptr(), ct.data_ptr(), bias.data_ptr(), scale.data_ptr(), mean.data_ptr(), pp.data_ptr(), s.data_ptr(), None, n, D, nc, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(so), "%.1f us" % (e0.elapsed_time(e1) * 50), "checksum", float(s.sum()))
