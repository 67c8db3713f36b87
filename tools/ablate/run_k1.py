"""Time K1 (mc_entropy) variants built with -DRUNIA_ABLATE=<n> (tools/ablate/build.sh)."""
import ctypes, gc, glob, os, sys, torch
gc.disable()
from ctypes import c_void_p, c_int64, c_int, c_double, c_size_t
here = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
n, C, H, W, n_mc = 10000, 512, 4, 4, 16
x = torch.relu(torch.randn(n, C, H, W, device="cuda")).contiguous()
rand = torch.rand(n, n_mc, H, W, device="cuda")
h = torch.empty(n, C, dtype=torch.float64, device="cuda")
for so in sorted(glob.glob(os.path.join(here, os.environ.get("K1_GLOB", "libk1_*.so")))):
    lib = ctypes.CDLL(so)
    f = lib.runia_mc_entropy_f32
    f.restype = c_int
    f.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_double, c_void_p]
    lib.runia_mc_entropy_workspace_bytes.restype = c_size_t
    lib.runia_mc_entropy_workspace_bytes.argtypes = [c_int64, c_int, c_int, c_int]
    wsb = lib.runia_mc_entropy_workspace_bytes(n, H, W, n_mc)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    def call():
        rc = f(x.data_ptr(), rand.data_ptr(), n_mc * H * W, h.data_ptr(), None, None, ws.data_ptr(), wsb, n, C, H, W, n_mc, 0.5, 2, 5, 1e-5, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:  # working clocks first
        for _ in range(20): call()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): call()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(so), "%.1f us" % (e0.elapsed_time(e1) * 5), "checksum", float(torch.nan_to_num(h).sum()))
