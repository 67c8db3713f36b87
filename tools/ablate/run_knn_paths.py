"""kNN entry point with the bf16-piece workspace vs the f32-only workspace (same scores, two candidate kernels) around the
sizes from which the bf16 kernel is taken: is the switch point right?   gpurun -- python tools/ablate/run_knn_paths.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
lib = _hip.load_library()
torch.manual_seed(0)
k = 50
for (n, m, d) in ((256, 20000, 2048), (256, 50000, 2048), (512, 50000, 2048), (1000, 20000, 2048), (600, 8192, 512), (1024, 4096, 256), (1024, 4096, 2048), (1024, 50000, 256), (2048, 8192, 512), (4096, 16384, 1024),
                  (8192, 50000, 256), (1024, 50000, 2048), (8192, 50000, 2048), (32768, 200000, 512)):
    q = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
    b = torch.nn.functional.normalize(torch.randn(m, d, device="cuda"), dim=1)
    full = lib.runia_knn_workspace_bytes(n, m, d, k)
    qc = min(n, 8192, max(256, (1 << 31) // (4 * m)))
    f32_only = (qc * m + qc + m + 4) * 4  # the f32 kernel's workspace: below what the bf16 kernel asks for
    res = []
    for ws_bytes in (full, f32_only):
        ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device="cuda")
        out = torch.empty(n, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.runia_knn_kth_f32(q.data_ptr(), b.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, n, m, d, k, st)
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / reps, out.clone()))
        del ws
    same = bool(torch.equal(res[0][1], res[1][1]))
    print(f"N {n:6d} M {m:6d} D {d:5d}: pieces {lib.runia_knn_piece_products(n, m, d)}  bf16-workspace {res[0][0]:8.3f} ms   f32-workspace {res[1][0]:8.3f} ms   x{res[1][0] / res[0][0]:.2f}  same bits {same}", flush=True)
