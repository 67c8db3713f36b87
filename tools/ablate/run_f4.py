"""f4 stages at cfg3-like sizes: linear head (React / DICE / ViM logits), ASH-S pruning, GEN, ViM residual norm."""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, D, C = 262144, 2048, 1000
x = torch.relu(torch.randn(N, D, device="cuda"))
w = torch.randn(C, D, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
ms = t(lambda: _hip.linear(x, w, b)); print(f"linear {N}x{D}->{C}: {ms:.3f} ms  {2.0*N*D*C/ms*1e-9:.1f} TFLOP/s")
ms = t(lambda: _hip.linear(x, w, b, 1.0)); print(f"linear clipped (ReAct): {ms:.3f} ms  {2.0*N*D*C/ms*1e-9:.1f} TFLOP/s")
ms = t(lambda: _hip.ash_s(x, 85)); print(f"ash_s {N}x{D}: {ms:.3f} ms  {2.0*N*D*4/ms*1e-6:.1f} GB/s")
lg = torch.randn(1_000_000, C, device="cuda")
ms = t(lambda: _hip.gen_score(lg, 0.1, 100)); print(f"gen 1Mx{C}, M=100: {ms:.3f} ms  {1e6*C*4/ms*1e-6:.1f} GB/s")
ms = t(lambda: _hip.gen_score(lg[:, :10].contiguous(), 0.1, 10)); print(f"gen 1Mx10: {ms:.3f} ms")
u = torch.randn(D, dtype=torch.float64, device="cuda"); ns = torch.linalg.qr(torch.randn(D, 1024, dtype=torch.float64, device="cuda"))[0]
pk = _hip.pack_weights(ns.contiguous())
ms = t(lambda: _hip.proj_norm(x[:65536], u.float(), pk, 1024)); print(f"vim proj_norm 65536x{D}->1024: {ms:.3f} ms  {2.0*65536*D*1024/ms*1e-9:.1f} TFLOP/s")
# small heads (CIFAR-10-sized)
w10 = torch.randn(10, 512, device="cuda") * 0.05; b10 = torch.randn(10, device="cuda")
x512 = torch.relu(torch.randn(1_000_000, 512, device="cuda"))
ms = t(lambda: _hip.linear(x512, w10, b10)); print(f"linear 1Mx512->10: {ms:.3f} ms  {1e6*512*4/ms*1e-6:.1f} GB/s read")
ms = t(lambda: _hip.ash_s(x512, 85)); print(f"ash_s 1Mx512: {ms:.3f} ms  {2.0*1e6*512*4/ms*1e-6:.1f} GB/s")
