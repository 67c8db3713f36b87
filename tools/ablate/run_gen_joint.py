"""The two HBM-streaming kernels round 5 left under 0.35 of HBM: GEN (1 M x 1000 logits, M = 100, gamma = 0.1) and the joint
k-NN entropy (10 000 images x 16 MC x 512; alone and with the per-dimension entropies from one read).  Timing, or the subject of
tools/pmc_cmd.sh / tools/prof_cmd.sh."""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
if os.environ.get('RUNIA_LIB'):
    _hip._LIB_PATH = os.environ['RUNIA_LIB']
gc.disable(); torch.manual_seed(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
if what in ("all", "gen"):
    N, C = 1_000_000, 1000
    g = torch.Generator(device="cuda").manual_seed(1)
    lg = torch.randn(N, C, device="cuda", generator=g) * 0.9
    for M in (100, 10, 1000):
        ms = t(lambda: _hip.gen_score(lg, 0.1, M))
        s = _hip.gen_score(lg, 0.1, M)
        print(f"gen 1M x {C}, M={M}: {ms:.3f} ms  {N*(C*4+4)/ms*1e-6:.0f} GB/s = {N*(C*4+4)/ms*1e-6/8000:.3f} of HBM   checksum {float(s.double().sum()):.9e}")
    pr = torch.softmax(lg[:262144], dim=1)
    ms = t(lambda: _hip.gen_entropy(pr, 0.1, 100))
    print(f"gen_entropy (probabilities) 262144 x {C}: {ms:.3f} ms  {262144*(C*4+4)/ms*1e-6:.0f} GB/s   checksum {float(_hip.gen_entropy(pr, 0.1, 100).double().sum()):.9e}")
if what in ("all", "joint"):
    N, n_mc, D = 10000, 16, 512
    z = torch.randn(N * n_mc, D, device="cuda")
    by = N * n_mc * D * 4
    ms = t(lambda: _hip.kl_entropy_joint(z, n_mc, 5)); print(f"joint {N} x {n_mc} x {D}: {ms:.4f} ms  {by/ms*1e-6/8000:.3f} of HBM")
    ms = t(lambda: _hip.kl_entropy_per_dim(z, n_mc, 5)); print(f"per-dim: {ms:.4f} ms  {(by + N*D*8)/ms*1e-6/8000:.3f} of HBM")
    ms = t(lambda: _hip.kl_entropy_both(z, n_mc, 5)); print(f"both from one read: {ms:.4f} ms  {(by + N*D*8)/ms*1e-6/8000:.3f} of HBM")
    a, b = _hip.kl_entropy_both(z, n_mc, 5)
    print(f"   checksums joint {float(a.sum()):.12e} per-dim {float(b.sum()):.12e}")
