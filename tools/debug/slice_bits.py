"""Do a row's bits depend on where it sits in a launch?  whole batch vs slices, per stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from runia_core_amd import _hip

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for C, r, n, cut in ((64, 16, 1001, 501), (64, 16, 1001, 500), (512, 256, 1001, 501), (512, 16, 1001, 501), (64, 64, 1001, 501), (128, 128, 1001, 501), (64, 16, 64, 32), (64,16,1001,512)):
    x = torch.relu(torch.randn(n, C, 4, 4, device=dev, generator=g)).contiguous()
    rand = torch.rand(n, 16, 4, 4, device=dev, generator=g); rand[:, :, 0, 0] = rand[:, :, 0, 0].clamp_min(0.2)
    rand = rand.contiguous()
    h = _hip.mc_entropy(x, rand, 16, 0.5, 2, 5)
    h2 = torch.cat([_hip.mc_entropy(x[:cut].contiguous(), rand[:cut].contiguous(), 16, 0.5, 2, 5), _hip.mc_entropy(x[cut:].contiguous(), rand[cut:].contiguous(), 16, 0.5, 2, 5)])
    m = torch.randn(C, r, dtype=torch.float64, device=dev, generator=g) * 0.1
    pm = _hip.pack_weights(m.contiguous())
    c = torch.randn(r, dtype=torch.float64, device=dev, generator=g)
    s = _hip.proj_sq_score(h, pm, c, r)
    s2 = torch.cat([_hip.proj_sq_score(h[:cut].contiguous(), pm, c, r), _hip.proj_sq_score(h[cut:].contiguous(), pm, c, r)])
    acc = torch.zeros(n, dtype=torch.float64, device=dev); _hip.proj_sq_accumulate(h, pm, c, r, acc)
    a1 = torch.zeros(cut, dtype=torch.float64, device=dev); _hip.proj_sq_accumulate(h[:cut].contiguous(), pm, c, r, a1)
    a2 = torch.zeros(n - cut, dtype=torch.float64, device=dev); _hip.proj_sq_accumulate(h[cut:].contiguous(), pm, c, r, a2)
    acc2 = torch.cat([a1, a2])
    def d(a, b):
        bad = (a != b).nonzero().flatten()
        return f"{bad.numel()} rows differ" + (f" first {bad[:6].tolist()} max {float((a-b).abs().max()):.3e}" if bad.numel() else "")
    print(f"C={C} r={r} n={n} cut={cut}: K1 {d(h, h2)} | proj_sq_score {d(s, s2)} | accumulate {d(acc, acc2)} | score vs accumulate {d(s, acc)}", flush=True)
