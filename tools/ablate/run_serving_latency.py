"""Latency of ONE call on few rows (serving one image or one request at a time), kernels through the Python wrappers:
host wall time per call with a synchronize after every call.    gpurun -- python tools/ablate/run_serving_latency.py"""
import gc, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runia_core_amd import _hip
gc.disable(); torch.manual_seed(0)

def t(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

dev = "cuda"
# kNN bank 50 000 x 2048
bank = torch.nn.functional.normalize(torch.randn(50000, 2048, device=dev), dim=1)
bank_state = _hip.knn_prepare_bank(bank)  # what a deployed KNN postprocessor holds (setup-time passes over the bank done once)
# Mahalanobis 2048-d, 10 classes
D, C = 2048, 10
a = torch.randn(D, D, dtype=torch.float64, device=dev); prec = (a @ a.T / D + torch.eye(D, dtype=torch.float64, device=dev)).contiguous()
pp = _hip.pack_weights(prec); cm = torch.randn(C, D, device=dev); mu_p = (cm.double() @ prec).contiguous()
# LaREM from latents (cfg2 shapes): folded weights 512 -> 256
m = torch.randn(512, 256, dtype=torch.float64, device=dev).contiguous() * 0.05; cvec = torch.randn(256, dtype=torch.float64, device=dev)
pm = _hip.pack_weights(m)
for n in (1, 2, 4, 8, 12, 16, 64, 128, 512):
    q = torch.nn.functional.normalize(torch.randn(n, 2048, device=dev), dim=1)
    f = torch.randn(n, D, device=dev)
    lg = torch.randn(n, 1000, device=dev)
    x = torch.relu(torch.randn(n, 512, 4, 4, device=dev)); rand = torch.rand(n, 16, 4, 4, device=dev)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    def larem():
        h = _hip.mc_entropy(x, rand, 16, 0.5, 2, 5)
        out.zero_()
        _hip.proj_sq_accumulate(h, pm, cvec, 256, out)
    print(f"rows {n:4d}:  kNN(k=50, bank 50000x2048) prepared bank {t(lambda: _hip.knn_kth(q, bank, 50, state=bank_state)):8.1f} us, "
          f"bank passes in the call {t(lambda: _hip.knn_kth(q, bank, 50)):8.1f} us   Mahalanobis(2048, 10 classes) "
          f"{t(lambda: _hip.mahalanobis_score(f, cm, pp, mu_p)):8.1f} us   Energy+MSP(1000) {t(lambda: _hip.row_lse_msp(lg, True, True)):7.1f} us   "
          f"LaREM from latents (K0+K1+K2') {t(larem):7.1f} us", flush=True)

# second table: the other stages of the path on few rows
tr = torch.randn(10000, 256, dtype=torch.float64, device=dev); kst = _hip.kde_pack_train(tr)
tr16 = torch.randn(10000, 16, dtype=torch.float64, device=dev)
from runia_core_amd.inference.postprocessors import DetectorKDE
det16 = DetectorKDE(tr16.cpu().numpy())
comp = torch.linalg.qr(torch.randn(512, 256, dtype=torch.float64, device=dev))[0].contiguous(); pct = _hip.pack_weights(comp)
bias = torch.randn(256, dtype=torch.float64, device=dev)
a2 = torch.randn(256, 256, dtype=torch.float64, device=dev); p2 = _hip.pack_weights((a2 @ a2.T / 256 + torch.eye(256, dtype=torch.float64, device=dev)).contiguous())
mean2 = torch.randn(256, dtype=torch.float64, device=dev)
w1000 = torch.randn(1000, 2048, device=dev); b1000 = torch.randn(1000, device=dev)
a3 = torch.randn(2048, 2048, dtype=torch.float64, device=dev); p3 = _hip.pack_weights((a3 @ a3.T / 2048 + torch.eye(2048, dtype=torch.float64, device=dev)).contiguous())
mean3 = torch.randn(2048, dtype=torch.float64, device=dev)
for n in (1, 8, 64, 512):
    x256 = torch.randn(n, 256, dtype=torch.float64, device=dev); x16 = torch.randn(n, 16, dtype=torch.float64, device=dev)
    h = torch.randn(n, 512, dtype=torch.float64, device=dev); f = torch.randn(n, 2048, device=dev)
    z = torch.randn(n * 16, 512, device=dev); x2048 = torch.randn(n, 2048, dtype=torch.float64, device=dev)
    print(f"rows {n:4d}:  LaRED(train 10000x256) {t(lambda: _hip.kde_score_packed(kst, x256, 1.0)):8.1f} us   LaRED(10000x16) direct "
          f"{t(lambda: _hip.kde_score(tr16, x16, 1.0)):7.1f} us, as DetectorKDE routes it {t(lambda: det16.score_samples_device(x16)):7.1f} us   PCA 512->256 {t(lambda: _hip.pca_transform(h, pct, bias, None, 256)):6.1f} us   "
          f"MD(256) {t(lambda: _hip.md_score(x256, mean2, p2)):6.1f} us   MD(2048) {t(lambda: _hip.md_score(x2048, mean3, p3)):7.1f} us   "
          f"linear 2048->1000 {t(lambda: _hip.linear(f, w1000, b1000)):6.1f} us   entropy per dim (16 x 512) {t(lambda: _hip.kl_entropy_per_dim(z, 16, 5)):6.1f} us",
          flush=True)
