#!/usr/bin/env python3
"""Per-stage throughput of every C-ABI stage at BASELINE.json sizes (cfg2 / cfg3), against its roofline,
with the CPU oracle timed beside it on a bounded sample.  Writes gpurun_out/stages.{json,md}.

    gpurun -- python tools/bench_stages.py            (one MI355X)
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (CPU baseline / checker only)
from runia_core_amd import _hip  # noqa: E402

HBM, F32_MFMA, F64_MFMA = 8000.0, 157.3, 78.6  # GB/s, TFLOP/s (MI355X_MICROARCH.md; f64 matrix: AMD datasheet)
dev = torch.device("cuda", 0)
rows = []


def gpu_ms(fn, reps=7):
    """Median time of one call in steady state.  Short stages are timed as back-to-back batches of calls (~3 ms per
    batch) between one pair of events: a lone 50 us launch bracketed by events and a host synchronisation measures the
    launch gap and an idling GPU's clocks, not the kernel (the MD stage read 0.216 ms that way; its kernel takes 0.052)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    inner = int(max(1, min(200, 3.0 / max(e0.elapsed_time(e1), 1e-3))))
    # warm the stage up for ~0.25 s first: after an idle spell (a CPU-oracle timing, a gen-2 GC pass) the GPU takes
    # tens of ms of sustained load to return to its working clocks
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for _ in range(inner):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ts))


def cpu_rate(fn, units):
    t0 = time.perf_counter()
    fn()
    return units / (time.perf_counter() - t0)


def add(stage, shape, unit, n_units, ms, bound, work_per_unit, cpu, cpu_note, err):
    """bound: "hbm" (work_per_unit = bytes), "mfma_f32" / "mfma_f64" (FLOP), or - for stages whose time is neither - a
    free-text limiter ("vector ALU: one f64 exp per pair", "launch latency: 2(n-1) launches per sweep" ...): the achieved
    rate is then still shown in the unit of work_per_unit, with no fraction of an unrelated peak."""
    rate = n_units / (ms * 1e-3)
    if bound == "hbm":
        ach, peak, u = work_per_unit * rate / 1e9, HBM, "GB/s"
    elif bound in ("mfma_f32", "mfma_f64", "mfma_bf16"):
        ach, peak, u = work_per_unit * rate / 1e12, {"mfma_f32": F32_MFMA, "mfma_f64": F64_MFMA, "mfma_bf16": 2500.0}[bound], "TFLOP/s"
    else:
        ach, peak, u = work_per_unit * rate / 1e9, None, "G work-units/s"
    frac = None if peak is None else round(ach / peak, 4)
    rows.append(dict(stage=stage, shape=shape, unit=unit, gpu_ms=round(ms, 4), gpu_rate=rate, bound=bound,
                     work_per_unit=work_per_unit, achieved=round(ach, 2), peak=peak, ach_unit=u,
                     frac=frac, cpu_rate=cpu, cpu_note=cpu_note, max_rel_err=err))
    print(f"{stage:28s} {shape:34s} {ms:9.3f} ms  {rate:12.4g} {unit}/s  {ach:8.1f} {u} ({'-' if frac is None else format(frac, '5.1%')})  cpu {cpu:10.4g}/s  err {err:.1e}",
          flush=True)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


g = torch.Generator(device=dev).manual_seed(3)

# ---- a7 Energy / MSP -----------------------------------------------------------------------------
for n, c in ((1_000_000, 1000), (1_000_000, 10)):
    x = torch.randn(n, c, device=dev, generator=g) * 3
    ms = gpu_ms(lambda: _hip.row_lse_msp(x, True, True))
    lse, msp = _hip.row_lse_msp(x, True, True)
    m = 20000 if c == 1000 else 400000
    xs = x[:m].cpu().numpy()
    cpu = cpu_rate(lambda: (oracle.energy_score(xs), oracle.msp_score(xs)), m)
    err = max(rel(lse[:m].cpu().numpy(), oracle.energy_score(xs)), rel(msp[:m].cpu().numpy(), oracle.msp_score(xs)))
    add("energy+msp (a7)", f"{n}x{c} f32", "rows", n, ms, "hbm", 4 * c + 8, cpu, f"scipy logsumexp+softmax, {m} rows", err)
    del x

# ---- a8 normaliser + kNN ------------------------------------------------------------------------
M, D, K, NQ = 50_000, 2048, 50, 32768
bank = torch.randn(M, D, device=dev, generator=g)
q = torch.randn(NQ, D, device=dev, generator=g)
ms = gpu_ms(lambda: _hip.l2_normalize(bank))
bank_n, q_n = _hip.l2_normalize(bank), _hip.l2_normalize(q)
xs = bank[:20000].cpu().numpy()
cpu = cpu_rate(lambda: oracle.normalizer(xs), 20000)
add("normalizer (a8)", f"{M}x{D} f32", "rows", M, ms, "hbm", 8 * D, cpu, "numpy, 20000 rows", rel(bank_n[:20000].cpu().numpy(), oracle.normalizer(xs)))
ms = gpu_ms(lambda: _hip.knn_kth(q_n, bank_n, K), reps=3)
s = _hip.knn_kth(q_n, bank_n, K)
mq = 8
bn_host = bank_n.cpu().numpy()
qs = q[:mq].cpu().numpy()
cpu = cpu_rate(lambda: oracle.knn_kth_score(bn_host, qs, K, chunk=1), mq)
err = rel(s[:mq].cpu().numpy(), oracle.knn_kth_score(bn_host, qs, K, chunk=1))
pieces = int(_hip.load_library().runia_knn_piece_products(NQ, M, D))  # bf16 piece products per distance (0: f32 kernel)
add("kNN k=50 (a8)" + (f", {pieces} bf16 piece products" if pieces else ""), f"{NQ} q x bank {M}x{D}", "queries", NQ, ms,
    "mfma_bf16" if pieces else "mfma_f32", max(pieces, 1) * 2.0 * M * D, cpu, f"one query at a time, exact f32 differences, {mq} queries", err)
del bank, q, bank_n, q_n

# ---- a6 Mahalanobis ---------------------------------------------------------------------------------
N, D, C = 262_144, 2048, 10
centres = torch.randn(C, D, device=dev, generator=g) * 0.5
lab = torch.randint(0, C, (N,), device=dev, generator=g)
f = torch.relu(centres[lab] + torch.randn(N, D, device=dev, generator=g))
tr = f[:6000].cpu().numpy()
cm, prec = oracle.mahalanobis_setup(tr, lab[:6000].cpu().numpy(), C)
packed = _hip.pack_weights(torch.from_numpy(prec).to(dev))
mu_p = torch.from_numpy(cm.astype(np.float64) @ prec).to(dev)
cmd = torch.from_numpy(cm).to(dev)
ms = gpu_ms(lambda: _hip.mahalanobis_score(f, cmd, packed, mu_p), reps=3)
s = _hip.mahalanobis_score(f, cmd, packed, mu_p)
m = 48
fs = f[:m].cpu().numpy()
cpu = cpu_rate(lambda: oracle.mahalanobis_score_reference_form(fs, cm, prec, C), m)
err = rel(s[:m].cpu().numpy(), oracle.mahalanobis_score(fs, cm, prec, C))
add("Mahalanobis C=10 (a6)", f"{N}x{D} f32", "rows", N, ms, "mfma_f64", 2.0 * D * D + 2 * D * C + 3 * D, cpu, f"reference double loop, {m} rows", err)
del f

# ---- a9 KDE -------------------------------------------------------------------------------------------
Mt, D, N = 10_000, 16, 8192
tr = torch.randn(Mt, D, dtype=torch.float64, device=dev, generator=g)
x = torch.randn(N, D, dtype=torch.float64, device=dev, generator=g)
ms = gpu_ms(lambda: _hip.kde_score(tr, x))
s = _hip.kde_score(tr, x)
m = 512
cpu = cpu_rate(lambda: oracle.kde_score(tr.cpu().numpy(), x[:m].cpu().numpy()), m)
err = rel(s[:m].cpu().numpy(), oracle.kde_score(tr.cpu().numpy(), x[:m].cpu().numpy()))
add("KDE / LaRED (a9), direct", f"{N} x train {Mt}x{D} f64", "rows", N, ms, "vector ALU: one f64 exp per (row, train row) pair; work unit = pair", 1.0 * Mt, cpu, f"numpy brute force, {m} rows", err)
# LaRED at PCA-256: pair distances on the f64 matrix cores (DetectorKDE takes this path for D >= 24)
D = 256
tr = torch.randn(Mt, D, dtype=torch.float64, device=dev, generator=g)
x = torch.randn(N, D, dtype=torch.float64, device=dev, generator=g) * 1.05
kst = _hip.kde_pack_train(tr)
bw = 8.0
ms = gpu_ms(lambda: _hip.kde_score_packed(kst, x, bw))
s = _hip.kde_score_packed(kst, x, bw)
m = 256
cpu = cpu_rate(lambda: oracle.kde_score(tr.cpu().numpy(), x[:m].cpu().numpy(), bw), m)
err = rel(s[:m].cpu().numpy(), oracle.kde_score(tr.cpu().numpy(), x[:m].cpu().numpy(), bw))
add("KDE / LaRED (a9), matrix cores", f"{N} x train {Mt}x{D} f64", "rows", N, ms, "mfma_f64", 2.0 * Mt * D, cpu, f"numpy brute force, {m} rows", err)

# ---- cfg2 stages ----------------------------------------------------------------------------------------
N, NMC, C, H, W, NP = 10_000, 16, 512, 4, 4, 256
import bench  # noqa: E402  (synthetic cfg2 latents without fully dropped maps)

x, rand = bench.synth_latents(N, 1235, 0.0, dev)
ms = gpu_ms(lambda: _hip.mc_stack(x, rand, NMC, 0.5, 2))
z = _hip.mc_stack(x, rand, NMC, 0.5, 2)
m = 256
cpu = cpu_rate(lambda: [oracle.mc_stack(x[i:i + 1].cpu().numpy(), rand[i].cpu().numpy(), 0.5, 2) for i in range(m)], m)
zo = np.concatenate([oracle.mc_stack(x[i:i + 1].cpu().numpy(), rand[i].cpu().numpy(), 0.5, 2) for i in range(8)])
add("mc_stack (a1)", f"{N}x{C}x{H}x{W} f32, 16 MC", "images", N, ms, "hbm", C * H * W * 4 + NMC * H * W * 4 + NMC * C * 4, cpu, f"numpy, {m} images", rel(z[:8 * NMC].cpu().numpy(), zo))
ms = gpu_ms(lambda: _hip.kl_entropy_per_dim(z, NMC, 5))
h = _hip.kl_entropy_per_dim(z, NMC, 5)
m = 96
zs = z[:m * NMC].cpu().numpy()
cpu = cpu_rate(lambda: oracle.get_dl_h_z(zs, NMC), m)
add("entropy per dim (a2)", f"{N} img x 16 x {C} f32", "images", N, ms, "hbm", NMC * C * 4 + C * 8, cpu, f"k-d tree per (image,dim) [reference form], {m} images", rel(h[:m].cpu().numpy(), oracle.kl_entropy_per_dim_vectorized(zs, NMC)))
ms = gpu_ms(lambda: _hip.kl_entropy_joint(z, NMC, 5))
hj = _hip.kl_entropy_joint(z, NMC, 5)
add("entropy joint (a2)", f"{N} img x 16 x {C} f32", "images", N, ms, "vector ALU f64: 120 pair maxima x 512 dims per image; work unit = (pair, dim)", 120.0 * C, float("nan"), "included in the row above", rel(hj[:m].cpu().numpy(), oracle.kl_entropy_joint_vectorized(zs, NMC)[:, 0]))
ms = gpu_ms(lambda: _hip.mc_entropy(x, rand, NMC, 0.5, 2, 5))
add("K1 mc_entropy (a1+a2)", f"{N}x{C}x{H}x{W} f32", "images", N, ms, "hbm", C * H * W * 4 + NMC * H * W * 4 + C * 8, float("nan"), "-", rel(_hip.mc_entropy(x, rand, NMC, 0.5, 2, 5)[:m].cpu().numpy(), h[:m].cpu().numpy()))
rng = np.random.default_rng(0)
comp = np.linalg.qr(rng.standard_normal((C, NP)))[0].T
mean, var = rng.standard_normal(C), rng.random(NP) + 0.05
a = rng.standard_normal((NP, NP))
prec = a @ a.T / NP + np.eye(NP)
mdm = rng.standard_normal((1, NP)) * 0.1
pct = _hip.pack_weights(torch.from_numpy(np.ascontiguousarray(comp.T)).to(dev))
pp = _hip.pack_weights(torch.from_numpy(prec).to(dev))
bias = torch.from_numpy((mean.reshape(1, -1) @ comp.T).ravel()).to(dev)
scale = torch.from_numpy(np.sqrt(var)).to(dev)
mdmd = torch.from_numpy(mdm.ravel()).to(dev)
hs = h[:2000].cpu().numpy()
ms = gpu_ms(lambda: _hip.pca_transform(h, pct, bias, scale, NP))
y = _hip.pca_transform(h, pct, bias, scale, NP)
cpu = cpu_rate(lambda: oracle.pca_transform(hs, comp, mean, var), 2000)
add("PCA transform (a4)", f"{N}x{C} -> {NP} f64", "rows", N, ms, "mfma_f64", 2.0 * C * NP, cpu, "numpy (BLAS), 2000 rows", rel(y[:2000].cpu().numpy(), oracle.pca_transform(hs, comp, mean, var)))
ms = gpu_ms(lambda: _hip.md_score(y, mdmd, pp))
s = _hip.md_score(y, mdmd, pp)
ys = y[:2000].cpu().numpy()
cpu = cpu_rate(lambda: oracle.md_score_reference_form(ys, mdm, prec), 2000)
add("LaREM / MD (a5)", f"{N}x{NP} f64", "rows", N, ms, "mfma_f64", 2.0 * NP * NP + 2 * NP, cpu, "reference N x N diag form, 2000 rows", rel(s[:2000].cpu().numpy(), oracle.md_score(ys, mdm, prec)))
ms = gpu_ms(lambda: _hip.pca_md_score(h, pct, bias, scale, mdmd, pp, NP))
add("K2 pca_md (a4+a5)", f"{N}x{C} f64", "rows", N, ms, "mfma_f64", 2.0 * C * NP + 2.0 * NP * NP + 2 * NP, float("nan"), "-", rel(_hip.pca_md_score(h, pct, bias, scale, mdmd, pp, NP)[:2000].cpu().numpy(), oracle.md_score(ys, mdm, prec)))

# K2': the same score from one folded contraction (LaREMPipeline default): M = W diag(1/scale) C, c = W(-bias/scale - mu)
evals, evecs = np.linalg.eigh(prec)
Wf = (evecs * np.sqrt(np.maximum(evals, 0.0))).T
Mf = Wf @ (comp / np.sqrt(var)[:, None])
cf = Wf @ (-(mean.reshape(1, -1) @ comp.T).ravel() / np.sqrt(var) - mdm.ravel())
pm = _hip.pack_weights(torch.from_numpy(np.ascontiguousarray(Mf.T)).to(dev))
cfd = torch.from_numpy(cf).to(dev)
ms = gpu_ms(lambda: _hip.proj_sq_score(h, pm, cfd, NP))
add("K2' proj_sq (a4+a5 folded)", f"{N}x{C} f64", "rows", N, ms, "mfma_f64", 2.0 * C * NP + 2 * NP, float("nan"), "-", rel(_hip.proj_sq_score(h, pm, cfd, NP)[:2000].cpu().numpy(), oracle.md_score(ys, mdm, prec)))

# ---- round 2: metrics step, setup-time fits, per-ROI path ---------------------------------------------------------------
ind = torch.randn(1_000_000, dtype=torch.float64, device=dev, generator=g) + 0.4
ood = torch.randn(1_000_000, dtype=torch.float64, device=dev, generator=g) - 0.4
ms = gpu_ms(lambda: _hip.ood_metrics(ind, ood), reps=5)
got = _hip.ood_metrics(ind, ood).cpu().numpy()
m = 200_000
t0 = time.perf_counter()
exp = oracle.auroc_fpr95_aupr(ind[:m].cpu().numpy(), ood[:m].cpu().numpy())
cpu = 2 * m / (time.perf_counter() - t0)
got_s = _hip.ood_metrics(ind[:m].contiguous(), ood[:m].contiguous()).cpu().numpy()
add("AUROC/FPR95/AUPR (f2)", "1M + 1M f64 scores", "scores", 2_000_000, ms, "hbm", 8 * (8 + 1) * 2 + 16, cpu,
    f"numpy argsort + cumsum restatement of torchmetrics, {2 * m} scores", float(np.max(np.abs(got_s - np.array(exp)))))

for n_e in (256, 512, 2048):
    a_ = torch.randn(n_e, n_e, dtype=torch.float64, device=dev, generator=g)
    a_ = a_ @ a_.T / n_e + torch.eye(n_e, dtype=torch.float64, device=dev)
    _hip.eigh(a_)  # first call: code-object load and LDS attribute, not the solver
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w_, v_ = _hip.eigh(a_)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    w_ref = np.linalg.eigvalsh(a_.cpu().numpy())
    cpu = 1.0 / (time.perf_counter() - t0)
    add("blocked Jacobi eigh (f1)", f"{n_e}x{n_e} f64 symmetric", "matrices", 1, ms, "blocked Jacobi: 63 barrier-separated inner steps per 64 x 64 sub-problem (LDS latency) + 64^3 matrix-core updates, (n/32 - 1) x 2 launches per sweep; work unit = rotation-element update", 10 * 2.0 * n_e**3, cpu,
        "numpy.linalg.eigvalsh (LAPACK, all host cores)", rel(w_.cpu().numpy(), w_ref))

xr = torch.randn(50_000, 512, dtype=torch.float64, device=dev, generator=g) * (0.2 + torch.rand(512, dtype=torch.float64, device=dev, generator=g))
from runia_core_amd.device_fit import pca_fit_device  # noqa: E402

t0 = time.perf_counter()
fit = pca_fit_device(xr.cpu().numpy(), 256)
ms = (time.perf_counter() - t0) * 1e3
from sklearn.decomposition import PCA as _PCA  # noqa: E402

t0 = time.perf_counter()
ref_fit = _PCA(n_components=256, svd_solver="covariance_eigh", whiten=True).fit(xr.cpu().numpy())
cpu = 1.0 / (time.perf_counter() - t0)
add("PCA fit covariance_eigh (f1)", "50000x512 f64 -> 256 (incl. H2D of the rows)", "fits", 1, ms, "eigen-solver launch latency + H2D; work unit = covariance FLOP", 2.0 * 50_000 * 512 * 512, cpu,
    "sklearn PCA(svd_solver='covariance_eigh')", float(np.abs(fit.components_ - ref_fit.components_).max()))

fm = torch.relu(torch.randn(1, 256, 50, 80, device=dev, generator=g))
kb = 1000
xy = torch.rand(kb, 2, device=dev, generator=g) * torch.tensor([400.0, 250.0], device=dev)
wh = 30 + torch.rand(kb, 2, device=dev, generator=g) * torch.tensor([200.0, 120.0], device=dev)
boxes = torch.cat([xy, xy + wh], dim=1)
ms = gpu_ms(lambda: _hip.roi_align(fm, boxes, 7, 80 / 640, 2, True))
rois = _hip.roi_align(fm, boxes, 7, 80 / 640, 2, True)
mb = 4
t0 = time.perf_counter()
exp = oracle.roi_align(fm.cpu().numpy(), boxes[:mb].cpu().numpy(), 7, 80 / 640, 2, True)
cpu = mb / (time.perf_counter() - t0)
add("roi_align (f3)", f"{kb} boxes x 256 ch x 7x7, sampling 2", "boxes", kb, ms, "hbm", 256 * 49 * 4 * (1 + 4 * 4), cpu,
    f"numpy restatement, {mb} boxes", rel(rois[:mb].cpu().numpy(), exp))
ms = gpu_ms(lambda: _hip.mc_entropy(rois, _hip.CounterDraws(3, 0), 16, 0.4, 3, 5))
add("per-ROI MC entropy 7x7 (f3)", f"{kb} ROIs x 256 ch x 7x7, 16 MC", "boxes", kb, ms, "hbm", 256 * 49 * 4 + 256 * 8, float("nan"), "-", 0.0)
# the same two launches (K0 + K1) at cfg4's size: 1 000 boxes are one round of workgroups and mostly launch latency
kb2 = 60_000
rois2 = torch.relu(torch.randn(kb2, 256, 7, 7, device=dev, generator=g))
for n_mc_r in (16, 32):
    ms = gpu_ms(lambda: _hip.mc_entropy(rois2, _hip.CounterDraws(3, 0), n_mc_r, 0.4, 3, 5), reps=3)
    ref_r = _hip.kl_entropy_per_dim(_hip.mc_stack(rois2[:64], _hip.CounterDraws(3, 0), n_mc_r, 0.4, 3), n_mc_r, 5)
    got_r = _hip.mc_entropy(rois2[:64].contiguous(), _hip.CounterDraws(3, 0), n_mc_r, 0.4, 3, 5)
    fin = torch.isfinite(ref_r)
    add(f"per-ROI MC entropy 7x7, {n_mc_r} MC (f3)", f"{kb2} ROIs x 256 ch x 7x7", "boxes", kb2, ms, "hbm", 256 * 49 * 4 + 256 * 8,
        float("nan"), "fused vs unfused kernels", float((got_r[fin] - ref_r[fin]).abs().max()))
del rois2

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"host_cores": os.cpu_count(), "device": torch.cuda.get_device_name(0), "rows": rows},
          open(os.path.join(ROOT, "gpurun_out", "stages.json"), "w"), indent=1)
with open(os.path.join(ROOT, "gpurun_out", "stages.md"), "w") as f:
    f.write("| stage | shape | GPU ms | GPU units/s | bound | achieved | frac of peak | CPU oracle units/s (1 core) | CPU form | max rel err |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r['stage']} | {r['shape']} | {r['gpu_ms']} | {r['gpu_rate']:.4g} {r['unit']}/s | {r['bound']} | {r['achieved']} {r['ach_unit']} | {'-' if r['frac'] is None else format(r['frac'], '.1%')} | {r['cpu_rate']:.4g} | {r['cpu_note']} | {r['max_rel_err']:.1e} |\n")
print("written gpurun_out/stages.{json,md}")
