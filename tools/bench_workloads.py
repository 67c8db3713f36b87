#!/usr/bin/env python3
"""Workloads of bench.py beyond the headline cfg2 step (imported by bench.py; nothing here runs on import).

* cfg3 (BASELINE.json configs[2], SURVEY 8d "cfg3-synth"): synthetic F f32[1 000 000, 2048], 10 classes, logits
  1 M x 1000 and 1 M x 10, kNN bank 50 000 x 2048, scored by ``Mahalanobis`` (reference inference/funcs.py:69-102),
  ``Energy`` (inference/postprocessors.py:549) and ``KNN(k=50)`` (:873-880) through ``postprocess_device``; the rows
  are cut with ``shard_bounds``, every rank builds ONLY its own block on its GPU, and each postprocessor ends in one
  all_gather of the score shards (``gather_scores``).  The synthetic rows are generated in fixed blocks of 15 625 rows
  keyed by the block index, so the 1 M rows are the same for every world size.
* cfg4 LaRED leg (configs[3]): 100 000 proposals x 16 MC x 1024-d -> per-dimension entropy -> PCA-256 -> KDE.

The oracle is imported only for the bounded parity / CPU-baseline slices, after the timed regions.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D_FEAT, N_CLASSES, BANK_ROWS, K_NN, N_LOGITS = 2048, 10, 50_000, 50, 1000
GEN_BLOCK = 15_625                       # 1 000 000 / 64: the shard boundaries of 1, 2, 4, 8, 16, 32, 64 ranks fall on blocks
HBM_PEAK_GBS, F32_MFMA_TF, F64_MFMA_TF = 8000.0, 157.3, 78.6  # MI355X_MICROARCH.md; f64 matrix rate: AMD datasheet
BF16_MFMA_TF = 2500.0  # dense bf16 (same guide)


def _gen(device, seed):
    return torch.Generator(device=device).manual_seed(seed)


def class_centres(device):
    return torch.randn(N_CLASSES, D_FEAT, device=device, generator=_gen(device, 2024)) * 0.5  # N(0, 0.25 I)


def linear_head(device):
    g = _gen(device, 2025)
    w = torch.randn(N_LOGITS, D_FEAT, device=device, generator=g) * (D_FEAT ** -0.5)
    b = torch.randn(N_LOGITS, device=device, generator=g) * 0.1
    return w.contiguous(), b.contiguous()


def feature_block(j: int, split: int, device, centres):
    """Block j (GEN_BLOCK rows) of split 0 (test) / 1 (train): F = ReLU(mu_label + N(0, I)), labels uniform."""
    g = _gen(device, 2024 + 7919 * (j + 1) + 1_000_003 * split)
    lab = torch.randint(0, N_CLASSES, (GEN_BLOCK,), device=device, generator=g)
    f = torch.relu(centres[lab] + torch.randn(GEN_BLOCK, D_FEAT, device=device, generator=g))
    return f, lab


def feature_rows(a: int, b: int, split: int, device, centres):
    """Rows [a, b) of the split, identical whatever the sharding."""
    f = torch.empty((b - a, D_FEAT), dtype=torch.float32, device=device)
    lab = torch.empty((b - a,), dtype=torch.int64, device=device)
    j = a // GEN_BLOCK
    while j * GEN_BLOCK < b:
        fb, lb = feature_block(j, split, device, centres)
        lo, hi = max(a, j * GEN_BLOCK), min(b, (j + 1) * GEN_BLOCK)
        f[lo - a: hi - a] = fb[lo - j * GEN_BLOCK: hi - j * GEN_BLOCK]
        lab[lo - a: hi - a] = lb[lo - j * GEN_BLOCK: hi - j * GEN_BLOCK]
        j += 1
    return f, lab


def logits_of(f, w, b):
    from runia_core_amd import _hip

    out = torch.empty((f.shape[0], w.shape[0]), dtype=torch.float32, device=f.device)
    for a in range(0, f.shape[0], 131072):  # the head of the synthetic model: setup, untimed
        out[a: a + 131072] = _hip.linear(f[a: a + 131072], w, b)
    return out


def warm_clocks(fn, seconds: float = 1.0):
    """Run ``fn`` untimed for ``seconds`` so that the leg's own kernels bring the clocks up after the host-side work before it
    (a fit or a CPU-baseline leg on the host leaves the GPU idle for seconds; the first passes after are slower by 5-15 %, the
    HBM-bound ones included).  Ends WITHOUT a synchronisation: the clock probe and the timed reps queue directly behind it."""
    t = time.perf_counter()
    n = 0
    while n < 2 or time.perf_counter() - t < seconds:
        torch.cuda.synchronize()
        fn()
        n += 1
    return n


def spread(values) -> dict:
    v = [float(x) for x in values]
    return {"min": round(min(v), 4), "median": round(float(np.median(v)), 4), "max": round(max(v), 4), "reps": len(v)}


def clock_bracket():
    """A clock probe now (queued on the current stream, directly behind whatever was launched last - call it without a
    synchronisation in between); call the result right after the region's last launch for {"before", "after"} in GHz."""
    from runia_core_amd import _hip

    before = _hip.clock_probe()

    def done():
        after = _hip.clock_probe()
        torch.cuda.synchronize()
        return {"before": _hip.clock_ghz(before)["ghz"], "after": _hip.clock_ghz(after)["ghz"]}

    return done


def stage_clocks(stages) -> dict:
    """One more (untimed) pass of a chain with a clock probe queued directly behind every stage: {stage: GHz}.  ``stages`` = list
    of (name, fn(previous result) -> result).  The same launch reads a different fraction of its roofline behind a bandwidth-bound
    kernel than behind an instruction-bound one (cfg4's PCA: 0.67 in one leg, 0.78 in the other, round 5) - the clock each stage
    STARTS at is the previous stage's reading here, the one it holds is its own."""
    from runia_core_amd import _hip

    probes, prev = [], None
    for name, fn in stages:
        prev = fn(prev)
        probes.append((name, _hip.clock_probe()))
    torch.cuda.synchronize()
    return {name: _hip.clock_ghz(pr)["ghz"] for name, pr in probes}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def fit_cfg3(device, fit_rows: int, centres, w, b, time_host_fit: bool = False):
    """setup() of the three postprocessors exactly as a user of the reference would call it (host arrays in)."""
    from runia_core_amd import config as rc_config
    from runia_core_amd.inference.postprocessors import KNN, Energy, Mahalanobis

    t0 = time.perf_counter()
    f_tr, lab_tr = feature_rows(0, BANK_ROWS, 1, device, centres)
    f_val, _ = feature_rows(BANK_ROWS, BANK_ROWS + 2 * GEN_BLOCK, 1, device, centres)
    f_val = f_val[:2048]
    tr_host, lab_host, val_host = f_tr.cpu().numpy(), lab_tr.cpu().numpy(), f_val.cpu().numpy()
    t_data = time.perf_counter() - t0

    def fit():
        t1 = time.perf_counter()
        maha = Mahalanobis(flip_sign=False, num_classes=N_CLASSES)
        maha.setup(tr_host[:fit_rows], train_labels=lab_host[:fit_rows], valid_feats=val_host)
        knn = KNN(flip_sign=False, k_neighbors=K_NN)
        knn.setup(tr_host, valid_feats=val_host)
        energy = Energy(flip_sign=False)
        energy.setup(logits_of(f_tr[:8192], w, b).cpu().numpy())
        torch.cuda.synchronize()
        return maha, knn, energy, time.perf_counter() - t1

    maha, knn, energy, t_fit = fit()  # as configured (default: the device when there is one)
    rec = {"mahalanobis": maha, "knn": knn, "energy": energy, "fit_s": t_data + t_fit,
           "fit_mode": "device" if rc_config.use_device_fit() else "host", "fit_only_s": t_fit}
    if time_host_fit and rc_config.use_device_fit():  # the reference's own host calls beside it (reported, not used)
        before = rc_config.device_fit
        rc_config.device_fit = False
        try:
            rec["fit_only_s_host_calls"] = fit()[3]
        finally:
            rc_config.device_fit = before
    return rec


def run_cfg3(device, rank: int, world: int, dist, rows_total: int, fit_rows: int, steps: int, warmup: int,
             cpu_legs: bool = True, log=None, f4: bool = False) -> dict:
    """One step = the rank's block of the 1 M rows through Mahalanobis, Energy (C = 1000 and C = 10) and kNN, each ending
    in its all_gather.  Returns the timing record (rank 0: plus parity and the CPU legs)."""
    from runia_core_amd.distributed import broadcast_fitted, gather_scores, shard_bounds

    say = log or (lambda *_: None)
    use_dist = dist is not None
    centres = class_centres(device)
    w, b = linear_head(device)
    fitted = fit_cfg3(device, fit_rows, centres, w, b, time_host_fit=(world == 1 and cpu_legs)) if rank == 0 else None
    t_b = time.perf_counter()
    if use_dist:
        fitted = broadcast_fitted(fitted, src=0)
    bcast_s = time.perf_counter() - t_b
    maha, knn, energy = fitted["mahalanobis"], fitted["knn"], fitted["energy"]
    say(f"cfg3: fitted in {fitted['fit_s']:.1f} s on rank 0, state broadcast in {bcast_s:.2f} s")

    a, bnd = shard_bounds(rows_total, world, rank)
    feats, _ = feature_rows(a, bnd, 0, device, centres)
    logits = logits_of(feats, w, b)
    logits10 = logits[:, :N_CLASSES].contiguous()
    n_loc = bnd - a
    per = -(-rows_total // world)
    bufs = {name: torch.empty(world * per, dtype=dt, device=device)
            for name, dt in (("mahalanobis", torch.float64), ("energy_c1000", torch.float32),
                             ("energy_c10", torch.float32), ("knn", torch.float32))} if use_dist else {}
    legs = (("mahalanobis", maha, feats), ("energy_c1000", energy, logits), ("energy_c10", energy, logits10),
            ("knn", knn, feats))
    events = {name: [] for name, _, _ in legs}
    gather_events = []

    def one_pass(timed: bool):
        out = {}
        for name, pp, x in legs:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            s = pp.postprocess_device(x)
            e1.record()
            if use_dist:
                s = gather_scores(s.reshape(-1), rows_total, out=bufs[name])
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record()
                if timed:
                    gather_events.append((e1, e2))
            if timed:
                events[name].append((e0, e1))
            out[name] = s
        return out

    for _ in range(max(1, warmup)):
        scores = one_pass(False)
    from runia_core_amd import _hip

    probe_before = _hip.clock_probe()  # directly behind the warm-up pass
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        scores = one_pass(True)
    probe_after = _hip.clock_probe()   # directly behind the last pass: ~30 us of one wave inside a region of >= 0.5 s
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    clocks = {"before": _hip.clock_ghz(probe_before)["ghz"], "after": _hip.clock_ghz(probe_after)["ghz"]}
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_rep = {name: [x.elapsed_time(y) for x, y in ev] for name, ev in events.items()}
    ms = {name: float(np.mean(v)) for name, v in per_rep.items()}
    gather_ms = float(np.mean([x.elapsed_time(y) for x, y in gather_events])) if gather_events else 0.0
    rec = {"elapsed": elapsed, "rows_total": rows_total, "rows_local": n_loc, "steps": steps,
           "ms_per_step": 1e3 * elapsed / steps, "value": rows_total * steps / elapsed, "stage_ms": ms,
           "gather_ms_per_call": gather_ms, "fit_s": fitted["fit_s"], "broadcast_s": bcast_s, "fit_rows": fit_rows,
           "fit_mode": fitted.get("fit_mode"), "fit_only_s": fitted.get("fit_only_s"),
           "fit_only_s_host_calls": fitted.get("fit_only_s_host_calls"), "clock_ghz_observed": clocks}
    if rank != 0:
        return rec

    work = {  # algorithmic work per row
        "mahalanobis": ("mfma_f64", 2.0 * D_FEAT * D_FEAT + 2.0 * D_FEAT * N_CLASSES + 3.0 * D_FEAT, F64_MFMA_TF, "TFLOP/s"),
        "energy_c1000": ("hbm", 4.0 * N_LOGITS + 4, HBM_PEAK_GBS, "GB/s"),
        "energy_c10": ("hbm", 4.0 * N_CLASSES + 4, HBM_PEAK_GBS, "GB/s"),
        "knn": ("mfma_f32", 2.0 * BANK_ROWS * D_FEAT, F32_MFMA_TF, "TFLOP/s"),
    }
    # large problems take their candidate distances from bf16 piece products: the executed work is that many contractions
    from runia_core_amd import _hip

    pieces = int(_hip.load_library().runia_knn_piece_products(n_loc, BANK_ROWS, D_FEAT))
    if pieces:
        work["knn"] = ("mfma_bf16", pieces * 2.0 * BANK_ROWS * D_FEAT, BF16_MFMA_TF, "TFLOP/s")
    stages = {}
    for name, (bound, per_row, peak, unit) in work.items():
        rate = n_loc / (ms[name] * 1e-3)
        ach = per_row * rate / (1e9 if bound == "hbm" else 1e12)
        stages[name] = {"rows": n_loc, "ms": round(ms[name], 4), "ms_spread": spread(per_rep[name]), "rows_per_s": round(rate, 1), "bound": bound,
                        "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4)}
    stages["knn"]["shape"] = f"{n_loc} queries x bank {BANK_ROWS}x{D_FEAT} f32, k={K_NN} (normaliser + distances + select)"
    if pieces:
        stages["knn"]["piece_products"] = pieces
        stages["knn"]["f32_equivalent_tflops"] = round(stages["knn"]["achieved"] / pieces, 2)
        stages["knn"]["note"] = ("candidate distances from bf16 piece products (csrc/knn_bf16.hip), `achieved` counts every executed "
                                 "product; the score is the exactly re-measured f32 distance")
    stages["mahalanobis"]["shape"] = f"{n_loc}x{D_FEAT} f32, {N_CLASSES} classes, f64 quadratic forms"
    # bytes past the L2 from the PMC passes of this command (profiles/pmc_traffic.json, "stages"), scaled to this rank's rows
    try:
        import json as _json

        pmc = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")))["stages"]
        for name, per in (("mahalanobis", "rows_per_launch"), ("knn", "queries_per_launch"), ("energy_c1000", "rows_per_launch")):
            e = pmc[name]
            stages[name]["traffic"] = int(e["bytes_per_launch"] * n_loc / e[per])
            stages[name]["traffic_source"] = f"profiles/pmc_traffic.json stages.{name} ({e['kernel']}): 2 x FETCH_SIZE + WRITE_SIZE per launch, scaled by rows; not measured in this run"
            for extra in ("matrix_pipe_busy_share", "clock_ghz_held"):
                if extra in e:
                    stages[name][extra] = e[extra]
    except Exception:
        pass
    stages["energy_c1000"]["shape"] = f"{n_loc}x{N_LOGITS} f32"
    stages["energy_c10"]["shape"] = f"{n_loc}x{N_CLASSES} f32"
    rec["stages"] = stages
    if f4:  # the remaining registry kernels on the same resident rows (their own warm-up and reps)
        rec["f4"] = run_f4_legs(device, feats, logits, w, b, centres, cpu_legs)
    if not cpu_legs:
        return rec

    import oracle  # checker / CPU baseline only (after the timed region)

    host = {name: scores[name][: min(n_loc, 4096)].cpu().numpy() for name in scores}
    cm, prec = np.asarray(maha.class_mean), np.asarray(maha.precision)
    m_m, m_k, m_e = min(n_loc, 96), min(n_loc, 16), min(n_loc, 4096)
    fs = feats[:m_m].cpu().numpy()
    t0 = time.perf_counter()
    ref_m = oracle.mahalanobis_score_reference_form(fs, cm, prec, N_CLASSES)
    t_m = (time.perf_counter() - t0) / max(1, m_m)
    stages["mahalanobis"]["max_rel_err"] = _rel(host["mahalanobis"][:m_m], oracle.mahalanobis_score(fs, cm, prec, N_CLASSES))
    stages["mahalanobis"]["max_rel_err_reference_form"] = _rel(host["mahalanobis"][:m_m], ref_m)
    stages["mahalanobis"]["cpu_rows_per_s"] = round(1.0 / t_m, 2)
    stages["mahalanobis"]["cpu_form"] = f"reference double loop over rows and classes, {m_m} rows, 1 core"
    bank_host = knn.index._host
    qs = feats[:m_k].cpu().numpy()
    t0 = time.perf_counter()
    ref_k = oracle.knn_kth_score(bank_host, qs, K_NN, chunk=1)
    t_k = (time.perf_counter() - t0) / max(1, m_k)
    stages["knn"]["max_rel_err"] = _rel(host["knn"][:m_k], ref_k)
    stages["knn"]["cpu_rows_per_s"] = round(1.0 / t_k, 2)
    stages["knn"]["cpu_form"] = f"one query at a time, exact f32 differences against the bank, {m_k} queries, 1 core"
    t_e = {}
    for name, x in (("energy_c1000", logits), ("energy_c10", logits10)):
        xs = x[:m_e].cpu().numpy()
        t0 = time.perf_counter()
        ref_e = oracle.energy_score(xs)
        t_e[name] = (time.perf_counter() - t0) / max(1, m_e)
        stages[name]["max_rel_err"] = _rel(host[name][:m_e], ref_e)
        stages[name]["cpu_rows_per_s"] = round(1.0 / t_e[name], 1)
        stages[name]["cpu_form"] = f"scipy logsumexp, {m_e} rows, 1 core"
    per_row = t_m + t_k + sum(t_e.values())
    rec["cpu_baseline"] = {
        "value": round(1.0 / per_row, 3), "unit": "rows/s", "cores": 1, "kind": "port",
        "sample": f"oracle in the reference's algorithmic form on slices of the same rows: Mahalanobis {m_m} rows "
                  f"({t_m * 1e3:.1f} ms/row), kNN {m_k} queries ({t_k * 1e3:.1f} ms/query), Energy {m_e} rows twice; "
                  f"value = 1 / (sum of the per-row times); host has {os.cpu_count()} cores",
    }
    rec["parity"] = {name: stages[name]["max_rel_err"] for name in stages}
    return rec


def run_cfg4_lared(device, n_props: int = 100_000, n_train: int = 4000, n_mc: int = 16, d: int = 1024, n_pca: int = 256,
                   reps: int = 10) -> dict:
    """BASELINE configs[3], LaRED leg: per-proposal MC samples (n_props*16, 1024) f32 resident in HBM -> per-dimension
    entropy -> PCA-256 (whitened) -> KDELatentSpace fitted on 4 000 in-distribution proposals."""
    import runia_core_amd as rc
    from runia_core_amd import _hip
    from runia_core_amd.dimensionality_reduction import device_pca_for
    from runia_core_amd.inference.postprocessors import KDELatentSpace

    g = _gen(device, 44)
    rel_spread = 0.10 * (0.5 + torch.rand(1, 1, d, device=device, generator=_gen(device, 7)))

    def proposals(n):
        out = torch.empty((n * n_mc, d), dtype=torch.float32, device=device)
        for a in range(0, n, 20_000):
            m = min(20_000, n - a)
            base = torch.randn(m, 1, d, device=device, generator=g) + 2
            out[a * n_mc: (a + m) * n_mc] = (base * (1 + rel_spread * torch.randn(m, n_mc, d, device=device, generator=g))).reshape(m * n_mc, d)
        return out

    h_tr = _hip.to_host(_hip.kl_entropy_per_dim(proposals(n_train), n_mc, 5))
    np.random.seed(4)
    red, pca = rc.apply_pca_ds_split(h_tr, n_pca)
    kde = KDELatentSpace()
    kde.setup(red)
    dp = device_pca_for(pca)
    z = proposals(n_props)

    def chain(ev=None):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        marks[0].record()
        h = _hip.kl_entropy_per_dim(z, n_mc, 5)
        marks[1].record()
        y = dp.transform_device(h)
        marks[2].record()
        s = kde.postprocess_device(y)
        marks[3].record()
        if ev is not None:
            ev.append(marks)
        return s

    warm_clocks(chain)
    ev = []
    clocks = clock_bracket()
    t0 = time.perf_counter()
    for _ in range(reps):
        s = chain(ev)
    clocks = clocks()  # the probe directly behind the last rep (~30 us of one wave), then the synchronisation
    wall = (time.perf_counter() - t0) / reps
    by_stage = stage_clocks([("entropy", lambda _: _hip.kl_entropy_per_dim(z, n_mc, 5)), ("pca", dp.transform_device),
                             ("kde", kde.postprocess_device)])
    per = [[m[i].elapsed_time(m[i + 1]) for m in ev] for i in range(3)]
    ms = [float(np.median(p)) for p in per]   # median of the reps (min / max beside it)
    ent_gbs = (n_mc * d * 4 + d * 8) * n_props / (ms[0] * 1e-3) / 1e9
    pca_tf = 2.0 * d * n_pca * n_props / (ms[1] * 1e-3) / 1e12
    kde_tf = 2.0 * n_train * n_pca * n_props / (ms[2] * 1e-3) / 1e12
    rec = {"shape": f"{n_props} proposals x {n_mc} MC x {d} f32 -> entropy -> PCA-{n_pca} -> KDE on {n_train} train rows",
           "rows": n_props, "ms": round(1e3 * wall, 4), "rows_per_s": round(n_props / wall, 1), "clock_ghz_observed": clocks,
           "clock_ghz_behind_stage": by_stage,
           "timing": "0.3 s of the same chain untimed, then the median of %d reps per stage (HIP events)" % reps,
           "entropy": {"ms": round(ms[0], 4), "ms_spread": spread(per[0]), "bound": "hbm", "achieved": round(ent_gbs, 1), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(ent_gbs / HBM_PEAK_GBS, 4)},
           "pca": {"ms": round(ms[1], 4), "ms_spread": spread(per[1]), "bound": "mfma_f64", "achieved": round(pca_tf, 2), "peak": F64_MFMA_TF,
                   "unit": "TFLOP/s", "frac": round(pca_tf / F64_MFMA_TF, 4)},
           "kde": {"ms": round(ms[2], 4), "ms_spread": spread(per[2]), "bound": "mfma_f64", "achieved": round(kde_tf, 2), "peak": F64_MFMA_TF,
                   "unit": "TFLOP/s", "frac": round(kde_tf / F64_MFMA_TF, 4)}}
    import oracle  # checker / CPU baseline only

    m = 24
    zs = z[: m * n_mc].cpu().numpy()
    t0 = time.perf_counter()
    _, h_o = oracle.get_dl_h_z(zs, n_mc)  # reference form: one k-d tree per (proposal, dim)
    y_o = oracle.pca_transform(h_o, pca.components_, pca.mean_, pca.explained_variance_)
    s_o = oracle.kde_score(red, y_o)
    t_cpu = (time.perf_counter() - t0) / m
    rec["max_rel_err"] = _rel(s[:m].cpu().numpy(), s_o)
    rec["cpu_rows_per_s"] = round(1.0 / t_cpu, 2)
    rec["cpu_form"] = f"k-d tree per (proposal, dim) + numpy PCA + brute-force KDE, {m} proposals, 1 core"
    del z
    return rec


def _roi_valu(k: int, c: int, ms: float) -> dict:
    """Vector-instruction account of the fused ROI launch: instructions per (ROI, 64 channels) and the clock held from the PMC passes
    of the same command (profiles/pmc_traffic.json), cycles per instruction per SIMD from this run's time."""
    import json as _json
    try:
        e = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")))["stages"][
            "cfg4_roi_sampler_entropy"]
    except (OSError, KeyError, ValueError):
        return {}
    insts, ghz = float(e["valu_insts_per_wave"]), float(e["clock_ghz_held"])
    cycles = ms * 1e-3 * ghz * 1e9 * 1024 / (k * (c / 64))
    return {"insts_per_roi_64ch": insts, "clock_ghz_held": ghz, "cycles_per_inst_per_simd": round(cycles / insts, 2),
            "floor_ms_at_2p5_cycles": round(k * (c / 64) * insts * 2.5 / 1024 / (ghz * 1e9) * 1e3, 2),
            "note": "SQ_INSTS_VALU per wave and GRBM_GUI_ACTIVE / 8 / duration from profiles/r4b_cfg4_pmc_summary.json (not measured in this run); "
                    "2.5 cycles = the issue cost of a plain f32 instruction (tools/microbench/valu_issue.hip), most of this mix"}


def run_cfg4_from_maps(device, n_img: int = 100, per_img: int = 1000, c: int = 1024, fh: int = 45, fw: int = 80, n_mc: int = 16,
                       n_pca: int = 256, n_train: int = 4000, reps: int = 8) -> dict:
    """BASELINE configs[3] from where the reference's object-level path starts (feature_extraction/object_level.py:312-367):
    hooked feature maps (n_img, 1024, 45, 80) f32 + per_img proposal boxes per image -> roi_align 7x7 (sampling 2) ->
    per-proposal MC DropBlock x 16 -> per-dimension entropy -> PCA-256 -> KDELatentSpace.  The ROI part is ONE launch per
    slice of 65 535 proposals (runia_roi_mc_entropy_f32: roi_align folded into the sampler's load) behind a channels-last copy of
    the maps; the (K, C, 7, 7) tensor of roi_align is never written."""
    import runia_core_amd as rc
    from runia_core_amd import _hip
    from runia_core_amd.dimensionality_reduction import device_pca_for
    from runia_core_amd.inference.postprocessors import KDELatentSpace

    g = _gen(device, 45)
    fm = torch.relu(torch.randn(n_img, c, fh, fw, device=device, generator=g) + 0.3)
    k = n_img * per_img
    cg = torch.Generator().manual_seed(46)
    img_w, img_h = fw * 16.0, fh * 16.0
    wh = 24.0 + torch.rand(k, 2, generator=cg) ** 2 * torch.tensor([img_w * 0.5, img_h * 0.6])   # many small, few large boxes
    xy = torch.rand(k, 2, generator=cg) * (torch.tensor([img_w, img_h]) - wh).clamp_min(1.0)
    boxes = torch.cat([xy, xy + wh], dim=1).to(device)
    bidx = torch.arange(k, device=device, dtype=torch.int32) // per_img
    rand = torch.rand(k, n_mc, 7, 7, device=device, generator=g)
    drop, bs = 0.3, 3

    def rois_entropy(first, count, ev=None):
        t = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t[0].record()
        nhwc = _hip.nchw_to_nhwc(fm)
        t[1].record()
        h = _hip.roi_mc_entropy(nhwc, boxes[first: first + count], 7, fw / img_w, 2, True, rand[first: first + count], n_mc, drop, bs, 5,
                                batch_idx=bidx[first: first + count])
        t[2].record()
        if ev is not None:
            ev.append(t)
        return h

    h_tr = _hip.to_host(rois_entropy(0, n_train))
    h_tr = np.nan_to_num(h_tr, nan=0.0, posinf=0.0, neginf=0.0)  # (a fully dropped map gives NaN upstream as well; the fit wants finite rows)
    np.random.seed(4)
    red, pca = rc.apply_pca_ds_split(h_tr, n_pca)
    kde = KDELatentSpace()
    kde.setup(red)
    dp = device_pca_for(pca)

    def chain(ev_roi=None, ev=None):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        marks[0].record()
        h = rois_entropy(0, k, ev_roi)
        marks[1].record()
        y = dp.transform_device(h)
        marks[2].record()
        s = kde.postprocess_device(y)
        marks[3].record()
        if ev is not None:
            ev.append(marks)
        return s, h

    warm_clocks(chain)
    ev, ev_roi = [], []
    clocks = clock_bracket()
    t0 = time.perf_counter()
    for _ in range(reps):
        s, h = chain(ev_roi, ev)
    clocks = clocks()  # the probe directly behind the last rep (~30 us of one wave), then the synchronisation
    wall = (time.perf_counter() - t0) / reps
    by_stage = stage_clocks([("roi_sampler_entropy", lambda _: rois_entropy(0, k)), ("pca", dp.transform_device),
                             ("kde", kde.postprocess_device)])
    per = [[m[i].elapsed_time(m[i + 1]) for m in ev] for i in range(3)]
    ms = [float(np.median(p)) for p in per]
    # the ROI leg takes one launch group per slice of 65 535 proposals: sum the slices of a rep, then the median over reps
    per_rep = len(ev_roi) // reps
    per_t = [sum(t[0].elapsed_time(t[1]) for t in ev_roi[r * per_rep:(r + 1) * per_rep]) for r in range(reps)]
    per_r = [sum(t[1].elapsed_time(t[2]) for t in ev_roi[r * per_rep:(r + 1) * per_rep]) for r in range(reps)]
    ms_t, ms_r = float(np.median(per_t)), float(np.median(per_r))
    taps_gb = k * c * 49 * 4 * 4 * 4 / 1e9                       # 196 bilinear samples x 4 taps x 4 bytes per (proposal, channel)
    out_gb = k * c * 49 * 4 / 1e9                                 # what roi_align would have written (and K1 read)
    rec = {"shape": f"{n_img} maps x {c} ch x {fh}x{fw} f32 + {per_img} boxes each -> roi_align 7x7/2 -> {n_mc} MC DropBlock -> entropy "
                    f"-> PCA-{n_pca} -> KDE on {n_train} train rows",
           "rows": k, "ms": round(1e3 * wall, 3), "rows_per_s": round(k / wall, 1), "clock_ghz_observed": clocks,
           "clock_ghz_behind_stage": by_stage,
           "timing": "0.3 s of the same chain untimed, then the median of %d reps per stage (HIP events)" % reps,
           "channels_last_copy": {"ms": round(ms_t, 4), "ms_spread": spread(per_t), "bound": "hbm", "achieved": round(2 * fm.numel() * 4 / (ms_t * 1e-3) / 1e9, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s"},
           "roi_sampler_entropy": {"ms": round(ms_r, 4), "ms_spread": spread(per_r), "bound": "valu-issue", "bilinear_taps_TBps": round(taps_gb / ms_r, 2),
                                   "valu": _roi_valu(k, c, ms_r),
                                   "roi_tensor_not_written_GB": round(out_gb, 2),
                                   "unfused_roi_tensor_GBps_equiv": round(2 * out_gb / (ms_r * 1e-3), 1),
                                   "note": "one launch per 65 535 proposals: per-ROI sample table + keep-flag table + fused load/sampler/entropy"},
           "pca": {"ms": round(ms[1], 4), "ms_spread": spread(per[1]), "bound": "mfma_f64", "achieved": round(2.0 * c * n_pca * k / (ms[1] * 1e-3) / 1e12, 2),
                   "peak": F64_MFMA_TF, "unit": "TFLOP/s"},
           "kde": {"ms": round(ms[2], 4), "ms_spread": spread(per[2]), "bound": "mfma_f64", "achieved": round(2.0 * n_train * n_pca * k / (ms[2] * 1e-3) / 1e12, 2),
                   "peak": F64_MFMA_TF, "unit": "TFLOP/s"}}
    for key in ("pca", "kde"):
        rec[key]["frac"] = round(rec[key]["achieved"] / rec[key]["peak"], 4)
    import oracle  # checker / CPU baseline only

    m = 6
    t0 = time.perf_counter()
    fm_h = fm[: 1].cpu().numpy()
    rois_o = oracle.roi_align(fm_h, boxes[:m].cpu().numpy(), 7, fw / img_w, 2, True)
    z_o = np.concatenate([oracle.mc_stack(rois_o[i: i + 1], rand[i].cpu().numpy(), drop, bs) for i in range(m)])
    _, h_o = oracle.get_dl_h_z(z_o, n_mc)
    y_o = oracle.pca_transform(np.nan_to_num(h_o, nan=0.0), pca.components_, pca.mean_, pca.explained_variance_)
    s_o = oracle.kde_score(red, y_o)
    t_cpu = (time.perf_counter() - t0) / m
    ok = np.isfinite(h_o).all(axis=1)
    rec["max_rel_err_entropy"] = _rel(h[:m].cpu().numpy()[ok], h_o[ok])
    rec["max_rel_err"] = _rel(s[:m].cpu().numpy()[ok], s_o[ok])
    rec["cpu_rows_per_s"] = round(1.0 / t_cpu, 3)
    rec["cpu_form"] = f"numpy roi_align + DropBlock + k-d tree per (proposal, dim) + PCA + KDE, {m} proposals, 1 core"
    del fm, rand
    return rec


# ---------------------------------------------------------------------------------------------------------------------
# f-rows under the driver's clock (VERDICT r4 "next" #2): fits, metrics, the remaining registry kernels, joint entropy
# ---------------------------------------------------------------------------------------------------------------------
def _timed(fn, reps: int = 3, warm: float = 0.3):
    """(median ms, spread, last result) of ``fn`` by HIP events after ``warm`` seconds of the same call.  At least three reps: a leg
    that allocates its output per call (ASH-S: 8 GB) can catch one rep behind the caching allocator returning memory to the driver
    (57.8 ms beside 3.7 in one run of round 6) - the median of two is then the mean of both."""
    reps = max(3, int(reps))
    warm_clocks(fn, warm)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    marks[0].record()
    out = None
    for i in range(reps):
        out = fn()
        marks[i + 1].record()
    from runia_core_amd import _hip

    probe = _hip.clock_probe()  # directly behind the last rep: the clock THIS leg's kernels left the GPU at
    torch.cuda.synchronize()
    per = [marks[i].elapsed_time(marks[i + 1]) for i in range(reps)]
    sp = spread(per)
    sp["clock_ghz"] = _hip.clock_ghz(probe)["ghz"]
    return float(np.median(per)), sp, out


def _leg(ms, sp, rows, bound, per_row, peak, unit, **extra):
    rate = rows / (ms * 1e-3)
    ach = per_row * rate / (1e9 if bound == "hbm" else 1e12)
    rec = {"rows": rows, "ms": round(ms, 4), "ms_spread": sp, "rows_per_s": round(rate, 1), "bound": bound, "achieved": round(ach, 2),
           "peak": peak, "unit": unit, "frac": round(ach / peak, 4)}
    rec.update(extra)
    return rec


def run_fit_legs(device, centres, cpu_legs: bool = True) -> dict:
    """f1: the fits behind setup() at cfg3 / cfg2 sizes as kernels - covariance of 50 000 x 2048 f32 rows (gram_kernel on the f64
    matrix cores), pinvh of the 2048 x 2048 covariance (blocked Jacobi: sweeps, rotations, ms) - beside np.cov + scipy pinvh."""
    from runia_core_amd import _hip
    from runia_core_amd.device_fit import pinvh_device

    f_tr, _ = feature_rows(0, BANK_ROWS, 1, device, centres)
    n, d = f_tr.shape
    ms_c, sp_c, (mean, cov) = _timed(lambda: _hip.covariance(f_tr), reps=3)
    nt = (d + 127) // 128
    executed = 2.0 * (nt * (nt + 1) // 2) * 128 * 128  # per row: the 128 x 128 tile pairs of the upper triangle
    rec = {"covariance": _leg(ms_c, sp_c, n, "mfma_f64", executed, F64_MFMA_TF, "TFLOP/s",
                              shape=f"{n} x {d} f32 rows -> mean + {d} x {d} f64 covariance (np.cov(X.T, bias=1))",
                              kernel="col_sum + gram_kernel (128 x 128 tile pairs of the upper triangle, split over rows) + "
                                     "gram_finish (sums the slices, mirrors); `achieved` counts the executed products - "
                                     "the full square would be 2 D^2 per row",
                              full_square_equivalent_tflops=round(2.0 * d * d * n / (ms_c * 1e-3) * 1e-12, 2))}
    info = {}
    pinvh_device(cov)  # (first use of the factorisation kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prec = pinvh_device(cov)
    torch.cuda.synchronize()
    t_p = time.perf_counter() - t0
    t0 = time.perf_counter()
    _hip.eigh(cov, info=info)
    torch.cuda.synchronize()
    t_e = time.perf_counter() - t0
    rec["pinvh"] = {"n": d, "ms": round(1e3 * t_p, 2),
                    "what": "scipy.linalg.pinvh of a covariance that loses no direction to its cut-off (||A|| ||A^-1|| < 1e8): Cholesky factor + "
                            "triangular inverse + W^T W (round 6); anything else: the eigen route below",
                    "eigen_route_ms": round(1e3 * t_e, 2), "sweeps": info.get("sweeps"), "rotations": info.get("rotations"),
                    "eigen_route": "blocked Jacobi eigen-decomposition (runia_eigh_block_*) [+ cut-off + (U / s) U^T]",
                    "ms_per_sweep": round(1e3 * t_e / max(1, info.get("sweeps", 1)), 2)}
    if cpu_legs:
        import oracle  # checker / CPU baseline only
        from scipy.linalg import pinvh

        from runia_core_amd.host_threads import host_compute, usable_cpus

        x_h = f_tr.cpu().numpy()
        with host_compute():  # BLAS pools at the container's CPU quota (oversubscribed they are throttled: a slower baseline)
            t0 = time.perf_counter()
            cov_h = np.cov(x_h.astype(np.float64).T, bias=1)
            t_cov = time.perf_counter() - t0
            t0 = time.perf_counter()
            prec_h = pinvh(cov_h)
            t_pin = time.perf_counter() - t0
        rec["covariance"]["max_rel_err"] = _rel(cov.cpu().numpy(), cov_h)
        rec["covariance"]["cpu_ms"] = round(1e3 * t_cov, 1)
        rec["pinvh"]["max_rel_err"] = float(np.max(np.abs(prec.cpu().numpy() - prec_h)) / np.max(np.abs(prec_h)))
        rec["pinvh"]["cpu_ms"] = round(1e3 * t_pin, 1)
        rec["cpu_form"] = f"np.cov(X.T, bias=1) in f64 + scipy.linalg.pinvh, the reference's calls (EmpiricalCovariance), BLAS on {usable_cpus()} CPUs (the container's quota; {os.cpu_count()} visible)"
    del f_tr
    return rec


def run_metrics_leg(device, n_each: int = 1_000_000, cpu_legs: bool = True) -> dict:
    """f2: AUROC / FPR@95 / AUPR of 1 M InD + 1 M OoD scores on the device (runia_ood_metrics_*: split into buckets + a sort per
    bucket + scans) on score sets shaped like the postprocessors' outputs - LaREM (-chi2(256), f64: every score far outside
    [0, 1], its sigmoid crowded against 0), energies (f32, ~ N(9, 2)) - and on N(0, 1)-like scores; 20 000 scores as well."""
    from runia_core_amd import _hip

    g = _gen(device, 91)

    def chi2(n, scale):
        out = torch.empty(n, dtype=torch.float64, device=device)
        for a in range(0, n, 131072):
            m = min(131072, n - a)
            out[a: a + m] = -(torch.randn(m, 256, device=device, generator=g, dtype=torch.float64) ** 2).sum(1) * scale
        return out

    sets = {"larem_f64": (chi2(n_each, 1.0), chi2(n_each, 1.15)),
            "energy_f32": ((torch.randn(n_each, device=device, generator=g) * 2 + 9).float(), (torch.randn(n_each, device=device, generator=g) * 2 + 7).float()),
            "normal_f64": (torch.randn(n_each, device=device, generator=g, dtype=torch.float64) + 0.6,
                           torch.randn(n_each, device=device, generator=g, dtype=torch.float64) * 1.3 - 0.4)}
    recs = {}
    for name, (ind, ood) in sets.items():
        for tag, a, b in ((f"{name}_2m", ind, ood), (f"{name}_20k", ind[:10000].contiguous(), ood[:10000].contiguous())):
            ms, sp, out = _timed(lambda a=a, b=b: _hip.ood_metrics(a, b), reps=20, warm=0.2)
            n = a.numel() + b.numel()
            by = a.element_size() + 10  # read the score; key + label written and read once by a one-pass sort: a lower bound
            recs[tag] = {"scores": n, "ms": round(ms, 4), "ms_spread": sp, "keys_per_s": round(n / (ms * 1e-3), 1),
                         "bound": "hbm", "achieved": round(n * by / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(n * by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_score": by,
                         "values": [float(v) for v in out.cpu().tolist()]}
            if cpu_legs:
                import oracle  # checker / CPU baseline only

                t0 = time.perf_counter()
                exp = oracle.auroc_fpr95_aupr(a.cpu().numpy(), b.cpu().numpy())
                t_cpu = time.perf_counter() - t0
                recs[tag]["max_abs_err"] = float(max(abs(x - y) for x, y in zip(out.cpu().tolist(), exp)))
                recs[tag]["cpu_ms"] = round(1e3 * t_cpu, 2)
    recs["cpu_form"] = "NumPy argsort + cumulative sums + float32 trapezoids (torchmetrics' definitions), 1 core"
    recs["note"] = ("round 6: six launches (four without the sketch) - buckets by splitter KEYS (order by construction), bucket sort + curve terms in "
                    "one launch; round 5: 0.234 / 0.082 ms per 2 M / 20 000 scores in eight / seven launches; DESIGN.md 4.18, "
                    "profiles/r6_metrics_2m_kernel_stats.csv, r6_metrics_20k_kernel_stats.csv")
    return recs


def run_entropy_joint_leg(device, n_img: int = 10000, n_mc: int = 16, d: int = 512, cpu_legs: bool = True) -> dict:
    """a2's other output: the joint (Chebyshev) entropy of get_dl_h_z and the per-dimension entropies of the same samples."""
    from runia_core_amd import _hip

    g = _gen(device, 92)
    z = (torch.randn(n_img, 1, d, device=device, generator=g) + 0.1 * torch.randn(n_img, n_mc, d, device=device, generator=g)).reshape(n_img * n_mc, d).contiguous()
    by = n_mc * d * 4
    ms_j, sp_j, hj = _timed(lambda: _hip.kl_entropy_joint(z, n_mc, 5), reps=10)
    ms_d, sp_d, hd = _timed(lambda: _hip.kl_entropy_per_dim(z, n_mc, 5), reps=10)
    rec = {"shape": f"{n_img} images x {n_mc} MC x {d} f32",
           "joint": _leg(ms_j, sp_j, n_img, "hbm", by + 8, HBM_PEAK_GBS, "GB/s"),
           "per_dim": _leg(ms_d, sp_d, n_img, "hbm", by + d * 8, HBM_PEAK_GBS, "GB/s")}
    if hasattr(_hip, "kl_entropy_both"):
        ms_b, sp_b, (hj2, hd2) = _timed(lambda: _hip.kl_entropy_both(z, n_mc, 5), reps=10)
        rec["both_one_read"] = _leg(ms_b, sp_b, n_img, "hbm", by + d * 8 + 8, HBM_PEAK_GBS, "GB/s",
                                    same_bits_as_the_two_kernels=bool(torch.equal(hj2, hj) and torch.equal(hd2, hd)))
    if cpu_legs:
        import oracle  # checker / CPU baseline only

        m = 48
        t0 = time.perf_counter()
        ej, ed = oracle.get_dl_h_z(z[: m * n_mc].cpu().numpy(), n_mc)
        t_cpu = (time.perf_counter() - t0) / m
        rec["max_rel_err_joint"] = _rel(hj[:m].cpu().numpy(), np.ravel(ej))
        rec["max_rel_err_per_dim"] = _rel(hd[:m].cpu().numpy(), ed)
        rec["cpu_rows_per_s"] = round(1.0 / t_cpu, 2)
        rec["cpu_form"] = f"get_dl_h_z: one k-d tree per (image, dim) + one joint tree per image, {m} images, 1 core"
    return rec


def run_f4_legs(device, feats, logits, w, b, centres, cpu_legs: bool = True) -> dict:
    """f4: the remaining registry kernels on the resident cfg3 rows - ViM residual norm, ReAct / DICE / ASH linear heads,
    GEN, class-wise Gaussians (GMM / DDU) and pred_h / mi - each: rows/s, fraction of its bound, parity on a slice, CPU form."""
    from runia_core_amd import _hip
    from runia_core_amd.inference.funcs import GmmState, gmm_fit

    n, d = feats.shape
    c = logits.shape[1]
    rec = {}
    rng = np.random.default_rng(7)
    # ViM: || (x - u) NS ||, NS = 2048 x 1048 (DIM = 1000 of 2048 kept out), + energy of the logits
    dim_ns = d - (1000 if d >= 2048 else (512 if d >= 768 else d // 2))
    ns_h = np.linalg.qr(rng.standard_normal((d, dim_ns)))[0]
    u_h = (rng.standard_normal(d) * 0.1).astype(np.float32)
    packed_ns = _hip.pack_weights(_hip.to_device(np.ascontiguousarray(ns_h), torch.float64))
    u_d = _hip.to_device(u_h, torch.float32)
    alpha = 1.3

    def vim():
        lse, _ = _hip.row_lse_msp(logits, True, False)
        return lse.to(torch.float64) - alpha * _hip.proj_norm(feats, u_d, packed_ns, dim_ns)

    ms, sp, s_vim = _timed(vim, reps=3)
    rec["vim"] = _leg(ms, sp, n, "mfma_f64", 2.0 * d * dim_ns, F64_MFMA_TF, "TFLOP/s", shape=f"{n} x {d} f32, NS {d} x {dim_ns} f64 (+ logsumexp of {c} logits)")
    # ReAct (clipped) / DICE (masked weight): f32 linear head + logsumexp
    thr = float(np.float32(1.0))
    mask_w = (w * (torch.rand(w.shape, device=device, generator=_gen(device, 5)) > 0.9)).contiguous()

    def head(weight, clip):
        return _hip.row_lse_msp(_hip.linear(feats, weight, b, clip), True, False)[0]

    ms, sp, s_react = _timed(lambda: head(w, thr), reps=3)
    rec["react"] = _leg(ms, sp, n, "mfma_f32", 2.0 * d * c, F32_MFMA_TF, "TFLOP/s", shape=f"min(x, t) @ W^T + b, W {c} x {d} f32, then logsumexp")
    ms, sp, s_dice = _timed(lambda: head(mask_w, float("inf")), reps=3)
    rec["dice"] = _leg(ms, sp, n, "mfma_f32", 2.0 * d * c, F32_MFMA_TF, "TFLOP/s", shape="x @ (W * mask)^T + b (10 % of the weights kept), then logsumexp")
    ms, sp, s_ash = _timed(lambda: _hip.ash_s(feats, 85), reps=3)
    rec["ash_s"] = _leg(ms, sp, n, "hbm", 8.0 * d, HBM_PEAK_GBS, "GB/s", shape=f"{n} x {d} f32: top 15 % of every row kept, rescaled (the head + logsumexp follow as in react)")
    ms, sp, s_gen = _timed(lambda: _hip.gen_score(logits, 0.1, 100), reps=3)
    rec["gen"] = _leg(ms, sp, n, "hbm", 4.0 * c + 4, HBM_PEAK_GBS, "GB/s", shape=f"{n} x {c} f32 logits, M = 100, gamma = 0.1")
    n_mc = 16
    n_img = n // n_mc
    ms, sp, (ph, mi, _) = _timed(lambda: _hip.mcd_uncertainty(logits[: n_img * n_mc], n_mc), reps=3)
    rec["pred_h_mi"] = _leg(ms, sp, n_img, "hbm", 4.0 * c * n_mc + 8, HBM_PEAK_GBS, "GB/s", shape=f"{n_img} images x {n_mc} MC x {c} f32 logits")
    # class-wise Gaussians (GMM / DDU): fit on 31 250 train rows (host torch, as upstream), score = logsumexp_c log N(x; mu_c, S_c)
    f_tr, lab_tr = feature_rows(0, 2 * GEN_BLOCK, 1, device, centres)  # ~3 100 rows per class > D: full-rank class covariances, no jitter ladder
    t0 = time.perf_counter()
    gmm, jitter = gmm_fit(f_tr.cpu(), lab_tr.cpu(), N_CLASSES)
    state = GmmState(gmm)
    t_fit = time.perf_counter() - t0
    n_g = min(n, 262_144)
    xg = feats[:n_g]
    ms, sp, s_gmm = _timed(lambda: state.energy_device(xg), reps=3)
    # round 6: || L_c^-1 (x - mu_c) ||^2 with the triangular inverse factor on the f32 matrix cores (torch's own arithmetic), all
    # components in one launch: D^2 flop per (row, component) - the zero half of L_c^-1 is not multiplied (rounds 4-5: dense f64
    # x P_c x^T per component, 2 D^2 multiply-adds, 353 ms)
    rec["gmm_ddu"] = _leg(ms, sp, n_g, "mfma_f32", N_CLASSES * (1.0 * d * d), F32_MFMA_TF, "TFLOP/s",
                          shape=f"{n_g} x {d} f32, {N_CLASSES} full-covariance components, triangular L^-1 in f32 (D^2 flop per row and component)",
                          fit_s=round(t_fit, 2), jitter=float(jitter))
    # bytes past the L2 from the PMC passes of the driver's command (profiles/pmc_traffic.json), scaled by rows
    try:
        import json as _json

        pmc = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")))["stages"]
        for leg, key in (("vim", "f4_vim"), ("react", "f4_react"), ("dice", "f4_react"), ("ash_s", "f4_ash_s"), ("gen", "f4_gen"), ("pred_h_mi", "f4_pred_h_mi"),
                         ("gmm_ddu", "f4_gmm_whiten")):
            if key not in pmc:
                continue
            e = pmc[key]
            rec[leg]["traffic"] = int(e["bytes_per_launch"] * rec[leg]["rows"] / e["rows_per_launch"])
            rec[leg]["traffic_source"] = f"profiles/pmc_traffic.json stages.{key} ({e['kernel']}): 2 x FETCH_SIZE + WRITE_SIZE per launch, scaled by rows; not measured in this run"
            for extra in ("matrix_pipe_busy_share", "clock_ghz_held"):
                if extra in e:
                    rec[leg][extra] = e[extra]
    except Exception:
        pass
    if not cpu_legs:
        return rec
    import oracle  # checker / CPU baseline only

    def cpu(fn, m):
        t0 = time.perf_counter()
        out = fn()
        return out, (time.perf_counter() - t0) / m

    m = 256
    fs, ls = feats[:m].cpu().numpy(), logits[:m].cpu().numpy()
    wh, bh = w.cpu().numpy(), b.cpu().numpy()
    exp, t = cpu(lambda: oracle.vim_score(fs, ls, u_h, ns_h, alpha), m)
    rec["vim"].update(max_rel_err=_rel(s_vim[:m].cpu().numpy(), exp), cpu_rows_per_s=round(1 / t, 1))
    exp, t = cpu(lambda: oracle.react_score(fs, wh, bh, thr), m)
    rec["react"].update(max_rel_err=_rel(s_react[:m].cpu().numpy(), exp), cpu_rows_per_s=round(1 / t, 1))
    mwh = mask_w.cpu().numpy()
    exp, t = cpu(lambda: logsumexp_rows(oracle.dice_logits(fs[:32], mwh, bh)), 32)
    rec["dice"].update(max_rel_err=_rel(s_dice[:32].cpu().numpy(), exp), cpu_rows_per_s=round(1 / t, 1),
                       cpu_form="RouteDICE.forward's (N, C, D) broadcast product, 32 rows, 1 core")
    _, t = cpu(lambda: oracle.ash_s_linear_layer(fs, 85), m)     # the reference's call (timed) ...
    exp = oracle.ash_s_defined(fs, 85)                           # ... and ASH-S as defined (the reference permutes kept values within a row)
    rec["ash_s"].update(max_rel_err=_rel(s_ash[:m].cpu().numpy(), exp), cpu_rows_per_s=round(1 / t, 1))
    exp, t = cpu(lambda: oracle.gen_score(ls, 0.1, 100), m)
    rec["gen"].update(max_rel_err=_rel(s_gen[:m].cpu().numpy(), exp), cpu_rows_per_s=round(1 / t, 1))
    mm = 64
    exp, t = cpu(lambda: oracle.predictive_uncertainty(logits[: mm * n_mc].cpu().numpy(), n_mc), mm)
    rec["pred_h_mi"].update(max_rel_err=max(_rel(ph[:mm].cpu().numpy(), exp[0]), _rel(mi[:mm].cpu().numpy(), exp[1])),
                            cpu_rows_per_s=round(1 / t, 1))
    mg = 16
    exp, t = cpu(lambda: oracle.gmm_energy(gmm, fs[:mg]), mg)
    rec["gmm_ddu"].update(max_rel_err=_rel(s_gmm[:mg].cpu().numpy(), exp), cpu_rows_per_s=round(1 / t, 2),
                          cpu_form="torch MultivariateNormal.log_prob on (N, 1, D) + scipy logsumexp, 16 rows")
    return rec


def logsumexp_rows(a):
    from scipy.special import logsumexp

    return logsumexp(a, axis=1)


# ---------------------------------------------------------------------------------------------------------------------
# the harness-shaped workload (VERDICT r4 "next" #3): log_evaluate_larex over a PCA sweep, five postprocessors, two OoD sets
# ---------------------------------------------------------------------------------------------------------------------
LAREX_SWEEP = (2, 4, 8, 16, 32, 64, 128, 256)
LAREX_POSTPROCESSORS = ("KDE", "MD", "cMD", "KNN", "GMM")


def larex_entropies(device, n: int, seed: int, kind: str, n_mc: int = 16, classes: int = N_CLASSES):
    """cfg2-synth entropies (N, 512) f64 + labels: class c's latent maps carry their own per-channel scale pattern; kind "ind",
    "ood_corr" (spatially correlated maps: DropBlock perturbs the channel mean less) or "ood_shift" (+0.5 sigma)."""
    import bench  # synth_latents (cfg2-synth generator of the headline workload)
    from runia_core_amd.inference import LaREMPipeline

    probe = LaREMPipeline(None, None, n_mc, 0.5, 2)
    per = -(-n // classes)
    hs, labs = [], []
    for c in range(classes):
        m = min(per, n - c * per)
        if m <= 0:
            break
        x, r = bench.synth_latents(m, seed + 97 * c, 0.5 if kind == "ood_shift" else 0.0, device, scale=1.0 + 0.06 * c,
                                   corr=0.25 if kind == "ood_corr" else 0.0)
        hs.append(probe.entropy(probe.stack(x, r)))
        labs.append(torch.full((m,), c, dtype=torch.int64))
    h = torch.cat(hs)
    lab = torch.cat(labs)
    perm = torch.randperm(h.shape[0], generator=torch.Generator().manual_seed(seed))
    return h[perm.to(device)].cpu().numpy(), lab[perm].numpy()


def run_larex_eval(device, n_train: int = 50_000, n_valid: int = 10_000, n_ood: int = 10_000, sweep=LAREX_SWEEP,
                   cpu_sample=(6400, 400, 400), cpu_legs: bool = True, log=None) -> dict:
    """One wall clock for the reference's evaluation loop (evaluation/latent_space.py:105-207): the five latent-space
    postprocessors on the full 512-d entropies, then PCA refit + transform + the five again for every n of the sweep, AUROC /
    FPR@95 / AUPR per (OoD set, postprocessor, n), best PCA size per postprocessor, thresholds.  Timed twice through
    runia_core_amd.evaluation.log_evaluate_larex: host arrays in and out as upstream, and device_resident=True.  The oracle's CPU
    form of the same loop runs on a bounded subset; its table is the parity reference for the device run on that subset."""
    from runia_core_amd.evaluation import log_evaluate_larex

    say = log or (lambda *_: None)
    ood_names = ["ood_corr", "ood_shift"]
    cfg = {"ind_dataset": "cfg2-synth", "ood_datasets": ood_names, "n_pca_components": list(sweep), "num_classes": N_CLASSES, "k_neighbors": 50}

    class Cfg:  # the postprocessors read attributes (cfg.num_classes, cfg.k_neighbors)
        pass

    cfg_obj = Cfg()
    for k, v in cfg.items():
        setattr(cfg_obj, k, v)
    t0 = time.perf_counter()
    tr, tr_lab = larex_entropies(device, n_train, 100, "ind")
    va, va_lab = larex_entropies(device, n_valid, 200, "ind")
    ind = {"train latent_space_means": tr, "valid latent_space_means": va, "train labels": tr_lab, "valid labels": va_lab}
    ood = {}
    for i, name in enumerate(ood_names):
        x, lab = larex_entropies(device, n_ood, 300 + 100 * i, name)
        ood[f"{name} latent_space_means"], ood[f"{name} labels"] = x, lab
    t_data = time.perf_counter() - t0
    say(f"larex_eval: entropies of {n_train} + {n_valid} + 2 x {n_ood} images in {t_data:.1f} s")

    def sweep_run(ind_d, ood_d, device_resident, thresholds=True):
        np.random.seed(2024)  # the randomized PCA draws from NumPy's global generator (sklearn's random_state=None)
        torch.cuda.synchronize()
        t = time.perf_counter()
        df, best, thr, _ = log_evaluate_larex(cfg_obj, [], {}, dict(ind_d), dict(ood_d), postprocessors=list(LAREX_POSTPROCESSORS),
                                              device_resident=device_resident, thresholds=thresholds)
        torch.cuda.synchronize()
        return df, best, thr, time.perf_counter() - t

    sweep_run(ind, ood, True)  # warm-up: library load, first-use packs
    df_d, best_d, thr_d, t_dev = sweep_run(ind, ood, True)
    df_h, best_h, thr_h, t_host = sweep_run(ind, ood, False)
    rows = len(df_d)
    cols = ["auroc", "fpr@95", "aupr"]
    same = float(np.max(np.abs(df_d[cols].to_numpy(dtype=np.float64) - df_h.loc[df_d.index, cols].to_numpy(dtype=np.float64))))
    scored = (n_valid + len(ood_names) * n_ood) * len(LAREX_POSTPROCESSORS) * (len(sweep) + 1)
    rec = {"shape": f"train {n_train} / valid {n_valid} / {len(ood_names)} OoD sets of {n_ood} x 512 f64 entropies; PCA sweep {list(sweep)} + the full "
                    f"vectors; postprocessors {list(LAREX_POSTPROCESSORS)}; {rows} table rows",
           "seconds_device_resident": round(t_dev, 3), "seconds_host_arrays_api": round(t_host, 3),
           "rows_scored": scored, "rows_scored_per_s": round(scored / t_dev, 1),
           "table_rows_per_s": round(rows / t_dev, 2), "max_abs_diff_between_the_two_modes": same,
           "best": {k: v["best_comp"] for k, v in best_d.items() if k != "best"},
           "thresholds": {k: float(v) for k, v in thr_d.items()},
           "auroc_of_best": {k: float(v["auroc"]) for k, v in best_d.items() if k != "best"},
           "data_generation_s": round(t_data, 2)}
    if not cpu_legs:
        return rec
    from oracle import harness  # checker / CPU baseline only

    a, b, c = cpu_sample
    ind_s = {"train latent_space_means": tr[:a], "valid latent_space_means": va[:b], "train labels": tr_lab[:a], "valid labels": va_lab[:b]}
    ood_s = {}
    for name in ood_names:
        ood_s[f"{name} latent_space_means"], ood_s[f"{name} labels"] = ood[f"{name} latent_space_means"][:c], ood[f"{name} labels"][:c]
    df_s, _, _, t_dev_s = sweep_run(ind_s, ood_s, True, thresholds=False)
    from runia_core_amd.host_threads import host_compute, usable_cpus

    np.random.seed(2024)
    with host_compute():  # BLAS / torch pools at the container's CPU quota
        t0 = time.perf_counter()
        table, secs = harness.larex_eval_sweep(ind_s, ood_s, ood_names, list(sweep), LAREX_POSTPROCESSORS, N_CLASSES, 50)
        t_cpu = time.perf_counter() - t0
    diffs = {}
    for name, (au, fp, ap) in table.items():
        g = df_s.loc[name]
        pp = name.split()[1]
        diffs.setdefault(pp, [0.0, 0.0, 0.0])
        diffs[pp] = [max(diffs[pp][0], abs(float(g["auroc"]) - au)), max(diffs[pp][1], abs(float(g["fpr@95"]) - fp)),
                     max(diffs[pp][2], abs(float(g["aupr"]) - ap))]
    rec["cpu_baseline"] = {"seconds": round(t_cpu, 2), "cores": usable_cpus(), "kind": "port",
                           "sample": f"the same loop on train {a} / valid {b} / 2 x {c} rows (oracle/harness.py: sklearn randomized PCA, NumPy / "
                                     f"BLAS distances, SciPy pinvh, CPU torch Gaussians); the device run of the SAME subset takes {t_dev_s:.2f} s",
                           "seconds_by_part": {k: round(v, 2) for k, v in secs.items()}, "device_seconds_same_subset": round(t_dev_s, 3)}
    rec["parity"] = {"max_abs_diff_auroc_fpr95_aupr_by_postprocessor": {k: [float(x) for x in v] for k, v in diffs.items()},
                     "rows_compared": len(table), "note": "device table against the oracle's on the subset (float32 metric values; "
                     "KDE is the exact density on both sides - the reference's tree diverges above D ~ 24, INTEGRATION.md)"}
    # the same subset with the fits on the HOST (sklearn / SciPy, what the oracle calls): kNN scores are float32 values crowded
    # near their k-th distance - a PCA fit that differs in the 11th digit (device Jacobi vs LAPACK) moves a few of them by one
    # float32 step and swaps a handful of the 400 x 400 score pairs (one pair = 6e-6 of AUROC)
    from runia_core_amd import config as rc_config

    before = rc_config.device_fit
    rc_config.device_fit = False
    try:
        df_hf, _, _, _ = sweep_run(ind_s, ood_s, True, thresholds=False)
    finally:
        rc_config.device_fit = before
    worst = {}
    for name, (au, fp, ap) in table.items():
        g = df_hf.loc[name]
        pp = name.split()[1]
        worst[pp] = max(worst.get(pp, 0.0), abs(float(g["auroc"]) - au), abs(float(g["fpr@95"]) - fp), abs(float(g["aupr"]) - ap))
    rec["parity"]["max_abs_diff_with_host_fits"] = {k: float(v) for k, v in worst.items()}
    return rec


# ---------------------------------------------------------------------------------------------------------------------
# the baselines harness (round 6): calculate_all_baselines over cfg3-synth features / logits
# ---------------------------------------------------------------------------------------------------------------------
BASELINES_ALL = ("vim", "msp", "raw", "knn", "energy", "ash", "gen", "react", "dice", "dice_react", "mdist", "ddu")


def run_baselines_eval(device, n_train: int = 50_000, n_valid: int = 10_000, n_ood: int = 10_000, names=BASELINES_ALL,
                       cpu_sample=(4000, 256, 256), cpu_legs: bool = True, log=None) -> dict:
    """One wall clock for the reference's baselines loop (evaluation/baselines.py:713-854) on cfg3-synth rows: features
    ReLU(mu_class + N(0, I)) of width 2048, logits of the 10-class slice of the synthetic head (``gen`` refuses more than 21
    classes upstream), two OoD sets (shifted by +0.5 and -0.5).  Timed twice through
    runia_core_amd.evaluation.calculate_all_baselines: host arrays as upstream (every postprocess uploads its rows), and
    device_resident=True (each split uploaded once).  The oracle's CPU form of the loop runs on a bounded subset: parity of the
    device scores on that subset, and the CPU figure."""
    from runia_core_amd.evaluation import calculate_all_baselines

    say = log or (lambda *_: None)
    centres = class_centres(device)
    w_all, b_all = linear_head(device)
    w, b = w_all[:N_CLASSES].contiguous(), b_all[:N_CLASSES].contiguous()
    ood_names = ["ood_up", "ood_down"]
    t0 = time.perf_counter()

    def split(n, which, shift):
        f, _ = feature_rows(0, n, which, device, centres)
        if shift:
            f = torch.relu(f + shift)
        return f.cpu().numpy(), logits_of(f, w, b).cpu().numpy()

    trf, trl = split(n_train, 1, 0.0)
    vaf, val = split(n_valid, 0, 0.0)
    ind = {"train features": trf, "train logits": trl, "valid features": vaf, "valid logits": val}
    ood = {}
    for i, name in enumerate(ood_names):
        f, _ = feature_rows(n_valid + i * n_ood, n_valid + (i + 1) * n_ood, 0, device, centres)
        f = torch.relu(f + (0.5 if i == 0 else -0.5))
        ood[f"{name} features"], ood[f"{name} logits"] = f.cpu().numpy(), logits_of(f, w, b).cpu().numpy()
    fc = {"weight": w.cpu().numpy(), "bias": b.cpu().numpy()}
    cfg = {"ood_datasets": ood_names, "ash_percentile": 90, "react_percentile": 90, "dice_percentile": 90, "gen_gamma": 0.1,
           "k_neighbors": K_NN}
    t_data = time.perf_counter() - t0
    say(f"baselines_eval: {n_train} + {n_valid} + 2 x {n_ood} rows x {D_FEAT} in {t_data:.1f} s")

    def loop(ind_d, ood_d, resident, which=names):
        import contextlib
        import io
        import warnings

        torch.cuda.synchronize()
        t = time.perf_counter()
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):  # (the loop prints one line per baseline, as upstream)
            warnings.simplefilter("ignore")
            i2, _, sc = calculate_all_baselines(list(which), dict(ind_d), dict(ood_d), fc, cfg, N_CLASSES, device_resident=resident)
        torch.cuda.synchronize()
        return i2, sc, time.perf_counter() - t

    loop(ind, ood, True)  # warm-up
    i_d, s_d, t_dev = loop(ind, ood, True)
    i_h, s_h, t_host = loop(ind, ood, False)
    same = max([float(np.max(np.abs(s_d[k].astype(np.float64) - s_h[k].astype(np.float64)))) for k in s_d] +
               [float(np.max(np.abs(i_d[n].astype(np.float64) - i_h[n].astype(np.float64)))) for n in names])
    scored = (n_valid + len(ood_names) * n_ood) * len(names)
    # per-baseline seconds of the device-resident mode (one more pass, one baseline at a time)
    per = {}
    for n in names:
        _, _, per[n] = loop(ind, ood, True, which=(n,))
    vim_opt = None
    if "vim" in names:  # the opt-in device fit of ViM's residual space (config.vim_device_fit): seconds and distance to the default's scores
        from runia_core_amd import config as _config

        _config.vim_device_fit = True
        try:
            loop(ind, ood, True, which=("vim",))
            i_v, s_v, t_v = loop(ind, ood, True, which=("vim",))
        finally:
            _config.vim_device_fit = False
        vim_opt = {"seconds": round(t_v, 3), "seconds_default_host_eig": round(per["vim"], 3),
                   "max_rel_diff_to_default_scores": max(_rel(s_v[f"{o} vim"], s_d[f"{o} vim"]) for o in ood_names),
                   "loop_seconds_with_it": round(t_dev - per["vim"] + t_v, 3)}
    rec = {"shape": f"train {n_train} / valid {n_valid} / {len(ood_names)} OoD sets of {n_ood} x {D_FEAT} f32 features + {N_CLASSES}-class logits; "
                    f"baselines {list(names)}",
           "seconds_device_resident": round(t_dev, 3), "seconds_host_arrays_api": round(t_host, 3), "rows_scored": scored,
           "rows_scored_per_s": round(scored / t_dev, 1), "max_abs_diff_between_the_two_modes": same,
           "seconds_per_baseline": {k: round(v, 3) for k, v in per.items()}, "data_generation_s": round(t_data, 2)}
    if vim_opt:
        rec["vim_device_fit_opt_in"] = vim_opt
    if not cpu_legs:
        return rec
    from oracle import harness  # checker / CPU baseline only
    from runia_core_amd.host_threads import host_compute, usable_cpus

    a, v, c = cpu_sample
    ind_s = {"train features": trf[:a], "train logits": trl[:a], "valid features": vaf[:v], "valid logits": val[:v]}
    ood_s = {f"{n} {kind}": ood[f"{n} {kind}"][:c] for n in ood_names for kind in ("features", "logits")}
    # ddu fits on the WHOLE training split in both legs: ten Gaussians in 2048 dimensions from 400 rows each are singular
    # matrices behind a jitter - their log-densities are the factorisation's round-off, not something to compare
    ind_full = dict(ind_s, **{"train features": trf, "train logits": trl})
    rest = tuple(n for n in names if n != "ddu")
    i_s, s_s, t_dev_s = loop(ind_s, ood_s, True, which=rest)
    if "ddu" in names:
        i_s2, s_s2, t2 = loop(ind_full, ood_s, True, which=("ddu",))
        i_s["ddu"] = i_s2["ddu"]
        s_s.update(s_s2)
        t_dev_s += t2
    secs = {}
    import warnings

    with host_compute(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = time.perf_counter()
        exp = harness.all_baselines(rest, ind_s, ood_s, ood_names, fc["weight"], fc["bias"], N_CLASSES, K_NN, 90, 90, 90, 0.1, secs)
        if "ddu" in names:
            exp.update(harness.all_baselines(("ddu",), ind_full, ood_s, ood_names, fc["weight"], fc["bias"], N_CLASSES, K_NN, 90, 90, 90,
                                             0.1, secs))
        t_cpu = time.perf_counter() - t
    err = {}
    for n in names:
        e = [_rel(i_s[n], exp[n]["valid"])] + [_rel(s_s[f"{o} {n}"], exp[n][o]) for o in ood_names]
        err[n] = float(max(e))
    rec["parity_on_subset"] = {"max_rel_err_per_baseline": {k: float(f"{v:.3g}") for k, v in err.items()},
                               "note": "ash: the oracle's defined op (upstream's scatter permutes kept values in some rows)"}
    rec["cpu_baseline"] = {"seconds": round(t_cpu, 2), "cores": usable_cpus(), "kind": "port",
                           "sample": f"train {a} (ddu: all {n_train}) / valid {v} / 2 x {c} rows of the same splits, oracle.harness.all_baselines (NumPy / SciPy / CPU torch)",
                           "seconds_per_baseline": {k: round(x, 2) for k, x in secs.items()}, "device_seconds_same_subset": round(t_dev_s, 3)}
    return rec
