#!/usr/bin/env python3
"""Headline benchmark: OOD scores/sec on the LaREM-16MC / PCA-256 hot path (BASELINE.json configs[1]).

One "step" = one pass of the scoring hot path over the 10 000 test images of the workload,
inputs resident in HBM:  hooked latent maps (N,512,4,4) f32 + DropBlock draws (N,16,4,4)
  -> MC-dropout latent stacking -> per-dimension KL entropy -> PCA 512->256 (whitened)
  -> LaREM (Mahalanobis) score -> [N>1: one RCCL all_gather of the score shards].
Weak scaling: every rank scores its own 10 000-image shard (rows are independent).
Three distinct input sets (3 x 338 MB > the 256 MiB Infinity Cache) rotate through the timed region, so no step
finds its inputs in a cache.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts its own N ranks (torch.distributed.run as a child
process, before this process makes any GPU call) and exits with the child's code.

Prints ONE JSON line of < 4 KB (rank 0; contract_line below) with the driver's contract plus `roofline`, `cpu_baseline`,
`parity` and `stages` as {leg: [ms, frac]} pairs.  The full record (everything listed below, per-step lists, notes, the harness
table) goes to stderr and to bench_detail.json next to this file (and under gpurun_out/ when that directory exists):
  `roofline`      dominant kernel (K1, sampler + entropy), timed inside the timed region with HIP events attached to its
                  dispatch on its launch stream (the kernel's own start / end timestamps, as rocprofv3's kernel trace
                  reports them; runia_time_next_launch); `achieved` uses BASELINE.md section 4's algorithmic 33 800 B/image;
  `cpu_baseline`  the CPU oracle in the reference's algorithmic form on a bounded sample, 1 core;
  `cpu_baseline_all_cores`  the same work fanned out per image over a process pool, as the reference's
                  parallel_run=True does (evaluation/entropy.py:86-91);
  `api_level`     the same workload through the public batched API with NO caller-supplied draws
                  (LaRExInference.get_scores_from_latents, sampler in counter-draw mode).
The oracle is the checker / baseline only; it is never the thing measured.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MC, C, H, W, N_PCA = 16, 512, 4, 4, 256
DROP_PROB, BLOCK = 0.5, 2
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# BASELINE.md section 4 / SURVEY 8(d): LaREM-16MC/PCA-256 "with MC stacking from latent 512x4x4":
# 512*16*4 (latent map) + 16*16*4 (draws) + 8 (score)
ALGO_BYTES_PER_IMAGE = C * H * W * 4 + N_MC * H * W * 4 + 8
# what K1 itself moves per image: latent map + keep-flag table of the image (n_mc*(H*W+2) floats) + C entropies in f64
K1_BOUNDARY_BYTES_PER_IMAGE = C * H * W * 4 + N_MC * (H * W + 2) * 4 + C * 8


CONTRACT_MAX_BYTES = 4096  # the driver keeps ~8 KB of stdout: the contract line stays well inside it
DETAIL_FILE = "bench_detail.json"
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data")
_ROOFLINE_KEYS = ("bound", "binding_limit", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                  "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_per_step", "launches_timed")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
_CONFIG_MAX_SCALARS = 8


def _clip(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[: n - 3] + "..."


def _num(v, digits=4):
    """Numbers on the contract line carry at most `digits` significant decimals (floats only; ints pass through)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float(f"{v:.{digits + 2}g}") if abs(v) < 1 else round(v, digits)


def stage_pairs(stages, prefix=""):
    """`stages` of the detail record -> {"leg": [ms, frac]} (frac None where the leg has no roofline): every dict that
    carries an "ms" entry is a leg; nested legs are named parent.child."""
    out = {}
    if not isinstance(stages, dict):
        return out
    for k, v in stages.items():
        if not isinstance(v, dict):
            continue
        name = f"{prefix}{k}"
        ms = v.get("ms")
        if ms is None and isinstance(v.get("seconds_device_resident"), (int, float)):
            ms = 1e3 * v["seconds_device_resident"]  # the harness sweep reports seconds
        if isinstance(ms, (int, float)):
            frac = v.get("frac")
            out[name] = [_num(float(ms)), _num(float(frac)) if isinstance(frac, (int, float)) else None]
        out.update(stage_pairs(v, name + "."))
    return out


def contract_line(full) -> str:
    """The ONE line bench.py writes to stdout: the driver's contract keys + `roofline` + `cpu_baseline` + `parity` +
    `stages` as {leg: [ms, frac]} pairs, at most CONTRACT_MAX_BYTES bytes.  Everything else of `full` (per-step lists,
    notes, spreads, the harness table, api_level, ...) goes to stderr and to DETAIL_FILE (emit_record below)."""
    line = {k: (_clip(full.get(k), 160) if k in ("metric", "dtype") else _num(full.get(k), 4)) for k in _TOP_KEYS}
    cfg = full.get("config") or {}
    slim = {"workload": _clip(str(cfg.get("workload", "")), 200)}
    for k, v in cfg.items():
        if k == "workload":
            continue
        if len(slim) > _CONFIG_MAX_SCALARS:
            break
        if isinstance(v, (bool, int, float)) or (isinstance(v, str) and len(v) <= 48):
            slim[k] = _num(v)
    line["config"] = slim
    roof = full.get("roofline")
    line["roofline"] = None if not isinstance(roof, dict) else {
        k: (_clip(roof.get(k), 96) if isinstance(roof.get(k), str) else _num(roof.get(k))) for k in _ROOFLINE_KEYS}
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        line["cpu_baseline"] = {k: (_clip(cpu.get(k), 120) if isinstance(cpu.get(k), str) else _num(cpu.get(k))) for k in _CPU_KEYS}
    par = full.get("parity")
    if isinstance(par, dict):
        errs = [v for k, v in par.items() if k.startswith("max_rel_err") and isinstance(v, (int, float))]
        if not errs:  # cfg3: one relative error per postprocessor
            errs = [v for k, v in par.items() if not k.startswith("auroc") and isinstance(v, float) and 0.0 <= v < 1.0]
        line["parity"] = {"max_rel_err": (float(f"{max(errs):.3g}") if errs else None),
                          "auroc_gpu": par.get("auroc_gpu"), "auroc_oracle": par.get("auroc_oracle")}
    line["detail"] = DETAIL_FILE
    pairs = stage_pairs(full.get("stages"))
    if isinstance(full.get("stages"), dict) and "error" in full["stages"]:
        line["stages_error"] = _clip(str(full["stages"]["error"]), 160)
    if pairs:
        line["stages"] = pairs
    text = json.dumps(line, separators=(",", ":"))
    while len(text.encode()) >= CONTRACT_MAX_BYTES and line.get("stages"):
        line["stages"].popitem()  # legs dropped from the end; all of them stay in DETAIL_FILE
        line["stages_truncated"] = True
        text = json.dumps(line, separators=(",", ":"))
    if len(text.encode()) >= CONTRACT_MAX_BYTES:
        raise AssertionError(f"contract line is {len(text.encode())} bytes (limit {CONTRACT_MAX_BYTES})")
    return text


def emit_record(full, saved_stdout=None):
    """Detail record -> stderr + DETAIL_FILE (next to bench.py, and under gpurun_out/ when that directory exists, so that it comes
    back from a GPU box); then the contract line, alone, as the LAST line of stdout."""
    detail = json.dumps(full)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    f.write(detail + "\n")
        except OSError as e:
            print(f"bench.py: could not write {DETAIL_FILE} under {d}: {e}", file=sys.stderr)
    print("bench detail: " + detail, file=sys.stderr, flush=True)
    text = contract_line(full)
    sys.stdout.flush()
    if saved_stdout is not None:
        os.dup2(saved_stdout, 1)
    print(text, flush=True)
    return text


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "larex_eval", "baselines_eval"], default="cfg2",
                    help="cfg2 (default, the headline metric): LaREM-16MC/PCA-256 from latents, weak scaling.  cfg3: "
                         "BASELINE.json configs[2] - synthetic 1M x 2048 rows through Mahalanobis + Energy + kNN(k=50), "
                         "the rows sharded over the ranks (strong scaling), one all_gather per postprocessor.  larex_eval: the "
                         "reference's evaluation harness loop (log_evaluate_larex: PCA refit sweep x five latent-space postprocessors x "
                         "two OoD sets -> AUROC table), one wall clock for the whole sweep; N > 1 runs independent replicas.  baselines_eval: the "
                         "reference's baselines loop (calculate_all_baselines: twelve features / logits postprocessors fitted and scored on "
                         "cfg3-synth splits), one wall clock for the loop; N > 1 runs independent replicas")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for cfg2, 3 for cfg3)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 10 for cfg2, 1 for cfg3)")
    ap.add_argument("--rows", type=int, default=1_000_000, help="cfg3: rows of the synthetic test set (all ranks together)")
    ap.add_argument("--fit-rows", type=int, default=None,
                    help="cfg3: training rows of the Mahalanobis fit (default: all 50 000 for --workload cfg3, 8 192 in "
                         "the stages block of the default run; the kNN bank always holds 50 000 rows)")
    ap.add_argument("--no-stages", action="store_true",
                    help="cfg2: leave out the `stages` block (cfg3 postprocessors + cfg4 LaRED leg at BASELINE sizes)")
    ap.add_argument("--images", type=int, default=10000, help="test images per GPU (workload: 10 000)")
    ap.add_argument("--train-images", type=int, default=4096)
    ap.add_argument("--input-sets", type=int, default=3,
                    help="distinct resident input sets rotated through the timed region (3 x 338 MB > 256 MiB of L3)")
    ap.add_argument("--draws", choices=["resident", "counter"], default="resident",
                    help="timed region's DropBlock draws: host-supplied tensors resident in HBM (default; the parity "
                         "path) or made inside the keep-flag kernel by the counter generator")
    ap.add_argument("--ood-corr", type=float, default=0.25, help="spatial correlation of the OOD sample's latent maps")
    ap.add_argument("--cpu-sample", type=int, default=1280, help="images timed on the CPU oracle (1 core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-level", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="two HIP streams: K1 of batch i+1 beside K2 of batch i (measured: no gain, 0.324 vs 0.329 ms)")
    ap.add_argument("--chunks", type=int, default=None, help="row blocks pipelined over two streams (default: pipeline's)")
    ap.add_argument("--k0-ahead", action="store_true",
                    help="build the keep-flag table (K0) of batch i+1 on a side stream while batch i is scored "
                         "(LaREMPipeline.prepare_draws).  Measured SLOWER than the in-line launch (0.199-0.201 vs 0.179 "
                         "ms/step: the cross-stream event waits cost more than the 9 us K0 they hide, and K1 slows from "
                         "0.1245 to 0.136 ms with K0 beside it), so it is off by default; profiles/README.md")
    ap.add_argument("--clock-warmup", type=float, default=1.0,
                    help="seconds of the same step run untimed before the W warm-up steps: the GPU needs ~0.1 s of "
                         "sustained load to reach its working clocks (measured: 0.261 ms/step over the first 10 steps, "
                         "0.205 ms/step over 2000)")
    ap.add_argument("--event-every", type=int, default=8,
                    help="the dominant kernel's dispatch carries HIP events on every n-th timed step (events from a pool "
                         "created before the region; such a step measured no slower than its neighbours)")
    ap.add_argument("--gather-stream", choices=["auto", "same", "side", "p2p"], default="auto",
                    help="queue the all_gather on the compute stream, or on a second stream behind an event with "
                         "two output buffers in turn (the next step's kernels then start without waiting for it); "
                         "auto: time both during the warm-up and keep the faster (all ranks agree through a MAX).  "
                         "p2p (opt-in, never chosen by auto): the one-shot gather of csrc/p2p.hip instead of RCCL - every "
                         "rank writes its shard into the peers' IPC-mapped buffers over xGMI, two launches on the compute "
                         "stream (runia_core_amd.distributed.OneShotGather)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"cfg2": 200, "cfg3": 3, "larex_eval": 1, "baselines_eval": 1}[args.workload]
    if args.warmup is None:
        args.warmup = {"cfg2": 10, "cfg3": 1, "larex_eval": 1, "baselines_eval": 1}[args.workload]
    return args


def self_launch(args) -> int:
    """--gpus N > 1 without a launcher: start N ranks as a CHILD process (never an exec, and before this process has
    touched the GPU - importing torch does not initialise it) and hand its exit code back."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def effective_cpus() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def cpu_pool_child(path, cores, n_mc, drop_prob, block):
    """cpu_baseline leg, parallel form (a child process of bench.py that never imports torch or touches the GPU):
    the reference fans get_dl_h_z out per image with process_map(single_image_entropy_calculation, ..., chunksize=1)
    (evaluation/entropy.py:86-91); here one task = oracle sampler + per-dimension k-d-tree entropy of one image over a
    multiprocessing.Pool(cores); PCA + LaREM then run once on the gathered rows, as in the reference."""
    import multiprocessing as mp

    import numpy as np

    import oracle  # checker / CPU baseline only

    d = np.load(path)
    x, rand = d["x"], d["rand"]
    k = 5 if n_mc > 5 else n_mc - 1
    global _pool_task

    def _pool_task(i):
        z = oracle.mc_stack(x[i : i + 1], rand[i], drop_prob, block)
        return oracle.single_image_entropy_calculation(z, k)

    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        h = np.asarray(pool.map(_pool_task, range(x.shape[0]), chunksize=1))
    y = oracle.pca_transform(h, d["comp"], d["mean"], d["var"])
    s = oracle.md_score(y, d["md_mean"], d["md_prec"])
    sec = time.perf_counter() - t0
    print(json.dumps({"seconds": sec, "images": int(x.shape[0]), "cores": cores, "scores_head": s[:16].tolist()}))


def synth_latents(n, seed, shift, device, scale=1.0, corr=0.0, dead="ones"):
    """cfg2-synth: X ~ ReLU(N(shift,1)) on (n,512,4,4) with a fixed per-channel scale; draws U(0,1) on (n,16,4,4).
    Draws whose block mask would drop the whole 4x4 map (sum(bm)=0 -> inf/NaN in the reference
    as well) are replaced by "no seed" so that every score is finite."""
    import torch

    g0 = torch.Generator(device=device).manual_seed(77)  # per-channel scale: a property of the "layer", same for all sets
    chan = torch.rand(1, C, 1, 1, device=device, generator=g0) * 1.5 + 0.25
    g = torch.Generator(device=device).manual_seed(seed)
    noise = torch.randn(n, C, H, W, device=device, generator=g)
    if corr > 0.0:  # OOD variant: part of every map is spatially constant -> DropBlock perturbs the channel mean less
        shared = torch.randn(n, C, 1, 1, device=device, generator=g)
        noise = corr * shared + (1.0 - corr * corr) ** 0.5 * noise
    x = torch.relu(noise * (chan * scale) + shift).contiguous()
    rand = torch.rand(n, N_MC, H, W, device=device, generator=g)
    for _ in range(16 if dead == "redraw" else 1):
        mask = (rand < DROP_PROB / BLOCK**2).float().reshape(n * N_MC, 1, H, W)
        bm = 1 - torch.nn.functional.max_pool2d(mask, BLOCK, 1, BLOCK // 2)[:, :, :-1, :-1]
        gone = bm.sum(dim=(1, 2, 3)) == 0
        if dead == "keep":
            break
        if dead == "redraw":  # the policy of CounterDraws(redraw_dead_layers=True): such a layer draws again
            if not bool(gone.any()):
                break
            rand.reshape(n * N_MC, H, W)[gone] = torch.rand(int(gone.sum()), H, W, device=device, generator=g)
        else:
            rand.reshape(n * N_MC, H, W)[gone] = 1.0
    return x, rand.contiguous()


def main_cfg3(args, device, rank, world, dist, saved_stdout):
    """--workload cfg3: BASELINE.json configs[2].  A step = this rank's block of the 1M x 2048 synthetic rows (and of
    their 1M x 1000 / 1M x 10 logits) through Mahalanobis, Energy and kNN(k=50), one all_gather of the score shards per
    postprocessor.  The rows are sharded (strong scaling): `value` = rows of the WHOLE set scored by all three
    postprocessors per second."""
    import gc

    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_workloads as bw

    gc.collect()
    gc.disable()
    log = (lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None
    rec = bw.run_cfg3(device, rank, world, dist, args.rows, args.fit_rows or bw.BANK_ROWS, args.steps, args.warmup,
                      cpu_legs=not args.no_cpu_baseline, log=log)
    gc.enable()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    st = rec["stages"]
    knn = st["knn"]
    out = {
        "metric": "OOD scores/sec, Mahalanobis + Energy + kNN(k=50) on synthetic 1M x 2048 features",
        "value": round(rec["value"], 1), "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(rec["ms_per_step"], 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64 (Mahalanobis) / f32 (Energy, kNN)" + (
            "; kNN scores are exactly re-measured f32 distances, the bank rows to re-measure are ranked by bf16 piece products "
            "(same bits as with the f32 matrix-core kernel: tests/test_hip_kernels.py::test_knn_bf16_candidate_distances_equal_the_f32_path)"
            if st["knn"].get("piece_products") else ""), "data": "synthetic",
        "config": {"workload": "Synthetic 1M x 2048 features, Mahalanobis + Energy + kNN(k=50) postprocessors, rows sharded "
                               "over the GPUs (BASELINE.json configs[2])",
                   "rows": rec["rows_total"], "rows_per_gpu": rec["rows_local"], "features": bw.D_FEAT,
                   "classes": bw.N_CLASSES, "logits": [bw.N_LOGITS, bw.N_CLASSES], "knn_bank": [bw.BANK_ROWS, bw.D_FEAT],
                   "k": bw.K_NN, "mahalanobis_fit_rows": rec["fit_rows"], "entry": "postprocess_device (rows resident in HBM)",
                   "gather": "none (1 GPU)" if dist is None else "one all_gather_into_tensor per postprocessor, compute stream",
                   "gather_ms_per_call": round(rec["gather_ms_per_call"], 4),
                   "setup_fit_s": round(rec["fit_s"], 2), "setup_broadcast_s": round(rec["broadcast_s"], 3),
                   "setup": {"mode": rec.get("fit_mode"), "fit_s": None if rec.get("fit_only_s") is None else round(rec["fit_only_s"], 3),
                             "fit_s_host_calls": None if rec.get("fit_only_s_host_calls") is None else round(rec["fit_only_s_host_calls"], 3),
                             "note": "setup() of Mahalanobis (covariance + pinvh of 2048 x 2048), KNN (bank) and Energy on host arrays; "
                                     "device = covariance on the f64 matrix cores + blocked Jacobi pinvh (runia_core_amd.config.device_fit, "
                                     "default where a GPU is present), host calls = the reference's scikit-learn / SciPy"}},
        "roofline": {"bound": "mfma",
                     "kernel": ("knn_dist_bf16_kernel" if knn.get("piece_products") else "knn_dist_kernel")
                     + " (+ normaliser, piece split and k-th select with exact re-measurement: the whole kNN stage is timed)",
                     "achieved": knn["achieved"], "peak": knn["peak"], "unit": "TFLOP/s", "frac": knn["frac"], "traffic": None,
                     "dtype_of_peak": "bf16 dense" if knn.get("piece_products") else "f32",
                     "piece_products": knn.get("piece_products", 0),
                     "f32_equivalent_tflops": knn.get("f32_equivalent_tflops", knn["achieved"]),
                     "algorithmic_flop_per_row": 2.0 * bw.BANK_ROWS * bw.D_FEAT, "avg_stage_ms": knn["ms"],
                     "binding_limit": "mfma", "avg_launch_ms": knn["ms"], "launches_per_step": 1, "launches_timed": args.steps,
                     "share_of_step": round(knn["ms"] / max(1e-9, sum(v["ms"] for v in st.values())), 4)},
        "stages": st,
    }
    for k in ("cpu_baseline", "parity"):
        if k in rec:
            out[k] = rec[k]
    emit_record(out, saved_stdout)
    if dist is not None:
        dist.destroy_process_group()


def main_larex(args, device, rank, world, dist, saved_stdout):
    """--workload larex_eval: a step = one full pass of the reference's evaluation loop (evaluation/latent_space.py:105-207)
    over train 50 000 / valid 10 000 / two OoD sets of 10 000 x 512 entropies.  The loop does not shard (every fit needs the
    whole training split): N > 1 runs N independent replicas ("replicas only")."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_workloads as bw

    log = (lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None
    rec = bw.run_larex_eval(device, cpu_legs=(rank == 0 and not args.no_cpu_baseline), log=log)
    if dist is not None:
        t = torch.tensor([rec["seconds_device_resident"]], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rec["seconds_device_resident"] = float(t.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    sec = rec["seconds_device_resident"]
    out = {"metric": "OOD scores/sec through the evaluation harness loop (log_evaluate_larex: PCA sweep x 5 latent-space postprocessors x 2 OoD sets)",
           "value": round(world * rec["rows_scored"] / sec, 1), "unit": "rows scored/s", "n_gpus": world, "steps": 1, "warmup": 1,
           "ms_per_step": round(1e3 * sec, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64 (KDE, MD) / f32 (cMD, KNN, GMM), as the reference's postprocessors return", "data": "synthetic",
           "config": {"workload": "log_evaluate_larex on cfg2-synth entropies (CIFAR10 ResNet-18 LaREx sizes): " + rec["shape"],
                      "multi_gpu": "replicas only (the loop's fits need the whole training split)"},
           "roofline": None, "larex_eval": rec}
    if "cpu_baseline" in rec:
        out["cpu_baseline"] = {"value": None, "unit": "s (bounded subset, see sample)", "cores": rec["cpu_baseline"]["cores"], "kind": "port",
                               "sample": rec["cpu_baseline"]["sample"], "seconds": rec["cpu_baseline"]["seconds"],
                               "device_seconds_same_subset": rec["cpu_baseline"]["device_seconds_same_subset"]}
    emit_record(out, saved_stdout)
    if dist is not None:
        dist.destroy_process_group()


def main_baselines(args, device, rank, world, dist, saved_stdout):
    """--workload baselines_eval: a step = one full pass of the reference's baselines loop (evaluation/baselines.py:713-854) over
    train 50 000 / valid 10 000 / two OoD sets of 10 000 x 2048 features + 10-class logits.  The loop does not shard (every fit
    needs the whole training split): N > 1 runs N independent replicas ("replicas only")."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_workloads as bw

    log = (lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None
    rec = bw.run_baselines_eval(device, cpu_legs=(rank == 0 and not args.no_cpu_baseline), log=log)
    if dist is not None:
        t = torch.tensor([rec["seconds_device_resident"]], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rec["seconds_device_resident"] = float(t.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    sec = rec["seconds_device_resident"]
    out = {"metric": "OOD scores/sec through the baselines harness loop (calculate_all_baselines: 12 features / logits postprocessors x (valid + 2 OoD sets))",
           "value": round(world * rec["rows_scored"] / sec, 1), "unit": "rows scored/s", "n_gpus": world, "steps": 1, "warmup": 1,
           "ms_per_step": round(1e3 * sec, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (f64 Mahalanobis), as the reference's postprocessors return", "data": "synthetic",
           "config": {"workload": "calculate_all_baselines on cfg3-synth splits: " + rec["shape"],
                      "multi_gpu": "replicas only (the loop's fits need the whole training split)"},
           "roofline": None, "baselines_eval": rec}
    if "cpu_baseline" in rec:
        out["cpu_baseline"] = {"value": None, "unit": "s (bounded subset, see sample)", "cores": rec["cpu_baseline"]["cores"], "kind": "port",
                               "sample": rec["cpu_baseline"]["sample"], "seconds": rec["cpu_baseline"]["seconds"],
                               "device_seconds_same_subset": rec["cpu_baseline"]["device_seconds_same_subset"]}
        errs = rec["parity_on_subset"]["max_rel_err_per_baseline"]
        out["parity"] = {"max_rel_err": max(errs.values()), **{f"max_rel_err_{k}": v for k, v in errs.items()}}
    emit_record(out, saved_stdout)
    if dist is not None:
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-pool-child":
        return cpu_pool_child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]))
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # nothing in this process has touched the GPU yet
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch

    # Exactly one line may reach stdout (the JSON record): RCCL prints its version banner to stdout when
    # NCCL_DEBUG=VERSION is set, so fd 1 points at stderr until the record is written.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RUNIA_BENCH_REHEARSE=gloo: several ranks share the visible GPUs over gloo (control-flow rehearsal of the N > 1
    # path on a one-GPU box; RCCL refuses two ranks on one device).  Never set by the driver.
    rehearse = os.environ.get("RUNIA_BENCH_REHEARSE")
    if rehearse:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("RUNIA_BENCH_FORCE_DIST"))  # the env var rehearses the RCCL path at N = 1
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group(backend=rehearse, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)

    import runia_core_amd as rc
    from runia_core_amd import _hip
    from runia_core_amd.inference import LaREMPipeline, MDLatentSpace

    _hip.require_gpu()
    if args.workload == "cfg3":
        return main_cfg3(args, device, rank, world, dist if use_dist else None, saved_stdout)
    if args.workload == "larex_eval":
        return main_larex(args, device, rank, world, dist if use_dist else None, saved_stdout)
    if args.workload == "baselines_eval":
        return main_baselines(args, device, rank, world, dist if use_dist else None, saved_stdout)

    # ---------------- setup (untimed): fit PCA-256 + LaREM on in-distribution entropies ----------
    probe = LaREMPipeline(None, None, N_MC, DROP_PROB, BLOCK)
    xtr, rtr = synth_latents(args.train_images, 1234, 0.0, device)
    h_train = probe.entropy(probe.stack(xtr, rtr)).cpu().numpy()
    del xtr, rtr
    from runia_core_amd import config as rc_config

    def fit_cfg2():
        np.random.seed(1234)  # sklearn's randomized SVD draws from the global NumPy state (the device fit makes the same draw)
        t_f = time.perf_counter()
        red_, pca_ = rc.apply_pca_ds_split(h_train, N_PCA)
        md_ = MDLatentSpace()
        md_.setup(red_)
        torch.cuda.synchronize()
        return red_, pca_, md_, time.perf_counter() - t_f

    red, pca, md, setup_s = fit_cfg2()
    setup_rec = {"mode": "device" if rc_config.use_device_fit() else "host", "fit_s": round(setup_s, 3),
                 "what": f"apply_pca_ds_split({h_train.shape[0]} x {h_train.shape[1]} -> {N_PCA}, randomized) + MDLatentSpace.setup"}
    if rc_config.use_device_fit() and rank == 0 and not args.no_cpu_baseline:
        before = rc_config.device_fit
        rc_config.device_fit = False  # the reference's own host calls beside it (reported, not used)
        try:
            setup_rec["fit_s_host_calls"] = round(fit_cfg2()[3], 3)
        finally:
            rc_config.device_fit = before
    pipe = LaREMPipeline(md, pca, N_MC, DROP_PROB, BLOCK)

    n = args.images
    n_sets = max(1, args.input_sets)
    # this rank's shard, n_sets distinct realisations of it (set 0 is the one the parity leg scores)
    sets = [synth_latents(n, 1235 + rank + 1000 * j, 0.0, device) for j in range(n_sets)]
    x, rand = sets[0]
    counter = args.draws == "counter"

    k1_events_region = []  # (start, end) HIP event pairs around every K1 launch of the timed region

    torch.cuda.synchronize()
    inputs_ready = torch.cuda.current_stream().record_event()  # the input sets are resident from here on

    gathered = [torch.empty(world * n, dtype=torch.float64, device=device) for _ in range(2)] if use_dist else None
    side_stream = torch.cuda.Stream() if (use_dist and args.gather_stream in ("auto", "side")) else None
    one_shot = None
    if use_dist and args.gather_stream in ("auto", "p2p"):
        from runia_core_amd.distributed import OneShotGather

        try:  # collective; every rank succeeds or every rank raises (distributed.OneShotGather)
            one_shot = OneShotGather(n, torch.float64)
        except _hip.RuniaHipError as e:
            if args.gather_stream == "p2p":
                raise
            if rank == 0:
                print(f"one-shot gather unavailable on this node, RCCL only: {e}", file=sys.stderr)
    # "kind": how the score shards are gathered - RCCL all_gather on the compute stream ("same"), on a second stream
    # ("side"), or the one-shot P2P writes of csrc/p2p.hip ("p2p")
    gather_mode = {"kind": args.gather_stream if args.gather_stream in ("same", "side", "p2p") else "same"}
    gather_turn = [0]
    step_no = [0]

    def gather(s):
        """The single RCCL all_gather of the path: equal shards, preallocated output.  Default: queued
        on the compute stream.  --gather-stream side: queued on a second stream behind an event; outputs alternate
        between two buffers (a buffer is rewritten two steps later, in stream order on the same stream)."""
        if gather_mode["kind"] == "p2p":
            return one_shot(s, world * n)
        out = gathered[gather_turn[0]]
        gather_turn[0] ^= 1
        if gather_mode["kind"] != "side":
            dist.all_gather_into_tensor(out, s)
            return out
        side_stream.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(side_stream):
            dist.all_gather_into_tensor(out, s)
        s.record_stream(side_stream)
        return out

    k0_ahead = args.k0_ahead and not args.overlap and args.chunks is None
    pending = {}  # step number -> keep-flag table prepared for it on the side stream

    def draws_of(no):
        """Draws of step number `no` (1-based, as counted by step_no after its increment)."""
        return _hip.CounterDraws(4242 + rank, no * n) if counter else sets[(no - 1) % n_sets][1]

    k1_events_outside = []  # the same brackets on untimed steps (warm-up, and a short loop right after the timed region)

    def step(timed=False, index=0, collective=True, bracket=False):
        timed = (timed and (index % max(1, args.event_every) == 0)) or bracket
        k1_events = k1_events_region if not bracket else k1_events_outside
        j = step_no[0] % n_sets
        step_no[0] += 1
        xs, rs = sets[j]
        if counter:
            rs = draws_of(step_no[0])  # fresh image ids every step
        if not args.overlap:
            prep = pending.pop(step_no[0], None) if k0_ahead else None
            s = pipe.score_latents(xs, rs, chunks=args.chunks, k1_events=k1_events if timed else None, prepared=prep)
            if k0_ahead:  # the NEXT step's table: a side stream builds it under this step's kernels
                pending.clear()
                pending[step_no[0] + 1] = pipe.prepare_draws(draws_of(step_no[0] + 1), n, H, W, inputs_ready=inputs_ready)
            return gather(s) if (use_dist and collective) else s
        # streaming form: K1 of this batch overlaps K2 of the previous one (two HIP streams); the gather is
        # queued behind this batch's K2 on the same stream, nothing waits on the host until the final sync
        a = pipe.score_latents_async(xs, rs, k1_events=k1_events if timed else None, inputs_ready=inputs_ready)
        if use_dist:
            with torch.cuda.stream(pipe.k2_stream):
                return gather(a.scores)
        return a.scores

    # The interpreter's cyclic GC stays out of everything from here on: a gen-2 pass over the torch / sklearn heap
    # takes 60-80 ms, during which the GPU sits idle and drops its clocks (and inside the timed region it would be
    # charged to ~300 steps).  It runs once now, BEFORE the warm-up, so that nothing idles the GPU between the
    # warm-up and the timed region.
    import gc

    gc.collect()
    gc.disable()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.clock_warmup and not args.overlap:
        # untimed: bring the clocks up with the step's own kernels.  No collective in here: the trip count of a
        # time-based loop differs from rank to rank, and mismatched collectives deadlock.
        for _ in range(50):
            step(collective=False)
        torch.cuda.synchronize()
    _hip.reserve_timed_events(args.warmup + args.steps + 40)  # (the brackets' events exist before the timed region: no record inside it)
    for _ in range(args.warmup):
        step(bracket=not args.overlap)   # the W warm-up steps carry the K1 brackets too (more launches behind avg_launch_ms)
    if use_dist:
        assert step().shape == (world * n,)
    gather_info = None
    if use_dist:
        gather_info = {"requested": args.gather_stream, "trial_ms_per_step": None, "one_shot_candidate": one_shot is not None,
                       "one_shot_agrees_with_rccl": None}
    if use_dist and args.gather_stream == "auto" and not args.overlap:
        # untimed: which placement of the collective is faster on this node?  (one-GPU rehearsal: same stream
        # +9 us, side stream +23 us per step; with real peers the same-stream form also exposes the ring latency)
        kinds = ["same", "side"]
        if one_shot is not None:
            # the one-shot form is a candidate only if it returns what RCCL returns, on every rank
            probe = torch.arange(n, dtype=torch.float64, device=device) * 0.25 + float(rank * n)
            want = torch.empty(world * n, dtype=torch.float64, device=device)
            dist.all_gather_into_tensor(want, probe)
            got = one_shot(probe, world * n)
            agree = torch.tensor([1.0 if bool(torch.equal(got, want)) else 0.0], device=device)
            try:
                one_shot.check()
            except _hip.RuniaHipError:
                agree.zero_()
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            gather_info["one_shot_agrees_with_rccl"] = float(agree.item()) == 1.0
            if float(agree.item()) == 1.0:
                kinds.append("p2p")
        trial = {k: float("inf") for k in kinds}
        for kind in kinds * 3:  # best of three interleaved trials per form
            gather_mode["kind"] = kind
            for _ in range(3):
                step()
            dist.barrier()
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            t_m = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device=device)
            dist.all_reduce(t_m, op=dist.ReduceOp.MAX)
            trial[kind] = min(trial[kind], float(t_m.item()))
        best = min(kinds, key=lambda k: trial[k])
        gather_mode["kind"] = best if trial[best] < 0.98 * trial["same"] else "same"  # the simplest form unless clearly slower
        gather_info["trial_ms_per_step"] = {k: round(trial[k] / 30 * 1e3, 4) for k in kinds}
        gather_info["trial"] = "best of 3 interleaved runs of 30 steps per form, MAX over ranks; a form replaces `same` only if > 2 % faster"
        if rank == 0:
            print("gather trial: " + ", ".join(f"{k} {trial[k] / 30 * 1e3:.4f} ms/step" for k in kinds) +
                  f" -> {gather_mode['kind']}", file=sys.stderr)
    step_no[0] = 0  # the timed region starts on set 0 and ends on set (K-1) % n_sets on every rank
    pending.clear()
    if k0_ahead:
        pending[1] = pipe.prepare_draws(draws_of(1), n, H, W, inputs_ready=inputs_ready)  # untimed, like the inputs being resident: step 1's table
    if use_dist:
        dist.barrier()
    # Clock readings bracket the timed region (runia_clock_probe: ~30 us of one wave).  Each probe is queued DIRECTLY BEHIND
    # steps of the same kind with no host synchronisation between - a probe launched after a synchronisation finds the GPU
    # already idle and reads a clock in transition (tools/microbench/clock_probe.py: 2.16-2.37 GHz there against 2.42-2.44
    # directly behind the load).  Both sit outside the timed region.
    for _ in range(3):
        step(collective=False)
    probe_before = _hip.clock_probe()
    torch.cuda.synchronize()
    step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    step_marks[0].record()
    for i in range(args.steps):
        scores = step(True, i)
        step_marks[i + 1].record()  # one event record per step on the compute stream: per-step times (min / median / max)
    torch.cuda.synchronize()  # every stream of the device, the gather stream included
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for i in range(32):  # more K1 launches under brackets, untimed, right after the region (same clocks, same inputs)
        step(collective=False, bracket=not args.overlap)
    probe_after = _hip.clock_probe()  # directly behind those steps
    torch.cuda.synchronize()
    per_step_ms = [step_marks[i].elapsed_time(step_marks[i + 1]) for i in range(args.steps)]
    clocks = {"before": _hip.clock_ghz(probe_before), "after": _hip.clock_ghz(probe_after),
              "how": "runia_clock_probe (shader-clock counter / 100 MHz counter over ~30 us of one wave) queued directly behind 3 "
                     "untimed steps before the region and behind the 32 bracketed steps after it; idle reads 2.400"}
    last_set = (args.steps - 1) % n_sets
    if use_dist:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)  # every rank's own clock around the same K steps (the line reports the MAX)
        every = torch.cat(parts)
        elapsed = float(every.max().item())
        gather_info["chosen"] = gather_mode["kind"]
        gather_info["per_rank_ms_per_step"] = [round(1e3 * float(v) / args.steps, 4) for v in every.tolist()]
        # the same steps without their collective, right after the timed region: what the gather costs on this node
        dist.barrier()
        torch.cuda.synchronize()
        t_n = time.perf_counter()
        for i in range(min(args.steps, 200)):
            step(collective=False)
        torch.cuda.synchronize()
        t_n = torch.tensor([time.perf_counter() - t_n], dtype=torch.float64, device=device)
        dist.all_reduce(t_n, op=dist.ReduceOp.MAX)
        gather_info["ms_per_step_without_gather"] = round(1e3 * float(t_n.item()) / min(args.steps, 200), 4)
        gather_info["gather_ms_per_step"] = round(1e3 * elapsed / args.steps - gather_info["ms_per_step_without_gather"], 4)

    if one_shot is not None:
        if gather_mode["kind"] == "p2p":
            one_shot.check()   # a wait that gave up on a peer would have produced garbage: fail loudly
        one_shot.close()
    k1_in = [a.elapsed_time(b) for a, b in k1_events_region]
    k1_out = [a.elapsed_time(b) for a, b in k1_events_outside]
    k1_all = k1_in + k1_out
    kernel_ms = float(np.mean(k1_all)) if k1_all else float('nan')
    bracketed_steps = len(range(0, args.steps, max(1, args.event_every)))
    k1_launches_per_step = max(1, len(k1_in) // max(1, bracketed_steps))
    if rank != 0:
        gc.enable()
        dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * n * args.steps / elapsed

    # ---------------- API level: the public batched entry point, no caller-supplied draws ---------------------------
    api = None
    if not args.no_api_level and not use_dist:
        from runia_core_amd import LaRExInference, MCSamplerModule

        infer = LaRExInference(torch.nn.Identity(), md, DROP_PROB, BLOCK, N_MC, MCSamplerModule, pca_transform=pca)
        infer.mc_sampler.use_counter_draws(seed=2026, redraw_dead_layers=True)
        k_api = max(10, min(args.steps, 200))
        infer.get_scores_from_latents(sets[0][0], to_host=False)  # folds the weights (device Jacobi eigh, pipeline.py): the GPU idles meanwhile
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.clock_warmup:   # ... so bring the clocks back up, as before the main region
            for i in range(50):
                infer.get_scores_from_latents(sets[i % n_sets][0], to_host=False)
            torch.cuda.synchronize()
        t_a = time.perf_counter()
        for i in range(k_api):
            s_api = infer.get_scores_from_latents(sets[i % n_sets][0], to_host=False)
        torch.cuda.synchronize()
        t_api = time.perf_counter() - t_a
        t_b = time.perf_counter()
        for i in range(k_api):
            s_host = infer.get_scores_from_latents(sets[i % n_sets][0])  # default: (N,) ndarray on the host, one sync per call
        t_host = time.perf_counter() - t_b
        api = {
            "entry": "LaRExInference.get_scores_from_latents(latents), sampler.use_counter_draws(seed, redraw_dead_layers=True): "
                     "no caller-supplied draws (Philox4x32-10 inside the keep-flag kernel; a drop layer that removes the "
                     "whole map draws again, so no score is NaN)",
            "value": round(n * k_api / t_api, 1), "unit": "images/s", "ms_per_call": round(1e3 * t_api / k_api, 4),
            "returns": "device tensor (to_host=False)", "calls": k_api, "frac_of_value": round(n * k_api / t_api / value, 4),
            "value_scores_to_host": round(n * k_api / t_host, 1),
            "nan_scores_in_last_call": int(np.isnan(s_host).sum()),
        }
    gc.enable()

    # ---------------- roofline of the dominant kernel ------------------------------------------
    # HIP events attached to K1's dispatch on its launch stream (pipeline.k1_events, _hip._timed_launch_events: the kernel's own
    # start / end timestamps - an event pair recorded AROUND the launch also counts the ~7 us dispatch gap behind K0, which is
    # how rounds 3-4 read 0.1113 / 0.1179 ms for a kernel the tracer saw at 0.1073 / 0.1086).  `achieved` = BASELINE.md section 4's
    # algorithmic bytes (33 800 B/image) per launch / that time; the kernel-boundary figure (what K1 itself reads and
    # writes, incl. the f64 entropy rows K2' re-reads) is reported beside it.
    kname = "mc_entropy_kernel"
    imgs_per_launch = n / k1_launches_per_step
    achieved = ALGO_BYTES_PER_IMAGE * imgs_per_launch / (kernel_ms * 1e-3) / 1e9
    boundary = K1_BOUNDARY_BYTES_PER_IMAGE * imgs_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src, valu = None, None, None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            rec = json.load(open(pmc_file)).get(kname, {})
            # PMC bytes were collected at `images_per_launch` images per launch; scale to this run's launch size
            traffic = int(rec["hbm_bytes_per_launch"] * imgs_per_launch / rec.get("images_per_launch", 10000))
            traffic_src = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate "
                           f"passes of this command ({rec.get('profile', 'see profiles/README.md')}); not measured in this run")
            if "valu_insts_per_launch" in rec:
                # VALU-issue floor: instructions x 4 issue cycles / 1024 SIMDs / the clock the kernel holds (measured:
                # GRBM_GUI_ACTIVE / 8 over the launches' durations, profiles/r5b_driver_pmc_summary.json; 2.4 GHz if not recorded)
                clock_ghz = float(rec.get("clock_ghz_held", 2.4))
                floor_ms = rec["valu_insts_per_launch"] * imgs_per_launch / rec.get("images_per_launch", 10000) * 4 / 1024 / (clock_ghz * 1e9) * 1e3
                # ... and at what K1's own instruction mix costs on gfx950 when every instruction reads three different
                # registers and kinds alternate (tools/microbench/valu_issue.hip, profiles/README.md): 4.46 cycles
                floor_mix_ms = floor_ms * 4.46 / 4.0
                valu = {"insts_per_launch": int(rec["valu_insts_per_launch"] * imgs_per_launch / rec.get("images_per_launch", 10000)),
                        "floor_ms": round(floor_ms, 4), "frac_of_floor": round(floor_ms / kernel_ms, 4),
                        "floor_ms_at_measured_mix_cost": round(floor_mix_ms, 4),
                        "frac_of_floor_at_measured_mix_cost": round(floor_mix_ms / kernel_ms, 4),
                        "clock_ghz": clock_ghz, "clock_source": rec.get("clock_source", "nominal"),
                        "cycles_per_instruction": {"nominal": 4.0, "measured_for_this_mix": 4.46,
                                                   "source": "profiles/r2_valu_issue_microbench.txt"},
                        "source": "SQ_INSTS_VALU, same profile"}
        except Exception:
            traffic = None
    roofline = {
        # what limits the dominant kernel: vector-instruction issue (SQ_INSTS_VALU x issue cycles, see valu_issue), not
        # HBM - the kernel moves 1.03 x its boundary bytes and the same loads without the arithmetic run at 5 TB/s.
        # achieved / peak / frac stay in the contract's HBM terms (algorithmic bytes per launch / kernel time / 8 TB/s);
        # bound_frac is the fraction of the BINDING limit (instruction floor / kernel time).
        # `bound` names the roofline that achieved / peak / frac are quoted against (the contract's "hbm" | "mfma");
        # `binding_limit` names what actually limits the kernel.
        "bound": "hbm",
        "binding_limit": "valu-issue" if valu else "hbm",
        "bound_frac": valu["frac_of_floor_at_measured_mix_cost"] if valu else round(achieved / HBM_PEAK_GBS, 4),
        "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": int(ALGO_BYTES_PER_IMAGE * imgs_per_launch),
        "algorithmic_bytes_per_image": ALGO_BYTES_PER_IMAGE,
        "kernel_boundary_bytes_per_image": K1_BOUNDARY_BYTES_PER_IMAGE,
        "kernel_boundary_gbs": round(boundary, 1), "kernel_boundary_frac": round(boundary / HBM_PEAK_GBS, 4),
        "limiter": "valu-issue (the kernel is instruction-bound, not HBM-bound; see valu_issue)" if valu else None,
        "valu_issue": valu,
        "step_frac_of_hbm_ceiling": round(value / world / (HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_IMAGE), 4),
        "avg_launch_ms": round(kernel_ms, 4),
        "launches_per_step": k1_launches_per_step, "launches_timed": len(k1_all),
        "launch_ms": {"min": round(min(k1_all), 4), "median": round(float(np.median(k1_all)), 4), "max": round(max(k1_all), 4),
                      "in_timed_region": {"launches": len(k1_in), "mean": round(float(np.mean(k1_in)), 4) if k1_in else None},
                      "bracketed_untimed_steps": {"launches": len(k1_out), "mean": round(float(np.mean(k1_out)), 4) if k1_out else None,
                                                  "where": "the W warm-up steps + 32 steps right after the timed region"}},
        "clock_ghz_observed": clocks,
    }

    # ---------------- parity on a bounded sample + CPU baseline (oracle = checker / baseline only) --
    out = {
        "metric": "OOD scores/sec, LaREM 16-MC PCA-256 (ResNet-18 layer4 latent 512x4x4)",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "ms_per_step_stats": {"min": round(min(per_step_ms), 4), "median": round(float(np.median(per_step_ms)), 4),
                              "max": round(max(per_step_ms), 4), "mean_of_events": round(float(np.mean(per_step_ms)), 4),
                              "per_step": [round(v, 4) for v in per_step_ms] if len(per_step_ms) <= 64 else None,
                              "bracketed_steps": list(range(0, args.steps, max(1, args.event_every))),
                              "how": "one HIP event per step on the compute stream (rank 0); ms_per_step above is the wall clock / K, MAX over ranks"},
        "clock_ghz_observed": clocks,
        "config": {"workload": "CIFAR10 ResNet-18 LaREM: 16 MC samples, 512-d latent -> PCA-256, 10000 test images per GPU",
                   "images_per_gpu": n, "mc_samples": N_MC, "latent": [C, H, W], "pca_components": N_PCA,
                   "row_blocks_per_step": k1_launches_per_step, "input_dtype": "f32",
                   "draws": ("in-kernel counter generator (Philox4x32-10), nothing read for them" if counter else
                             "host-supplied (N,16,4,4) f32 tensors, resident in HBM before the timed region"),
                   "input_sets_rotated": n_sets, "input_bytes_per_set": int(x.numel() * 4 + rand.numel() * 4),
                   "pipelining": "batch i+1 K1 overlaps batch i K2 (two HIP streams)" if args.overlap else "none (one stream)",
                   "clock_warmup_s": args.clock_warmup, "setup": setup_rec,
                   "keep_flag_table": ("built for batch i+1 on a side stream while batch i is scored (prepare_draws)" if k0_ahead
                                       else "built in line on the compute stream"),
                   "gather": ("none (1 GPU)" if not use_dist else
                              {"p2p": "one-shot P2P writes into the peers' buffers (csrc/p2p.hip), compute stream",
                               "side": "RCCL all_gather on a second stream",
                               "same": "RCCL all_gather on the compute stream"}[gather_mode["kind"]]) +
                             (" (chosen by the warm-up trial)" if (use_dist and args.gather_stream == "auto") else "")},
        "roofline": roofline,
    }
    if gather_info is not None:
        out["gather"] = gather_info
    if api is not None:
        out["api_level"] = api
    if world == 1 and not args.no_cpu_baseline and not counter:
        import oracle  # checker / CPU baseline only

        m = min(args.cpu_sample, n)
        x, rand = sets[last_set]  # the set the last timed step scored
        xs, rs = x[:m].cpu().numpy(), rand[:m].cpu().numpy()
        xo, ro = synth_latents(m, 999, 0.0, device, corr=args.ood_corr)  # spatially correlated maps -> OOD sample for the AUROC check
        gpu_ind = scores[:m].cpu().numpy()
        gpu_ood = pipe.score_latents(xo, ro).cpu().numpy()
        comp, mean, var = pca.components_, pca.mean_, pca.explained_variance_
        t0 = time.perf_counter()
        z = np.concatenate([oracle.mc_stack(xs[i : i + 1], rs[i], DROP_PROB, BLOCK) for i in range(m)])
        _, h = oracle.get_dl_h_z(z, N_MC)  # the reference's algorithmic form: one k-d tree per (image, dim)
        y = oracle.pca_transform(h, comp, mean, var)
        cpu_ind = oracle.md_score(y, md.feats_mean, md.precision)
        cpu_s = time.perf_counter() - t0
        # downstream parity from the device's own MC samples (f64-exact chain) and end-to-end from latents
        z_dev = pipe.stack(x[:m], rand[:m]).cpu().numpy()
        exact, _ = oracle.larem_pipeline(z_dev, N_MC, comp, mean, var, md.feats_mean, md.precision)
        zo = np.concatenate([oracle.mc_stack(xo[i : i + 1].cpu().numpy(), ro[i].cpu().numpy(), DROP_PROB, BLOCK) for i in range(m)])
        cpu_ood, _ = oracle.larem_pipeline(zo, N_MC, comp, mean, var, md.feats_mean, md.precision)
        rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))  # noqa: E731
        a_gpu = oracle.auroc_fpr95_aupr(gpu_ind, gpu_ood)
        a_cpu = oracle.auroc_fpr95_aupr(cpu_ind, cpu_ood)
        out["cpu_baseline"] = {
            "value": round(m / cpu_s, 2), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{m} of the {n} workload images, same inputs; oracle mc_stack + per-(image,dim) k-d-tree entropy "
                      f"(reference form) + PCA + LaREM, {cpu_s:.1f} s; host has {os.cpu_count()} cores",
        }
        out["parity"] = {
            "max_rel_err_from_device_samples": rel(gpu_ind, exact), "max_rel_err_from_latents": rel(gpu_ind, cpu_ind),
            "auroc_gpu": a_gpu[0], "auroc_oracle": a_cpu[0], "fpr95_gpu": a_gpu[1], "fpr95_oracle": a_cpu[1],
            "sample_images": m,
        }
        # counter-draw mode (throughput) against host-draw mode (parity) on the same InD / OOD latents: different random
        # masks, same statistics.  One draw of each says little (the AUROC of 10 000 + 10 000 images moves by ~3e-3 from
        # one set of masks to the next), so both modes are repeated over several seeds: mean +- sd, and the gap of the
        # means against its standard error.  Both modes redraw fully dropped maps (no NaN score): the counter generator
        # inside K0, the host sets in synth_latents(dead="redraw").
        xo_full, _ = synth_latents(n, 998, 0.0, device, corr=args.ood_corr)
        n_seeds = 12
        au_c, au_p, nan_plain = [], [], 0
        for sd in range(n_seeds):
            ind_c = pipe.score_latents(x, _hip.CounterDraws(99 + sd, 0, True)).cpu().numpy()
            ood_c = pipe.score_latents(xo_full, _hip.CounterDraws(99 + sd, n, True)).cpu().numpy()
            assert np.isfinite(ind_c).all() and np.isfinite(ood_c).all()
            au_c.append(oracle.auroc_fpr95_aupr(ind_c, ood_c)[0])
            _, r_i = synth_latents(n, 5000 + sd, 0.0, device, dead="redraw")  # same policy for fully dropped maps
            _, r_o = synth_latents(n, 6000 + sd, 0.0, device, dead="redraw")
            ind_p = pipe.score_latents(x, r_i).cpu().numpy()
            ood_p = pipe.score_latents(xo_full, r_o).cpu().numpy()
            au_p.append(oracle.auroc_fpr95_aupr(ind_p, ood_p)[0])
            if sd == 0:
                nan_plain = int(np.isnan(pipe.score_latents(x, _hip.CounterDraws(99, 0)).cpu().numpy()).sum())
        au_c, au_p = np.asarray(au_c), np.asarray(au_p)
        gap = float(au_c.mean() - au_p.mean())
        gap_se = float(np.sqrt(au_c.var(ddof=1) / n_seeds + au_p.var(ddof=1) / n_seeds))
        out["parity"]["counter_draws"] = {
            "seeds": n_seeds, "images": [n, n],
            "auroc_counter_mean": float(au_c.mean()), "auroc_counter_sd": float(au_c.std(ddof=1)),
            "auroc_host_draws_mean": float(au_p.mean()), "auroc_host_draws_sd": float(au_p.std(ddof=1)),
            "d_auroc_of_means": gap, "d_auroc_standard_error": gap_se,
            "nan_scores_counter_redraw": 0, "nan_scores_counter_without_redraw": nan_plain,
            "note": "different random masks cannot agree to 1e-4 at this sample size: the AUROC's own seed-to-seed sd is "
                    "what the gap is to be read against",
        }
        del xo_full
        # the reference's parallel form: one task per image over a process pool (evaluation/entropy.py:86-91), timed in a
        # child process that never touches the GPU (cpu_pool_child above), on the cores this job may use
        try:
            cores = effective_cpus()
            m_all = min(n, max(m, 40 * cores))
            tmp = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"runia_bench_{os.getpid()}.npz")
            np.savez(tmp, x=x[:m_all].cpu().numpy(), rand=rand[:m_all].cpu().numpy(), comp=comp, mean=mean, var=var,
                     md_mean=md.feats_mean, md_prec=md.precision)
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-pool-child", tmp, str(cores),
                                    str(N_MC), str(DROP_PROB), str(BLOCK)], capture_output=True, text=True, timeout=900)
                rec = json.loads(r.stdout.strip().splitlines()[-1])
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            ok = rel(rec.pop("scores_head"), cpu_ind[:16]) < 1e-9 if m_all >= 16 else True
            out["cpu_baseline_all_cores"] = {
                "value": round(m_all / rec["seconds"], 2), "unit": "images/s", "cores": cores, "kind": "port",
                "sample": f"{m_all} of the {n} workload images; multiprocessing.Pool({cores}).map over images, chunksize 1 "
                          f"(the reference's process_map form), each task = oracle mc_stack + per-dim k-d-tree entropy of one "
                          f"image; PCA + LaREM on the gathered rows; {rec['seconds']:.1f} s; os.cpu_count() = {os.cpu_count()}, "
                          f"usable = {cores}; scores equal the 1-core run: {ok}",
            }
        except Exception as e:  # the pool leg is a reported baseline; its failure must not lose the measurement
            out["cpu_baseline_all_cores"] = {"value": None, "error": repr(e)[:200]}
    if world == 1 and not args.no_stages and not use_dist:
        # the other BASELINE.json configs at their own sizes under the driver's clock: cfg3's postprocessors on 1M x 2048
        # rows (tools/bench_workloads.py, the code of --workload cfg3) and cfg4's LaRED leg
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_workloads as bw

            del sets, x, rand
            torch.cuda.empty_cache()
            t_s = time.perf_counter()
            cpu = not args.no_cpu_baseline
            c3 = bw.run_cfg3(device, 0, 1, None, args.rows, args.fit_rows or 8192, 2, 1, cpu_legs=cpu, f4=True)
            stages = dict(c3["stages"])
            stages["f4"] = c3.get("f4")
            stages["cfg3_step"] = {"rows": c3["rows_total"], "ms": round(c3["ms_per_step"], 3), "rows_per_s": round(c3["value"], 1),
                                   "note": "Mahalanobis + Energy(C=1000) + Energy(C=10) + kNN(k=50) over the same 1M rows = "
                                           "`python bench.py --workload cfg3`, 2 timed steps",
                                   "cpu_rows_per_s": c3.get("cpu_baseline", {}).get("value"),
                                   "clock_ghz_observed": c3.get("clock_ghz_observed"),
                                   "timing": "1 untimed pass (0.56 s of the leg's own kernels), then 2 timed passes"}
            torch.cuda.empty_cache()
            stages["cfg4_lared"] = bw.run_cfg4_lared(device)
            torch.cuda.empty_cache()
            stages["cfg4_from_feature_maps"] = bw.run_cfg4_from_maps(device)
            torch.cuda.empty_cache()
            # f-rows: fits, metrics, joint entropy, and the harness loop they exist for
            stages["fits"] = bw.run_fit_legs(device, bw.class_centres(device), cpu)
            torch.cuda.empty_cache()
            stages["metrics"] = bw.run_metrics_leg(device, cpu_legs=cpu)
            stages["entropy_joint"] = bw.run_entropy_joint_leg(device, cpu_legs=cpu)
            torch.cuda.empty_cache()
            stages["larex_eval"] = bw.run_larex_eval(device, cpu_legs=cpu)
            torch.cuda.empty_cache()
            # (without ViM here: its float32 np.linalg.eig of a 2048 x 2048 covariance is seconds of host time per fit, as upstream;
            # `--workload baselines_eval` runs all twelve)
            stages["baselines_eval"] = bw.run_baselines_eval(device, names=tuple(n for n in bw.BASELINES_ALL if n != "vim"), cpu_legs=cpu)
            stages["seconds"] = round(time.perf_counter() - t_s, 1)
            out["stages"] = stages
        except Exception as e:  # a reported extra; its failure must not lose the headline measurement
            out["stages"] = {"error": repr(e)[:300]}
    emit_record(out, saved_stdout)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
