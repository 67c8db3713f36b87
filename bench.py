#!/usr/bin/env python3
"""Headline benchmark: OOD scores/sec on the LaREM-16MC / PCA-256 hot path (BASELINE.json configs[1]).

One "step" = one pass of the scoring hot path over the 10 000 test images of the workload,
inputs resident in HBM:  hooked latent maps (N,512,4,4) f32 + DropBlock draws (N,16,4,4)
  -> MC-dropout latent stacking -> per-dimension KL entropy -> PCA 512->256 (whitened)
  -> LaREM (Mahalanobis) score -> [N>1: one RCCL all_gather of the score shards].
Weak scaling: every rank scores its own 10 000-image shard (rows are independent).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel,
timed with HIP events on the launch stream inside the timed region) and `cpu_baseline`
(the CPU oracle timed on a bounded sample; reported baseline, never the thing measured).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MC, C, H, W, N_PCA = 16, 512, 4, 4, 256
DROP_PROB, BLOCK = 0.5, 2
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_latents(n, seed, shift, device, scale=1.0, corr=0.0):
    """cfg2-synth: X ~ ReLU(N(shift,1)) on (n,512,4,4) with a fixed per-channel scale; draws U(0,1) on (n,16,4,4).
    Draws whose block mask would drop the whole 4x4 map (sum(bm)=0 -> inf/NaN in the reference
    as well) are replaced by "no seed" so that every score is finite."""
    g0 = torch.Generator(device=device).manual_seed(77)  # per-channel scale: a property of the "layer", same for all sets
    chan = torch.rand(1, C, 1, 1, device=device, generator=g0) * 1.5 + 0.25
    g = torch.Generator(device=device).manual_seed(seed)
    noise = torch.randn(n, C, H, W, device=device, generator=g)
    if corr > 0.0:  # OOD variant: part of every map is spatially constant -> DropBlock perturbs the channel mean less
        shared = torch.randn(n, C, 1, 1, device=device, generator=g)
        noise = corr * shared + (1.0 - corr * corr) ** 0.5 * noise
    x = torch.relu(noise * (chan * scale) + shift).contiguous()
    rand = torch.rand(n, N_MC, H, W, device=device, generator=g)
    mask = (rand < DROP_PROB / BLOCK**2).float().reshape(n * N_MC, 1, H, W)
    bm = 1 - torch.nn.functional.max_pool2d(mask, BLOCK, 1, BLOCK // 2)[:, :, :-1, :-1]
    dead = bm.sum(dim=(1, 2, 3)) == 0
    rand.reshape(n * N_MC, H, W)[dead] = 1.0
    return x, rand.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--images", type=int, default=10000, help="test images per GPU (workload: 10 000)")
    ap.add_argument("--train-images", type=int, default=4096)
    ap.add_argument("--ood-corr", type=float, default=0.25, help="spatial correlation of the OOD sample's latent maps")
    ap.add_argument("--cpu-sample", type=int, default=1280, help="images timed on the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="two HIP streams: K1 of batch i+1 beside K2 of batch i (measured: no gain, 0.324 vs 0.329 ms)")
    ap.add_argument("--chunks", type=int, default=None, help="row blocks pipelined over two streams (default: pipeline's)")
    ap.add_argument("--clock-warmup", type=float, default=1.0,
                    help="seconds of the same step run untimed before the W warm-up steps: the GPU needs ~0.1 s of "
                         "sustained load to reach its working clocks (measured: 0.261 ms/step over the first 10 steps, "
                         "0.205 ms/step over 2000)")
    ap.add_argument("--event-every", type=int, default=8,
                    help="HIP events bracket the dominant kernel on every n-th timed step (a bracketed step runs the "
                         "table launch and the kernel as two C calls with two event records between them: ~40 us "
                         "slower than an unbracketed step, so bracketing every step would tax the metric by 16 %%)")
    ap.add_argument("--gather-stream", choices=["auto", "same", "side"], default="auto",
                    help="queue the all_gather on the compute stream, or on a second stream behind an event with "
                         "two output buffers in turn (the next step's kernels then start without waiting for it); "
                         "auto: time both during the warm-up and keep the faster (all ranks agree through a MAX)")
    args = ap.parse_args()

    # Exactly one line may reach stdout (the JSON record): RCCL prints its version banner to stdout when
    # NCCL_DEBUG=VERSION is set, so fd 1 points at stderr until the record is written.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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

    # ---------------- setup (untimed): fit PCA-256 + LaREM on in-distribution entropies ----------
    probe = LaREMPipeline(None, None, N_MC, DROP_PROB, BLOCK)
    xtr, rtr = synth_latents(args.train_images, 1234, 0.0, device)
    h_train = probe.entropy(probe.stack(xtr, rtr)).cpu().numpy()
    del xtr, rtr
    np.random.seed(1234)  # sklearn's randomized SVD draws from the global NumPy state
    red, pca = rc.apply_pca_ds_split(h_train, N_PCA)
    md = MDLatentSpace()
    md.setup(red)
    pipe = LaREMPipeline(md, pca, N_MC, DROP_PROB, BLOCK)
    fused = _hip.mc_entropy_supported(H, W, N_MC, pipe.k)

    n = args.images
    x, rand = synth_latents(n, 1235 + rank, 0.0, device)  # this rank's shard

    k1_events = []  # (start, end) HIP event pairs around every K1 launch of the timed region

    torch.cuda.synchronize()
    inputs_ready = torch.cuda.current_stream().record_event()  # x / rand are resident from here on

    gathered = [torch.empty(world * n, dtype=torch.float64, device=device) for _ in range(2)] if use_dist else None
    side_stream = torch.cuda.Stream() if (use_dist and args.gather_stream != "same") else None
    gather_mode = {"side": args.gather_stream == "side"}
    gather_turn = [0]

    def gather(s):
        """The single RCCL all_gather of the path: equal shards, preallocated output.  Default: queued
        on the compute stream.  --gather-stream side: queued on a second stream behind an event; outputs alternate
        between two buffers (a buffer is rewritten two steps later, in stream order on the same stream)."""
        out = gathered[gather_turn[0]]
        gather_turn[0] ^= 1
        if not gather_mode["side"]:
            dist.all_gather_into_tensor(out, s)
            return out
        side_stream.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(side_stream):
            dist.all_gather_into_tensor(out, s)
        s.record_stream(side_stream)
        return out

    def step(timed=False, index=0, collective=True):
        timed = timed and (index % max(1, args.event_every) == 0)
        if not args.overlap:
            s = pipe.score_latents(x, rand, chunks=args.chunks, k1_events=k1_events if timed else None)
            return gather(s) if (use_dist and collective) else s
        # streaming form: K1 of this batch overlaps K2 of the previous one (two HIP streams); the gather is
        # queued behind this batch's K2 on the same stream, nothing waits on the host until the final sync
        a = pipe.score_latents_async(x, rand, k1_events=k1_events if timed else None, inputs_ready=inputs_ready)
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
    for _ in range(args.warmup):
        step()
    if use_dist:
        assert step().shape == (world * n,)
    if use_dist and args.gather_stream == "auto" and not args.overlap:
        # untimed: which placement of the collective is faster on this node?  (one-GPU rehearsal: same stream
        # +9 us, side stream +23 us per step; with real peers the same-stream form also exposes the ring latency)
        trial = {False: float("inf"), True: float("inf")}
        for mode in (False, True, False, True, False, True):  # best of three interleaved trials per placement
            gather_mode["side"] = mode
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
            trial[mode] = min(trial[mode], float(t_m.item()))
        gather_mode["side"] = trial[True] < 0.98 * trial[False]  # the simpler placement unless clearly slower
        if rank == 0:
            print(f"gather placement trial: same {trial[False] / 30 * 1e3:.4f} ms/step, side {trial[True] / 30 * 1e3:.4f} ms/step",
                  file=sys.stderr)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        scores = step(True, i)
    torch.cuda.synchronize()  # every stream of the device, the gather stream included
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in k1_events])) if k1_events else float('nan')
    bracketed_steps = len(range(0, args.steps, max(1, args.event_every)))
    k1_launches_per_step = max(1, len(k1_events) // max(1, bracketed_steps))
    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * n * args.steps / elapsed

    # ---------------- roofline of the dominant kernel ------------------------------------------
    # algorithmic bytes of K1 per image ("with MC stacking from latent"): the latent map C*H*W*4, the
    # keep-flag table of the image n_mc*(H*W+2)*4 (what the small launch before it makes of the n_mc*H*W*4 draws)
    # and the C entropies written as f64.  HIP events bracket this kernel alone (pipeline.k1_events).
    kname, bytes_per_img = "mc_entropy_kernel", C * H * W * 4 + N_MC * (H * W + 2) * 4 + C * 8
    imgs_per_launch = n / k1_launches_per_step
    achieved = bytes_per_img * imgs_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            rec = json.load(open(pmc_file)).get(kname, {})
            # PMC bytes were collected at `images_per_launch` images per launch; scale to this run's launch size
            traffic = int(rec["hbm_bytes_per_launch"] * imgs_per_launch / rec.get("images_per_launch", 10000))
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "algorithmic_bytes_per_launch": int(bytes_per_img * imgs_per_launch), "avg_launch_ms": round(kernel_ms, 4),
        "launches_per_step": k1_launches_per_step, "launches_timed": len(k1_events),
    }

    # ---------------- parity on a bounded sample + CPU baseline (oracle = checker / baseline only) --
    out = {
        "metric": "OOD scores/sec, LaREM 16-MC PCA-256 (ResNet-18 layer4 latent 512x4x4)",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "CIFAR10 ResNet-18 LaREM: 16 MC samples, 512-d latent -> PCA-256, 10000 test images per GPU",
                   "images_per_gpu": n, "mc_samples": N_MC, "latent": [C, H, W], "pca_components": N_PCA,
                   "row_blocks_per_step": k1_launches_per_step, "input_dtype": "f32",
                   "pipelining": "batch i+1 K1 overlaps batch i K2 (two HIP streams)" if args.overlap else "none (one stream)",
                   "clock_warmup_s": args.clock_warmup,
                   "gather": ("none (1 GPU)" if not use_dist else
                              ("all_gather on a second stream" if gather_mode["side"] else "all_gather on the compute stream"))},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        import oracle  # checker / CPU baseline only

        m = min(args.cpu_sample, n)
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
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
