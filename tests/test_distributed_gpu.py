"""Two and four ranks, real kernels: the N > 1 path of SURVEY 8(e) on the one GPU a test box has (its process guard allows six
processes on the card: four ranks + the test runner).

RCCL refuses two ranks on one device, so the processes share ``cuda:0`` over gloo (score shards are staged through
the host by ``gather_scores``); everything else - fitting on rank 0, ``broadcast_fitted`` with the arrays sent as
tensors, ``shard_bounds``, the HIP kernels on each rank's block, the single gather per postprocessor - is the code an
8-GPU RCCL job runs.  Sharded scores must equal the unsharded bits (rows are independent and every kernel scores a row
the same wherever it sits in a launch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_MC, C, HW, N_PCA = 16, 64, 4, 16
SIZES = (1001, 9, 3, 1)  # uneven blocks, 9 rows (four ranks: 3 + 3 + 3 + 0), fewer rows than ranks: empty tail shards


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.relu(torch.randn(n, C, HW, HW, device="cuda", generator=g)).contiguous()
    rand = torch.rand(n, N_MC, HW, HW, device="cuda", generator=g)
    rand[:, :, 0, 0] = rand[:, :, 0, 0].clamp_min(0.2)  # keep part of every map (no 0/0 upstream)
    feats = torch.relu(torch.randn(n, 96, device="cuda", generator=g) + 0.3).contiguous()
    return x, rand.contiguous(), feats


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import runia_core_amd as rc
        from runia_core_amd.distributed import ShardedPostprocessor, broadcast_fitted, gather_scores, shard_bounds
        from runia_core_amd.inference import LaREMPipeline, MDLatentSpace
        from runia_core_amd.inference.postprocessors import KNN, Mahalanobis

        state = None
        if rank == 0:  # the fit happens on one rank only
            xtr, rtr, ftr = _inputs(700, 1)
            probe = LaREMPipeline(None, None, N_MC, 0.5, 2)
            h_train = probe.entropy(probe.stack(xtr, rtr)).cpu().numpy()
            np.random.seed(3)
            red, pca = rc.apply_pca_ds_split(h_train, N_PCA)
            md = MDLatentSpace()
            md.setup(red)
            labels = np.arange(700) % 5
            ftr_h = ftr.cpu().numpy() + labels[:, None].astype(np.float32) * 0.2
            maha = Mahalanobis(flip_sign=False, num_classes=5)
            maha.setup(ftr_h, train_labels=labels, valid_feats=ftr_h[:64])
            knn = KNN(flip_sign=True, k_neighbors=50)
            knn.setup(ftr_h, valid_feats=ftr_h[:64])
            state = {"md": md, "pca": pca, "maha": maha, "knn": knn}
        state = broadcast_fitted(state, min_tensor_bytes=1024)
        pipe = LaREMPipeline(state["md"], state["pca"], N_MC, 0.5, 2)
        out = {}
        for n in SIZES:
            x, rand, feats = _inputs(n, 100 + n)  # every rank builds the same rows; only its block is scored
            a, b = shard_bounds(n, world, rank)
            larem = gather_scores(pipe.score_latents(x[a:b], rand[a:b]), n)
            sp_m, sp_k = ShardedPostprocessor(state["maha"]), ShardedPostprocessor(state["knn"])
            m_dev = sp_m.postprocess_device(feats)
            k_shard = sp_k.postprocess_shard(feats[a:b], n)
            k_host = sp_k.postprocess(feats.cpu().numpy())
            assert larem.is_cuda and m_dev.is_cuda and k_shard.is_cuda and larem.shape == m_dev.shape == k_shard.shape == (n,)
            out[f"larem_{n}"], out[f"maha_{n}"] = larem.cpu().numpy(), m_dev.cpu().numpy()
            out[f"knn_{n}"], out[f"knn_host_{n}"] = k_shard.cpu().numpy(), k_host
            if rank == 0:  # the unsharded scores of the same rows
                out[f"larem_full_{n}"] = pipe.score_latents(x, rand).cpu().numpy()
                out[f"maha_full_{n}"] = state["maha"].postprocess_device(feats).cpu().numpy()
                out[f"knn_full_{n}"] = state["knn"].postprocess_device(feats).cpu().numpy()
        np.savez(os.path.join(out_dir, f"g{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_real_kernels_equal_unsharded_bits(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    gs = [np.load(tmp_path / f"g{r}.npz") for r in range(world)]
    g0 = gs[0]
    for n in SIZES:
        for name, dt in (("larem", np.float64), ("maha", np.float64), ("knn", np.float32)):
            full = g0[f"{name}_full_{n}"]
            assert full.shape == (n,) and full.dtype == dt and np.isfinite(full).all()
            for g in gs:
                got = g[f"{name}_{n}"]
                assert got.dtype == dt and np.array_equal(got, full), (name, n, world)
        for g in gs:
            assert g[f"knn_host_{n}"].dtype == np.float32 and np.array_equal(g[f"knn_host_{n}"], g0[f"knn_full_{n}"])


def _p2p_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from runia_core_amd.distributed import OneShotGather, gather_scores, shard_bounds

        ok = True
        OneShotGather.set_debug(True)  # every launch asserts that the slot it overwrites was copied out two steps ago
        g64 = OneShotGather(5000, torch.float64, timeout_ms=5000)
        g32 = OneShotGather(300_000, torch.float32, timeout_ms=5000)
        kept = []
        for step, n in enumerate((5_000 * world, 9_999, 7, 2, 5_000 * world, 5_000 * world, 3, 1)):
            a, b = shard_bounds(n, world, rank)
            full = torch.arange(n, dtype=torch.float64, device="cuda") * 0.5 + 1000.0 * step
            got = g64(full[a:b].clone(), n)
            exp = gather_scores(full[a:b].clone(), n)
            ok = ok and bool(torch.equal(got, full)) and bool(torch.equal(exp, full))
            kept.append((got, full))
            if len(kept) >= 2:  # a gathered vector stays valid until the call after the next one
                prev_got, prev_full = kept[-2]
                ok = ok and bool(torch.equal(prev_got, prev_full))
        # several copy blocks per peer; an odd shard (4-byte granularity); cfg3's shard of an 8-GPU job (125 000 rows per rank)
        # the slot-reuse assertion goes off, comes back in the middle of the live buffer's steps on the even ranks only, then on
        # every rank: the acknowledgements are always recorded, so none of this may raise a false alarm
        OneShotGather.set_debug(False)
        for step, n in enumerate((600_000 // 2 * world, 599_999, 1, 125_000 * world, 9)):
            if step == 3:
                OneShotGather.set_debug(rank % 2 == 0)
            a, b = shard_bounds(n, world, rank)
            full = (torch.arange(n, dtype=torch.float32, device="cuda") % 4093) + step
            got = g32(full[a:b].clone(), n)
            ok = ok and bool(torch.equal(got, full))
        # back-to-back calls without any host synchronisation in between
        OneShotGather.set_debug(True)
        n = 5_000 * world
        a, b = shard_bounds(n, world, rank)
        outs = []
        for i in range(40):
            full = torch.full((n,), float(i), dtype=torch.float64, device="cuda")
            outs.append((g64(full[a:b], n).clone(), float(i)))
        torch.cuda.synchronize()
        ok = ok and all(bool((o == v).all()) for o, v in outs)
        g64.check()
        g32.check()
        try:
            g64(torch.zeros(6000, dtype=torch.float64, device="cuda"), 6_000 * world)
            refused = False
        except ValueError:
            refused = True
        g64.close()
        g32.close()
        np.savez(os.path.join(out_dir, f"p{rank}.npz"), ok=ok, refused=refused)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_one_shot_p2p_gather_ranks(tmp_path, world):
    """`OneShotGather` (csrc/p2p.hip: shards written straight into the peers' IPC-mapped buffers + flags, then a wait-and-copy
    launch) between two and four processes sharing the GPU: equal to `gather_scores`, even / uneven / tiny / empty / multi-block
    shards (9 rows and 1 row over four ranks; cfg3's 125 000-row shards), f64 and f32, 40 calls in flight without a host
    synchronisation, status clean - including the slot-reuse assertion of runia_p2p_debug, switched on for every launch -,
    capacity overflow refused."""
    mp.spawn(_p2p_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        g = np.load(tmp_path / f"p{r}.npz")
        assert bool(g["ok"]) and bool(g["refused"])
