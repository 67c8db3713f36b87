"""world_size-2 gloo checks of the N>1 path (sharding + the single all_gather), CPU only.
The per-shard scorer here is a trivial torch function: the kernels themselves need a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from runia_core_amd.distributed import ShardedPostprocessor, broadcast_fitted, gather_scores, shard_bounds, sharded_scores


def test_shard_bounds_cover_rows_exactly():
    for n in (0, 1, 2, 7, 8, 9, 10000, 10001):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a0, b0), (a1, b1) in zip(blocks, blocks[1:]):
                assert b0 == a1 and a0 <= b0
            per = -(-n // world) if n else 0
            assert all(b - a <= per for a, b in blocks)
    # the sizes of the driver's runs: cfg3's 1 M rows over 8 GPUs (125 000 each), cfg2's 8 x 10 000, and 9 rows over 8
    # ranks (blocks of 2: four full, one of a single row, three empty tails)
    assert [shard_bounds(1_000_000, 8, r) for r in range(8)] == [(125_000 * r, 125_000 * (r + 1)) for r in range(8)]
    assert [b - a for a, b in (shard_bounds(1_000_001, 8, r) for r in range(8))] == [125_001] * 7 + [124_994]
    assert [shard_bounds(9, 8, r) for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 9), (9, 9), (9, 9), (9, 9)]
    assert [shard_bounds(80_000, 8, r)[1] - shard_bounds(80_000, 8, r)[0] for r in range(8)] == [10_000] * 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _RowSum:
    """Stands in for a set-up postprocessor on CPU."""

    threshold = 1.5

    def postprocess(self, x, **kwargs):
        s = np.asarray(x, dtype=np.float64).sum(axis=1)
        if "pred_labels" in kwargs:
            s = s + kwargs["pred_labels"]
        return s


class _RowSum32(_RowSum):
    """A scorer that returns float32, as energy / msp / knn / cMD / gen / GMM / ddu do; handles zero rows."""

    def postprocess(self, x, **kwargs):
        return np.asarray(x, dtype=np.float32).sum(axis=1, dtype=np.float32)


def _worker(rank, world, port, n_rows, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        rows = rng.standard_normal((n_rows, 6))
        labels = np.arange(n_rows, dtype=np.float64)
        full = sharded_scores(lambda r: torch.from_numpy(np.asarray(r).sum(axis=1)), rows)
        sp = ShardedPostprocessor(_RowSum())
        full2 = sp.postprocess(rows, pred_labels=labels)
        state = broadcast_fitted({"mean": rows.mean(0)} if rank == 0 else None)
        a, b = shard_bounds(n_rows, world, rank)
        local = torch.arange(a, b, dtype=torch.float32)
        g32 = gather_scores(local, n_rows)
        full32 = ShardedPostprocessor(_RowSum32()).postprocess(rows.astype(np.float32))  # empty tail shard keeps f32
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), full=full.numpy(), full2=full2, mean=state["mean"],
                 g32=g32.numpy(), thr=sp.threshold, full32=full32)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 7), (2, 8), (2, 1), (2, 0), (4, 9), (8, 9), (8, 3)])
def test_sharded_scoring_gloo(tmp_path, world, n_rows):
    """world 2, 4 and 8 (the driver's 8-GPU shape: 9 rows = four blocks of two, one single row, three empty tails)."""
    mp.spawn(_worker, args=(world, _free_port(), n_rows, str(tmp_path)), nprocs=world, join=True)
    rows = np.random.default_rng(5).standard_normal((n_rows, 6))
    exp = rows.sum(axis=1)
    for r in range(world):
        g = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(g["full"], exp)
        assert np.array_equal(g["full2"], exp + np.arange(n_rows))
        assert np.array_equal(g["g32"], np.arange(n_rows, dtype=np.float32))
        assert g["full32"].dtype == np.float32 and g["full32"].shape == (n_rows,)
        assert np.array_equal(g["full32"], rows.astype(np.float32).sum(axis=1, dtype=np.float32))
        if n_rows:
            assert np.array_equal(g["mean"], rows.mean(0))
        assert float(g["thr"]) == 1.5


class _RowSumDevice(_RowSum32):
    """`postprocess_device` stand-in (tensor rows in, tensor scores out) for the device-resident sharding entry points."""

    def postprocess_device(self, x):
        return x.to(torch.float32).sum(dim=1)


def _worker_state(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sklearn.decomposition import PCA

        from runia_core_amd.inference.postprocessors import FlatL2Bank, KNNLatentSpace, MDLatentSpace

        state = None
        if rank == 0:
            rng = np.random.default_rng(11)
            md = MDLatentSpace()
            md.feats_mean = rng.standard_normal((1, 192))
            md.precision = rng.standard_normal((192, 192))           # 295 KB: travels as a tensor
            md._setup_flag = True
            md._dev = {"not": "picklable on purpose", "fn": lambda: 0}  # device caches must stay behind
            knn = KNNLatentSpace()
            knn.activation_log = rng.standard_normal((300, 64)).astype(np.float32)  # 77 KB
            knn.index = FlatL2Bank(64)
            knn.index.add(knn.activation_log)
            knn.index._dev = object()
            knn._setup_flag = True
            pca = PCA(n_components=8).fit(rng.standard_normal((400, 128)))
            assert pca.components_.nbytes < 65536
            pca.components_ = np.asfortranarray(rng.standard_normal((96, 128)))  # sklearn's layout; 98 KB: out of band
            shared = rng.standard_normal((128, 128))                   # one array referenced twice stays one array
            state = {"md": md, "knn": knn, "pca": pca, "twice": [shared, shared], "small": np.arange(5),
                     "tensor": torch.arange(40000, dtype=torch.float64).reshape(200, 200), "text": "ok"}
        got = broadcast_fitted(state)
        assert got["md"]._dev is None and got["knn"].index._dev is None if rank else True
        assert got["twice"][0] is got["twice"][1]
        # device-resident sharding entry points on a stand-in scorer
        rows = torch.from_numpy(np.random.default_rng(5).standard_normal((9, 6)))
        sp = ShardedPostprocessor(_RowSumDevice())
        full_dev = sp.postprocess_device(rows)
        a, b = shard_bounds(9, world, rank)
        full_shard = sp.postprocess_shard(rows[a:b], 9)
        try:
            sp.postprocess_shard(rows[:1], 9)  # neither rank owns exactly one row: refused before any collective
            wrong = False
        except ValueError:
            wrong = True
        np.savez(os.path.join(out_dir, f"s{rank}.npz"), precision=got["md"].precision, mean=got["md"].feats_mean,
                 bank=got["knn"].index._host, ntotal=got["knn"].index.ntotal, comp=got["pca"].components_,
                 pca_mean=got["pca"].mean_, twice=got["twice"][0], small=got["small"], tensor=got["tensor"].numpy(),
                 full_dev=full_dev.numpy(), full_shard=full_shard.numpy(), wrong=wrong,
                 comp_f_order=got["pca"].components_.flags.f_contiguous and not got["pca"].components_.flags.c_contiguous,
                 flag=got["md"]._setup_flag and got["text"] == "ok")
    finally:
        dist.destroy_process_group()


def test_broadcast_fitted_lifts_arrays_and_device_sharding_world2_gloo(tmp_path):
    """`broadcast_fitted` pickles the object graph without its large arrays (they travel through dist.broadcast as
    tensors), drops device caches, keeps shared references shared; `postprocess_device` / `postprocess_shard` gather the
    same vector as the host form."""
    world = 2
    mp.spawn(_worker_state, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "s0.npz"), np.load(tmp_path / "s1.npz")
    for key in ("precision", "mean", "bank", "comp", "pca_mean", "twice", "small", "tensor", "full_dev", "full_shard"):
        assert g0[key].dtype == g1[key].dtype and np.array_equal(g0[key], g1[key]), key
    assert bool(g0["comp_f_order"]) and bool(g1["comp_f_order"])  # the memory layout travels with the array
    assert int(g1["ntotal"]) == 300 and bool(g1["flag"]) and bool(g0["wrong"]) and bool(g1["wrong"])
    rows = np.random.default_rng(5).standard_normal((9, 6))
    exp = rows.astype(np.float32).sum(axis=1, dtype=np.float32)
    assert g1["full_dev"].dtype == np.float32 and np.allclose(g1["full_dev"], exp, atol=1e-6)
    assert np.array_equal(g1["full_dev"], g1["full_shard"])


def test_array_lifting_pickler_moves_large_arrays_out_of_band():
    import io

    from runia_core_amd.distributed import _ArrayLiftingPickler, _ArrayPlacingUnpickler

    big = np.random.default_rng(0).standard_normal((256, 256))
    obj = {"a": big, "b": [big, np.arange(3)], "s": "x", "c": np.array(["u", "v"], dtype=object)}
    buf = io.BytesIO()
    p = _ArrayLiftingPickler(buf, 1 << 16)
    p.dump(obj)
    assert len(p.arrays) == 1 and p.meta == [("nd", (256, 256), "<f8")]
    assert len(buf.getvalue()) < 4096  # the 512 KB array is not in the byte stream
    back = _ArrayPlacingUnpickler(io.BytesIO(buf.getvalue()), [p.arrays[0].numpy()]).load()
    assert back["a"] is back["b"][0] and np.array_equal(back["a"], big) and back["c"][1] == "v"


def _worker_big_state(rank, world, port, out_dir):
    import resource

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from runia_core_amd.inference.postprocessors import FlatL2Bank

        m, d = 50_000, 2048   # cfg3: kNN bank 50 000 x 2048 f32 = 410 MB, Mahalanobis precision 2048 x 2048 f64 = 33.5 MB
        state = None
        if rank == 0:
            bank = FlatL2Bank(d)
            rows = np.empty((m, d), dtype=np.float32)
            rows[:] = np.arange(d, dtype=np.float32)[None, :] * 1e-3
            rows[:, 0] = np.arange(m, dtype=np.float32)
            bank.add(rows)
            del rows
            prec = np.zeros((d, d), dtype=np.float64)
            prec[np.arange(d), np.arange(d)] = np.arange(1, d + 1, dtype=np.float64)
            state = {"bank": bank, "precision": prec, "class_mean": np.ones((10, d), dtype=np.float32)}
        state_bytes = m * d * 4 + d * d * 8
        dist.barrier()
        before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
        got = broadcast_fitted(state)
        after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
        host = got["bank"]._host
        ok = (host.shape == (m, d) and host.dtype == np.float32 and float(host[12345, 0]) == 12345.0 and host[777, 2047] == np.float32(2047) * np.float32(1e-3)
              and got["bank"].ntotal == m and float(got["precision"][2047, 2047]) == 2048.0 and got["class_mean"].shape == (10, d))
        np.savez(os.path.join(out_dir, f"b{rank}.npz"), grew=after - before, state_bytes=state_bytes, ok=ok)
    finally:
        dist.destroy_process_group()


def test_broadcast_fitted_of_a_cfg3_sized_state_makes_no_second_copy(tmp_path):
    """broadcast_fitted on the fitted state of BASELINE.json configs[2] (410 MB kNN bank + 33.5 MB precision) under gloo, world 2:
    the arrays travel out of band, so the sender's peak memory does not grow by another copy of the state and the receiver's
    grows by about one copy (not by pickle stream + unpickled arrays)."""
    world = 2
    mp.spawn(_worker_big_state, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "b0.npz"), np.load(tmp_path / "b1.npz")
    assert bool(g0["ok"]) and bool(g1["ok"])
    size = int(g0["state_bytes"])
    assert int(g0["grew"]) < 0.5 * size, (int(g0["grew"]), size)    # rank 0: no pickled copy of the 444 MB
    assert int(g1["grew"]) < 1.6 * size, (int(g1["grew"]), size)    # receiver: the arrays once (+ allocator slack)
