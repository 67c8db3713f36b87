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


@pytest.mark.parametrize("n_rows", [7, 8, 1, 0])
def test_sharded_scoring_world2_gloo(tmp_path, n_rows):
    world = 2
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
