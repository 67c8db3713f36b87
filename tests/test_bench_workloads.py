"""bench.py's cfg3 synthetic set is generated in fixed blocks keyed by the block index: the rows a rank builds for its shard
must be the rows of the whole set, whatever the world size (CPU generator here; the same code runs on the GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bench_workloads as bw  # noqa: E402
from runia_core_amd.distributed import shard_bounds  # noqa: E402


def test_cfg3_rows_do_not_depend_on_the_sharding():
    dev = torch.device("cpu")
    centres = bw.class_centres(dev)
    assert centres.shape == (bw.N_CLASSES, bw.D_FEAT)
    n = bw.GEN_BLOCK + 777  # not a whole number of blocks
    full, lab = bw.feature_rows(0, n, 0, dev, centres)
    assert full.shape == (n, bw.D_FEAT) and full.dtype == torch.float32 and bool((full >= 0).all())
    assert lab.shape == (n,) and int(lab.min()) >= 0 and int(lab.max()) < bw.N_CLASSES
    for world in (2, 3):
        parts = []
        for rank in range(world):
            a, b = shard_bounds(n, world, rank)
            f, l = bw.feature_rows(a, b, 0, dev, centres)
            assert torch.equal(l, lab[a:b])
            parts.append(f)
        assert torch.equal(torch.cat(parts), full)
    # the training split is a different stream of the same law
    train, _ = bw.feature_rows(0, 1000, 1, dev, centres)
    assert not torch.equal(train, full[:1000])
    w, b = bw.linear_head(dev)
    assert w.shape == (bw.N_LOGITS, bw.D_FEAT) and b.shape == (bw.N_LOGITS,)
