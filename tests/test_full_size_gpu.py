"""BASELINE.json configs at FULL size on the GPU, checked through size-independent properties
(the oracle cannot finish these sizes in seconds; bounded slices are compared against it)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from runia_core_amd import _hip

    _hip.require_gpu()
    return _hip


def test_cfg3_energy_msp_1m_rows(hip):
    """1M x 1000 logits (4 GB): shift invariance, permutation invariance, slice vs oracle."""
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(1_000_000, 1000, device="cuda", generator=g) * 2
    lse, msp = hip.row_lse_msp(x, True, True)
    assert lse.shape == (1_000_000,) and bool(torch.isfinite(lse).all()) and bool(((msp > 0) & (msp <= 1)).all())
    # logsumexp(x + c) = logsumexp(x) + c ; softmax is shift invariant  (c = 8 is exact in f32 here)
    x += 8.0
    lse2, msp2 = hip.row_lse_msp(x, True, True)
    assert float((lse2 - lse - 8.0).abs().max()) < 2e-5
    assert float((msp2 - msp).abs().max()) < 1e-6
    x -= 8.0
    perm = torch.randperm(1000, device="cuda", generator=g)
    lse3, msp3 = hip.row_lse_msp(x[:100_000][:, perm].contiguous(), True, True)
    assert float((lse3 - lse[:100_000]).abs().max()) < 5e-6 and float((msp3 - msp[:100_000]).abs().max()) < 1e-6
    sl = slice(500_000, 500_512)
    xs = x[sl].cpu().numpy()
    assert rel_err(lse[sl].cpu().numpy(), oracle.energy_score(xs)) < 1e-5
    assert rel_err(msp[sl].cpu().numpy(), oracle.msp_score(xs)) < 1e-5


def _cfg3_features(n, d, c, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    centres = torch.randn(c, d, device="cuda", generator=g) * 0.5
    lab = torch.randint(0, c, (n,), device="cuda", generator=g)
    f = torch.relu(centres[lab] + torch.randn(n, d, device="cuda", generator=g))
    return f, lab


def test_cfg3_mahalanobis_1m_rows(hip):
    """1M x 2048 f32 features, 10 classes: row independence, class-permutation invariance, slice vs oracle."""
    n, d, c = 1_000_000, 2048, 10
    f, lab = _cfg3_features(n, d, c, 2024)
    cm, prec = oracle.mahalanobis_setup(f[:4000].cpu().numpy(), lab[:4000].cpu().numpy(), c)
    packed = hip.pack_weights(torch.from_numpy(prec).cuda())
    mu_p = torch.from_numpy(cm.astype(np.float64) @ prec).cuda()
    s = hip.mahalanobis_score(f, torch.from_numpy(cm).cuda(), packed, mu_p)
    assert s.shape == (n,) and bool(torch.isfinite(s).all()) and bool((s <= 0).all())
    sl = slice(123_456, 123_456 + 3000)
    assert torch.equal(hip.mahalanobis_score(f[sl].contiguous(), torch.from_numpy(cm).cuda(), packed, mu_p), s[sl])
    p = np.random.default_rng(0).permutation(c)
    s_perm = hip.mahalanobis_score(f[sl].contiguous(), torch.from_numpy(cm[p]).cuda(), packed,
                                   torch.from_numpy(cm[p].astype(np.float64) @ prec).cuda())
    assert torch.equal(s_perm, s[sl])  # max over classes does not depend on their order
    fs = f[sl][:64].cpu().numpy()
    assert rel_err(s[sl][:64].cpu().numpy(), oracle.mahalanobis_score(fs, cm, prec, c)) < 1e-9


def test_cfg3_knn_bank_50000x2048(hip):
    """bank 50 000 x 2048, k = 50, 65 536 queries: known answers + row independence + slice vs oracle."""
    m, d, k, nq = 50_000, 2048, 50, 65_536
    g = torch.Generator(device="cuda").manual_seed(7)
    bank = hip.l2_normalize(torch.randn(m, d, device="cuda", generator=g))
    q = torch.randn(nq, d, device="cuda", generator=g)
    q[:m // 2] = bank[: m // 2] * 3.0  # scaled copies of bank rows: nearest neighbour at distance 0 after normalisation
    qn = hip.l2_normalize(q)
    s = hip.knn_kth(qn, bank, k)
    assert s.shape == (nq,) and s.dtype == torch.float32 and bool((s <= 0).all()) and bool((s >= -4.0001).all())
    s1 = hip.knn_kth(qn[:4096].contiguous(), bank, 1)
    assert float(s1.abs().max()) < 2e-6  # k = 1 finds the copied row
    assert torch.equal(hip.knn_kth(qn[30_000:31_000].contiguous(), bank, k), s[30_000:31_000])
    # k-th distance is monotone in k
    s60 = hip.knn_kth(qn[:2048].contiguous(), bank, 60)
    assert bool((s60 <= s[:2048]).all())
    idx = [0, 1, 40_000, 65_535]
    exp = oracle.knn_kth_score(bank.cpu().numpy(), q[idx].cpu().numpy(), k, chunk=1)
    assert rel_err(s[idx].cpu().numpy(), exp) < 1e-5
    # 768 rows spread over the batch (every chunk and tile position of the bf16 candidate kernel) against the oracle's
    # arithmetic at BLAS cost (oracle/harness.py::knn_kth_blas: float64 ranking, the 2k + 8 nearest re-measured in the
    # oracle's float32 form - pinned bit-equal to oracle.knn_kth_score by tests/test_oracle_goldens.py)
    from oracle.harness import knn_kth_blas
    rows = np.unique(np.concatenate([np.arange(0, nq, 97), np.arange(nq - 100, nq)]))[:768]
    exp = knn_kth_blas(bank.cpu().numpy(), q[rows].cpu().numpy(), k, chunk=256)
    got = s[rows].cpu().numpy()
    # two float32 sums of 2 048 squares in different orders (NumPy pairwise / the kernel's lanes): two ulps of the largest
    assert np.abs(got - exp).max() <= 2.4e-7 * np.abs(exp).max()


def test_cfg4_per_proposal_entropy_and_larem(hip):
    """cfg4 shape: 100 proposals x 16 MC x 1024-d per image (here 300 images = 30 000 proposals) -> PCA-64 -> LaREM."""
    n_prop, n_mc, d, n = 30_000, 16, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(4)
    base = torch.randn(n_prop, 1, d, device="cuda", generator=g) + 2
    z = (base * (1 + 0.1 * torch.randn(n_prop, n_mc, d, device="cuda", generator=g))).reshape(n_prop * n_mc, d).contiguous()
    h = hip.kl_entropy_per_dim(z, n_mc, 5)
    assert h.shape == (n_prop, d) and bool(torch.isfinite(h).all())
    rng = np.random.default_rng(1)
    comp = np.linalg.qr(rng.standard_normal((d, n)))[0].T
    mean, var = rng.standard_normal(d), rng.random(n) + 0.1
    a = rng.standard_normal((n, n))
    prec = a @ a.T / n + np.eye(n)
    md_mean = rng.standard_normal((1, n)) * 0.1
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()  # noqa: E731
    s = hip.pca_md_score(h, hip.pack_weights(dev(comp.T)), dev((mean.reshape(1, -1) @ comp.T).ravel()), dev(np.sqrt(var)),
                         dev(md_mean.ravel()), hip.pack_weights(dev(prec)), n)
    zs = z[: 48 * n_mc].cpu().numpy()
    exp, h_exp = oracle.larem_pipeline(zs, n_mc, comp, mean, var, md_mean, prec)
    assert np.abs(h[:48].cpu().numpy() - h_exp).max() < 1e-11
    assert rel_err(s[:48].cpu().numpy(), exp) < 1e-9
    # proposals are independent rows
    s2 = hip.pca_md_score(h[10_000:10_700].contiguous(), hip.pack_weights(dev(comp.T)), dev((mean.reshape(1, -1) @ comp.T).ravel()),
                          dev(np.sqrt(var)), dev(md_mean.ravel()), hip.pack_weights(dev(prec)), n)
    assert torch.equal(s2, s[10_000:10_700])


def test_row_gemm_stages_tile_height_does_not_change_bits(hip):
    """The row GEMM kernel (PCA transform, MD, ViM norm, KDE on the matrix cores) runs 16-row tiles below four tiles per
    CU and 32-row tiles above (40 000 rows here: 32-row tiles).  A row's result must not depend on that: slices scored on
    their own (16-row tiles) equal the same rows of the whole batch bit for bit, ragged ends included; sampled rows are
    checked against the oracle.  Round 4: the rows behind the whole rounds of 32-row tiles run as a second launch of
    smaller units (16-row tiles; KDE: 16 rows x 256 training rows + replay) - on 256 CUs with two resident workgroups
    each that boundary is row 32 768 here, and slices across it are compared too."""
    n_rows, d, n = 40_003, 512, 256
    g = torch.Generator(device="cuda").manual_seed(9)
    h = torch.randn(n_rows, d, dtype=torch.float64, device="cuda", generator=g)
    rng = np.random.default_rng(3)
    comp = np.linalg.qr(rng.standard_normal((d, n)))[0].T
    mean, var = rng.standard_normal(d), rng.random(n) + 0.1
    a = rng.standard_normal((n, n))
    prec = a @ a.T / n + np.eye(n)
    md_mean = rng.standard_normal((1, n)) * 0.1
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()  # noqa: E731
    pct, bias, scale = hip.pack_weights(dev(comp.T)), dev((mean.reshape(1, -1) @ comp.T).ravel()), dev(np.sqrt(var))
    pp, mdm = hip.pack_weights(dev(prec)), dev(md_mean.ravel())
    y = hip.pca_transform(h, pct, bias, scale, n)
    s = hip.md_score(y, mdm, pp)
    for lo, hi in ((0, 700), (17_001, 17_050), (32_750, 32_800), (39_990, n_rows)):
        y_part = hip.pca_transform(h[lo:hi].contiguous(), pct, bias, scale, n)
        assert torch.equal(y_part, y[lo:hi])
        assert torch.equal(hip.md_score(y_part, mdm, pp), s[lo:hi])
    idx = np.r_[0:24, 20_000:20_024, n_rows - 24:n_rows]
    y_exp = oracle.pca_transform(h[idx].cpu().numpy(), comp, mean, var)
    assert rel_err(y[idx].cpu().numpy(), y_exp) < 1e-11
    assert rel_err(s[idx].cpu().numpy(), oracle.md_score(y_exp, md_mean, prec)) < 1e-9
    # f32 rows (the reference's default feature dtype) and the LaRED contraction
    y32 = y.float()
    s32 = hip.md_score(y32, mdm.float(), pp)
    assert torch.equal(hip.md_score(y32[33_000:33_100].contiguous(), mdm.float(), pp), s32[33_000:33_100])
    tr = torch.randn(3_000, 64, dtype=torch.float64, device="cuda", generator=g)
    x = torch.randn(n_rows, 64, dtype=torch.float64, device="cuda", generator=g)
    st = hip.kde_pack_train(tr)
    kd = hip.kde_score_packed(st, x, 4.0)
    for lo, hi in ((5_000, 5_040), (32_750, 32_800), (n_rows - 40, n_rows)):
        assert torch.equal(hip.kde_score_packed(st, x[lo:hi].contiguous(), 4.0), kd[lo:hi])
    assert rel_err(kd[:64].cpu().numpy(), oracle.kde_score(tr.cpu().numpy(), x[:64].cpu().numpy(), 4.0)) < 1e-11


# ---------------- fused LaREM path (K0 + K1 + K2') at full size and at the launch boundaries -------------------------
def _larem_state(hip, seed=0, c=512, n=256):
    from runia_core_amd.dimensionality_reduction import DevicePCA
    from runia_core_amd.inference import LaREMPipeline, MDLatentSpace

    rng = np.random.default_rng(seed)
    comp = np.linalg.qr(rng.standard_normal((c, n)))[0].T
    pca_mean, var = rng.standard_normal(c), rng.random(n) + 0.05
    a = rng.standard_normal((n, n))
    md = MDLatentSpace()
    md.feats_mean, md.precision, md._setup_flag = rng.standard_normal((1, n)) * 0.1, a @ a.T / n + np.eye(n), True
    return LaREMPipeline(md, DevicePCA(comp, pca_mean, var, True), 16, 0.5, 2), (comp, pca_mean, var, md.feats_mean, md.precision)


def _latents(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.relu(torch.randn(n, 512, 4, 4, device="cuda", generator=g))
    rand = torch.rand(n, 16, 4, 4, device="cuda", generator=g)
    rand[:, :, 0, 0] = rand[:, :, 0, 0].clamp_min(0.2)  # no fully dropped map: every score finite
    return x, rand


@pytest.mark.parametrize("n", [10_000, 10_003, 65_535, 65_536, 70_001])
def test_fused_larem_from_latents_sizes_and_boundaries(hip, n):
    """score_latents at the bench size, at N % 8 != 0 (the XCD-aware workgroup order rounds the grid up to 8 images), at
    the C entry's 65 535-image limit, one beyond it and well beyond it (host chunk loop): sampled rows against the
    oracle, slices against the whole (rows are independent, so a shard scores the same bits as within the batch)."""
    pipe, (comp, pmean, var, mdm, prec) = _larem_state(hip)
    x, rand = _latents(n, 100 + n % 97)
    s = pipe.score_latents(x, rand)
    assert s.shape == (n,) and s.dtype == torch.float64 and bool(torch.isfinite(s).all())
    idx = np.r_[0:6, n // 2 - 3 : n // 2 + 3, n - 6 : n]
    if n > 65_540:
        idx = np.r_[idx, 65_530:65_540]  # both sides of the host chunk boundary
    idx = np.unique(idx)
    z = np.concatenate([oracle.mc_stack(x[i : i + 1].cpu().numpy(), rand[i].cpu().numpy(), 0.5, 2) for i in idx])
    exp, _ = oracle.larem_pipeline(z, 16, comp, pmean, var, mdm, prec)
    assert rel_err(s[idx].cpu().numpy(), exp) < 1e-9
    for a, b in ((0, 1000), (n // 2 - 123, n // 2 + 1001), (n - 777, n)):
        assert torch.equal(pipe.score_latents(x[a:b].contiguous(), rand[a:b].contiguous()), s[a:b]), (a, b)
    # counter-draw mode: chunks and shards line up through the image ids
    from runia_core_amd._hip import CounterDraws

    sc = pipe.score_latents(x, CounterDraws(5, 1_000_000))
    a, b = n - 3333, n - 1
    part = pipe.score_latents(x[a:b].contiguous(), CounterDraws(5, 1_000_000 + a))
    assert torch.equal(torch.nan_to_num(part, nan=1.0), torch.nan_to_num(sc[a:b], nan=1.0))


def test_sharded_postprocessor_real_kernels_nccl_world1(hip):
    """ShardedPostprocessor over a world-size-1 RCCL group with the real kernels behind it: f64 (LaREM, Mahalanobis) and
    f32 (Energy, kNN) scorers, uneven row counts, and the empty-input case (dtype kept)."""
    import socket

    import torch.distributed as dist

    from runia_core_amd.distributed import ShardedPostprocessor, shard_bounds
    from runia_core_amd.inference import Energy, KNNLatentSpace, MDLatentSpace

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(3)
        train = rng.standard_normal((500, 48))
        md = MDLatentSpace()
        md.setup(train)
        knn = KNNLatentSpace()
        knn.setup(train.astype(np.float32))
        en = Energy(flip_sign=False)
        en.setup(rng.standard_normal((100, 10)).astype(np.float32))
        for rows in (1, 7, 1001):
            xt = rng.standard_normal((rows, 48))
            assert np.array_equal(ShardedPostprocessor(md).postprocess(xt), md.postprocess(xt))
            g32 = ShardedPostprocessor(knn).postprocess(xt.astype(np.float32))
            assert g32.dtype == np.float32 and np.array_equal(g32, knn.postprocess(xt.astype(np.float32)))
            lg = rng.standard_normal((rows, 10)).astype(np.float32)
            assert np.array_equal(ShardedPostprocessor(en).postprocess(lg), en.postprocess(lg))
        # what a rank with an empty tail shard does (N = 9 on 8 ranks: rank 5.. get no rows): zero rows through the
        # real postprocessors keep their dtypes
        a, b = shard_bounds(9, 8, 7)
        assert a == b
        assert md.postprocess(np.zeros((0, 48))).dtype == np.float64
        assert knn.postprocess(np.zeros((0, 48), np.float32)).dtype == np.float32
        assert en.postprocess(np.zeros((0, 10), np.float32)).dtype == np.float32
        assert ShardedPostprocessor(en).postprocess(np.zeros((0, 10), np.float32)).shape == (0,)
    finally:
        dist.destroy_process_group()
