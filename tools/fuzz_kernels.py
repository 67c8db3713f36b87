"""Randomised differential check of the stand-alone stages against the CPU oracle (one MI355X, a few minutes):
joint / per-dimension entropy, Energy / MSP, normaliser, PCA transform + MD at sizes around the tile switches, kNN
(k-th distance) and LaRED on both kernel forms.  Shapes are drawn around the places where a kernel changes its launch
shape or code path (vector widths, register-resident row limits, 16- / 32-row tiles, column halves, chunk limits).

    gpurun -- python tools/fuzz_kernels.py [--seed S] [--rounds R]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from runia_core_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=2024)
ap.add_argument("--rounds", type=int, default=40)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
bad = 0


def dev(x, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t if dt is None else t.to(dt)


def rel(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    both_nan = np.isnan(x) & np.isnan(y)
    d = np.abs(x - y) / np.maximum(1.0, np.abs(y))
    d[both_nan] = 0.0
    d[(x == y)] = 0.0  # equal infinities
    return float(np.nanmax(d)) if d.size else 0.0


def check(name, args, err, tol):
    global bad
    if not (err <= tol):
        bad += 1
        print("MISMATCH", name, args, err, flush=True)


for t in range(a.rounds):
    # ---- entropy: joint (register form / LDS form) and per dimension ----
    n_mc = int(rng.choice([2, 3, 5, 6, 8, 9, 12, 16, 17, 24, 32, 33, 40, 64]))
    d = int(rng.choice([1, 2, 4, 12, 64, 130, 256, 510, 512, 1024, 1028, 2048]))
    n_img = int(rng.integers(1, 6))
    z = (rng.standard_normal((n_img * n_mc, d)) * float(rng.choice([1e-3, 1.0, 30.0]))).astype(np.float32)
    if rng.random() < 0.3:  # constant channels: the min_dist clip
        z.reshape(n_img, n_mc, d)[:, :, :: max(1, d // 7)] = 0.25
    k = 5 if n_mc > 5 else n_mc - 1
    hj = _hip.kl_entropy_joint(dev(z), n_mc, k).cpu().numpy()
    check("joint entropy", (n_mc, d, n_img), rel(hj, oracle.kl_entropy_joint_vectorized(z, n_mc, k)[:, 0]), 1e-11)
    hd = _hip.kl_entropy_per_dim(dev(z), n_mc, k).cpu().numpy()
    check("entropy per dim", (n_mc, d, n_img), rel(hd, oracle.kl_entropy_per_dim_vectorized(z, n_mc, k)), 1e-10)

    # ---- Energy / MSP, normaliser ----
    c = int(rng.choice([1, 2, 7, 10, 16, 17, 64, 65, 100, 252, 256, 1000, 1001, 2048, 2052, 3000]))
    n = int(rng.choice([1, 7, 8, 9, 513, 4097]))
    lg = (rng.standard_normal((n, c)) * float(rng.choice([0.1, 3.0, 40.0]))).astype(np.float32)
    lse, msp = _hip.row_lse_msp(dev(lg), True, True)
    check("energy", (n, c), rel(lse.cpu().numpy(), oracle.energy_score(lg)), 2e-6)
    check("msp", (n, c), rel(msp.cpu().numpy(), oracle.msp_score(lg)), 2e-6)
    check("normalizer", (n, c), rel(_hip.l2_normalize(dev(lg)).cpu().numpy(), oracle.normalizer(lg)), 1e-6)

    # ---- PCA transform + MD around the tile switches ----
    dd = int(rng.choice([8, 40, 512, 520]))
    nn = int(rng.choice([3, 16, 40, 250, 256, 260]))
    rows = int(rng.choice([1, 15, 17, 33, 1000, 8191, 32768 - 31, 32768 + 1, 40_000]))
    h = rng.standard_normal((rows, dd))
    nn = min(nn, dd)
    comp = np.linalg.qr(rng.standard_normal((dd, nn)))[0].T
    mean, var = rng.standard_normal(dd), rng.random(nn) + 0.1
    am = rng.standard_normal((nn, nn))
    prec = am @ am.T / nn + np.eye(nn)
    md_mean = rng.standard_normal((1, nn)) * 0.1
    y = _hip.pca_transform(dev(h), _hip.pack_weights(dev(comp.T)), dev((mean.reshape(1, -1) @ comp.T).ravel()), dev(np.sqrt(var)), nn)
    sub = np.unique(np.r_[0:min(rows, 40), max(0, rows - 40):rows])
    y_exp = oracle.pca_transform(h[sub], comp, mean, var)
    check("pca transform", (rows, dd, nn), rel(y[sub].cpu().numpy(), y_exp), 1e-10)
    s = _hip.md_score(y, dev(md_mean.ravel()), _hip.pack_weights(dev(prec)))
    check("md score", (rows, nn), rel(s[sub].cpu().numpy(), oracle.md_score(y_exp, md_mean, prec)), 1e-9)
    s2 = _hip.pca_md_score(dev(h), _hip.pack_weights(dev(comp.T)), dev((mean.reshape(1, -1) @ comp.T).ravel()), dev(np.sqrt(var)),
                           dev(md_mean.ravel()), _hip.pack_weights(dev(prec)), nn)
    check("pca_md (K2)", (rows, dd, nn), rel(s2[sub].cpu().numpy(), oracle.md_score(y_exp, md_mean, prec)), 1e-9)

    # ---- K2' (folded PCA + LaREM): store and accumulate forms over the component-count and batch-size switches ----
    rk = int(rng.choice([1, 3, 16, 64, 65, 128, 129, 256, 300]))
    dk2 = int(rng.choice([8, 100, 512]))
    nk = int(rng.choice([1, 15, 17, 2000, 4097, 8193, 16384, 20000]))
    hk = rng.standard_normal((nk, dk2))
    mk = rng.standard_normal((dk2, rk)) / np.sqrt(dk2)
    ck = rng.standard_normal(rk)
    ek = -((hk @ mk + ck) ** 2).sum(1)
    pmk = _hip.pack_weights(dev(mk))
    subk = np.unique(np.r_[0:min(nk, 50), max(0, nk - 50):nk])
    sk = _hip.proj_sq_score(dev(hk), pmk, dev(ck), rk).cpu().numpy()
    check("K2' store", (nk, dk2, rk), rel(sk[subk], ek[subk]), 1e-11)
    ak = torch.zeros(nk, dtype=torch.float64, device="cuda")
    _hip.proj_sq_accumulate(dev(hk), pmk, dev(ck), rk, ak)
    check("K2' accumulate == store", (nk, dk2, rk), 0.0 if np.array_equal(ak.cpu().numpy(), sk) else 1.0, 0.5)

    # ---- Mahalanobis: fused epilogue (C <= 16), matrix-core class terms (C > 16) and the class loop ----
    cm_n, dm, nm = int(rng.choice([1, 2, 10, 16, 17, 40, 130, 300])), int(rng.choice([3, 32, 100, 260])), int(rng.choice([1, 31, 33, 500]))
    ft = np.float32 if rng.random() < 0.7 else np.float64
    cen = (rng.standard_normal((cm_n, dm)) * 2 + float(rng.choice([0.0, 20.0]))).astype(ft)
    if cm_n > 2 and rng.random() < 0.3:
        cen[1] = np.nan
    xm = (np.nan_to_num(cen[rng.integers(0, cm_n, nm)]) + rng.standard_normal((nm, dm))).astype(ft)
    am2 = rng.standard_normal((dm, dm))
    pm = am2 @ am2.T / dm + 0.1 * np.eye(dm)
    with np.errstate(all="ignore"):
        em = oracle.mahalanobis_score(xm, cen, pm, cm_n)
        mup = cen.astype(np.float64) @ pm
    tdt = torch.float32 if ft == np.float32 else torch.float64
    pk = _hip.pack_weights(dev(pm))
    for loop in (False, True):
        gm = _hip.mahalanobis_score(dev(xm).to(tdt), dev(cen).to(tdt), pk, dev(mup), class_loop=loop).cpu().numpy()
        check("mahalanobis" + (" (class loop)" if loop else ""), (nm, dm, cm_n, ft.__name__), rel(gm, em), 1e-9)

    # ---- sampler alone on arbitrary maps: register kernels (<= 64 positions), matrix-core contraction (more), generic ----
    hh, ww = int(rng.integers(1, 33)), int(rng.integers(1, 33))
    if hh * ww > 1024:
        hh = 1024 // ww
    nm = int(rng.choice([1, 2, 5, 16, 17, 32, 40]))
    if nm * hh * ww > 16000:  # LDS limit of the mask builders (n_mc * H * W + n_mc floats in 64 KB for small maps)
        nm = max(1, 16000 // (hh * ww))
    cc, nn2, bsz = int(rng.choice([1, 15, 64, 65, 130])), int(rng.integers(1, 4)), int(rng.integers(1, min(hh, ww) + 1))
    pp = float(rng.choice([0.0, 0.1, 0.4, 0.8]))
    xs = np.maximum(rng.standard_normal((nn2, cc, hh, ww)), 0).astype(np.float32) * float(rng.choice([1e-2, 1.0, 30.0]))
    rs = rng.random((nn2, nm, hh, ww)).astype(np.float32)
    try:
        gz = _hip.mc_stack(dev(xs), dev(rs) if pp > 0 else None, nm, pp, bsz).cpu().numpy().reshape(nn2, nm, cc)
    except Exception:
        print("sampler call failed for", (nn2, cc, hh, ww, nm, bsz, pp), flush=True)
        raise
    with np.errstate(all="ignore"):
        ez = np.stack([oracle.mc_stack(xs[i:i + 1], rs[i], pp, bsz) for i in range(nn2)])
    fin = np.isfinite(ez)
    okz = np.array_equal(np.isfinite(gz), fin) and np.allclose(gz[fin], ez[fin], rtol=2e-6, atol=1e-30)
    check("sampler", (nn2, cc, hh, ww, nm, bsz, pp), 0.0 if okz else 1.0, 0.5)

    # ---- kNN ----
    m, dk = int(rng.choice([1, 50, 129, 1000, 5000])), int(rng.choice([3, 32, 100, 512]))
    nq, kk = int(rng.choice([1, 5, 127, 129, 300])), int(rng.choice([1, 5, 50]))
    bank = rng.standard_normal((m, dk)).astype(np.float32)
    if rng.random() < 0.3 and m > 10:
        bank[m // 2:] = bank[: m - m // 2]  # duplicated rows: ties at the k-th distance
    q = rng.standard_normal((nq, dk)).astype(np.float32)
    bn = oracle.normalizer(bank)
    got = _hip.knn_kth(_hip.l2_normalize(dev(q)), dev(bn), kk).cpu().numpy()
    check("knn", (nq, m, dk, kk), rel(got, oracle.knn_kth_score(bn, q, kk, chunk=64)), 2e-5)

    # ---- kNN, large problems: candidate distances from bf16 piece products vs the f32 matrix-core kernel ----
    # (the same entry point takes the bf16 kernel when the workspace holds the planes; both feed the exact f32
    #  re-measurement, so the scores have to agree bit for bit; a few rows against the oracle)
    if t % 3 == 0:
        nq2 = int(rng.choice([1024, 1300, 2048, 3000, 9000, 17000, 33000]))  # (9 000 / 17 000: two chunks of the candidate filter)
        m2 = int(rng.choice([4096, 5000, 8192, 12001]))
        d2 = int(rng.choice([9, 16, 40, 64, 128, 256, 300, 512, 1000, 2048]))  # (narrow features take the bf16 kernel since round 4)
        k2 = int(rng.choice([1, 5, 50, 200, 513, 1500]))  # > 512: the three-read selection; 1500 < every bank size here
        scale_rows = rng.random() < 0.4
        bank2 = rng.standard_normal((m2, d2)).astype(np.float32)
        q2 = rng.standard_normal((nq2, d2)).astype(np.float32)
        if scale_rows:  # un-normalised rows over several orders of magnitude
            bank2 *= (10.0 ** rng.uniform(-2, 2, size=(m2, 1))).astype(np.float32)
            q2 *= (10.0 ** rng.uniform(-2, 2, size=(nq2, 1))).astype(np.float32)
        else:
            bank2 /= np.linalg.norm(bank2, axis=1, keepdims=True)
            q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
        dup = rng.random()
        if dup < 0.5:
            bank2[m2 // 3: m2 // 3 + 300] = bank2[7]  # copied rows: ties around the k-th distance
            q2[:20] = bank2[100:120]                  # exact hits
        elif dup < 0.7:
            # more copies than a candidate list holds, and a crowd of queries next to them: those rows overflow their lists
            # and go through the dense kernels (several overflow rounds when more than 8 192 of them do)
            bank2[m2 // 4: m2 // 4 + 3000] = bank2[7]
            nn = nq2 // 2
            q2[:nn] = bank2[7] + 0.05 * q2[:nn] * np.abs(bank2[7]).mean()
            if not scale_rows:
                q2[:nn] /= np.linalg.norm(q2[:nn], axis=1, keepdims=True)
        lib = _hip.load_library()
        qd, bd = dev(q2), dev(bank2)
        outs = []
        full = lib.runia_knn_workspace_bytes(nq2, m2, d2, k2)
        f32_only = (min(nq2, 8192) * m2 + min(nq2, 8192) + m2 + 4) * 4
        for ws_bytes in (full, f32_only):
            ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device="cuda")
            o = torch.full((nq2,), 123.0, device="cuda")
            rc = lib.runia_knn_kth_f32(qd.data_ptr(), bd.data_ptr(), o.data_ptr(), ws.data_ptr(), ws_bytes, nq2, m2, d2, k2,
                                       torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            outs.append(o.cpu().numpy())
            del ws
        took16 = lib.runia_knn_piece_products(nq2, m2, d2) > 0  # (below 2^31 multiply-adds both calls take the f32 kernel)
        same = (full > f32_only) == took16 and np.array_equal(outs[0], outs[1])
        check("knn bf16 vs f32 kernel", (nq2, m2, d2, k2, scale_rows), 0.0 if same else 1.0, 0.5)
        pick = [0, 19, nq2 // 2, nq2 - 1]
        check("knn bf16 vs oracle", (nq2, m2, d2, k2, scale_rows),
              rel(outs[0][pick], oracle.knn_kth_score(bank2, q2[pick], k2, normalize=False)), 2e-5)

    # ---- LaRED: direct and matrix-core kernels against the exact definition ----
    dl, mt, nx = int(rng.choice([1, 2, 8, 16, 23, 24, 40, 64, 100])), int(rng.choice([1, 10, 700, 3000])), int(rng.choice([1, 63, 65, 900]))
    tr, x = rng.standard_normal((mt, dl)), rng.standard_normal((nx, dl)) * float(rng.choice([0.5, 1.0, 3.0]))
    bw = float(rng.choice([0.5, 1.0, 4.0]))
    exp = oracle.kde_score(tr, x, bw)
    check("kde direct", (nx, mt, dl, bw), rel(_hip.kde_score(dev(tr), dev(x), bw).cpu().numpy(), exp), 1e-10)
    st = _hip.kde_pack_train(dev(tr))
    check("kde matrix", (nx, mt, dl, bw), rel(_hip.kde_score_packed(st, dev(x), bw).cpu().numpy(), exp), 1e-9)
    # ---- f4: linear head (row-streaming kernel up to 16 classes, matrix cores beyond), GEN, ASH-S ----
    ch, dh, nh = int(rng.choice([1, 2, 10, 16, 17, 100, 1000])), int(rng.choice([4, 64, 512, 516, 2048])), int(rng.choice([1, 63, 129, 700]))
    xh = np.maximum(rng.standard_normal((nh, dh)), 0).astype(np.float32) * 2
    wh = (rng.standard_normal((ch, dh)) / np.sqrt(dh)).astype(np.float32)
    bh = rng.standard_normal(ch).astype(np.float32)
    clip = float(rng.choice([np.inf, 1.0]))
    xc = np.minimum(xh, np.float32(clip)).astype(np.float64)
    lin = _hip.linear(dev(xh), dev(wh), dev(bh), clip).cpu().numpy()
    sc = np.abs(xc) @ np.abs(wh.astype(np.float64)).T + 1.0
    check("linear", (nh, dh, ch, clip), float((np.abs(lin - (xc @ wh.astype(np.float64).T + bh)) / sc).max()), 3e-6)
    cg, mg = int(rng.choice([1, 2, 7, 10, 16, 17, 100, 1000, 1500])), int(rng.choice([1, 3, 10, 100, 600, 2000]))
    lgg = (rng.standard_normal((int(rng.choice([1, 65, 300])), cg)) * float(rng.choice([0.5, 3.0]))).astype(np.float32)
    # GEN in f32 is ill-conditioned for confident rows ((1 - p)^gamma with p -> 1 cancels: the reference's own f32 result
    # carries an error of gamma * ulp(p) / (1 - p)); a row counts as matching when it is within 1e-5 of the f32 oracle OR
    # no further from the f64 value than twice the f32 oracle is
    gg = _hip.gen_score(dev(lgg), 0.1, mg).cpu().numpy().astype(np.float64)
    o32 = oracle.gen_score(lgg, 0.1, mg).astype(np.float64)
    o64 = oracle.gen_score(lgg.astype(np.float64), 0.1, mg)
    okg = (np.abs(gg - o32) <= 1e-5 * np.maximum(1.0, np.abs(o32))) | (np.abs(gg - o64) <= np.maximum(1e-5, 2.0 * np.abs(o32 - o64)))
    check("gen", (lgg.shape, mg), 0.0 if bool(okg.all()) else float(np.abs(gg - o32).max()), 0.5e-5)
    da, pa = int(rng.choice([8, 100, 512, 2048, 3000])), int(rng.choice([0, 50, 85, 90, 100]))
    xa = np.maximum(rng.standard_normal((int(rng.choice([1, 9, 130])), da)), 0).astype(np.float32) + np.float32(0.01)
    check("ash_s", (xa.shape, pa), rel(_hip.ash_s(dev(xa), pa).cpu().numpy(), oracle.ash_s_defined(xa, pa)), 2e-6)
    # round 5: the k-th-largest search packs its keys down through LDS as the range narrows - rows whose values crowd into
    # one binade / a few distinct values / carry signs and outliers, at every register count; the kept SET must be the oracle's
    da2 = int(rng.choice([65, 300, 513, 1000, 1024, 2049, 4096]))
    kind = int(rng.integers(0, 5))
    rows2 = int(rng.choice([1, 8, 70]))
    if kind == 0: xa2 = 1.0 + rng.random((rows2, da2))
    elif kind == 1: xa2 = np.round(rng.random((rows2, da2)) * 4) / 4 + 0.25
    elif kind == 2: xa2 = rng.standard_normal((rows2, da2))
    elif kind == 3: xa2 = np.where(rng.random((rows2, da2)) < 0.02, 1e30, rng.random((rows2, da2)))
    else: xa2 = np.exp(rng.standard_normal((rows2, da2)) * 8)
    xa2 = xa2.astype(np.float32)
    pa2 = int(rng.choice([10, 50, 65, 90]))
    with np.errstate(all="ignore"):
        ea2 = oracle.ash_s_defined(xa2.copy(), pa2)
    ga2 = _hip.ash_s(dev(xa2), pa2).cpu().numpy()
    fin = np.isfinite(ea2)
    same_set = np.array_equal(np.isfinite(ga2), fin) and np.array_equal((ga2 != 0) & fin, (ea2 != 0) & fin)
    check("ash_s kept set (crowded rows)", (xa2.shape, kind, pa2), 0.0 if same_set else 1.0, 0.5)
    check("ash_s values (crowded rows)", (xa2.shape, kind, pa2), rel(ga2[fin], ea2[fin]), 1e-4)
    # GEN with the exponent and the softmax's spread drawn too (transcendental-unit exp2 / log2 with both ends rescaled)
    gam = float(rng.choice([0.05, 0.1, 0.5, 1.0, 2.0]))
    # (confident rows: (1 - p) ** gamma in float32 moves by gamma * ulp(p) / (1 - p) per ulp of p - the kernel and the float32
    # oracle both carry that error against the float64 value and which of the two is closer on a row is luck
    # (tools/debug/gen_peaked.py: the same mean error); rows are accepted as above, the batch by its mean error)
    lg3 = (rng.standard_normal((512, cg)) * float(rng.choice([1.0, 6.0]))).astype(np.float32)
    g3 = _hip.gen_score(dev(lg3), gam, mg).cpu().numpy().astype(np.float64)
    with np.errstate(all="ignore"):
        o32b = oracle.gen_score(lg3, gam, mg).astype(np.float64)
        o64b = oracle.gen_score(lg3.astype(np.float64), gam, mg)
    eg3, eo3 = np.abs(g3 - o64b), np.abs(o32b - o64b)
    okb = (np.abs(g3 - o32b) <= 1e-5 * np.maximum(1.0, np.abs(o32b))) | (eg3 <= np.maximum(1e-5, 2.0 * eo3))
    check("gen (gamma, spread drawn): rows off by more than the oracle's own error", (lg3.shape, mg, gam), float(1.0 - okb.mean()), 0.03)
    # (in units of the scores' size: a sum over 1 500 classes is ~40, one float32 step of it 4e-6)
    # (the 90th percentile, not the mean: ONE row with its winner an ulp of p away from the oracle's carries an error of 0.1)
    check("gen (gamma, spread drawn): 90th-percentile error against float64", (lg3.shape, mg, gam),
          float((np.percentile(eg3, 90) - 1.5 * np.percentile(eo3, 90)) / max(1.0, np.abs(o64b).mean())), 3e-7)
    # covariance: upper-triangle tile pairs + mirroring finish, widths / row counts around the tile, vector and slice edges
    nc_, dc_ = int(rng.choice([1, 7, 255, 257, 1000, 4100])), int(rng.choice([1, 3, 64, 127, 129, 200, 260, 516]))
    xc_ = (rng.standard_normal((nc_, dc_)) * (0.5 + rng.random(dc_)) + 3.0 * rng.standard_normal(dc_)).astype(rng.choice([np.float32, np.float64]))
    mc_, cc_ = _hip.covariance(dev(xc_))
    cc_ = cc_.cpu().numpy()
    check("covariance", (nc_, dc_, xc_.dtype.name), rel(cc_, np.cov(xc_.astype(np.float64).T, bias=1).reshape(dc_, dc_)), 1e-12)
    check("covariance symmetric", (nc_, dc_), 0.0 if np.array_equal(cc_, cc_.T) else 1.0, 0.5)

    # ---- f2: metrics on the device (ties, scores inside and outside [0, 1], both dtypes, uneven set sizes) ----
    ni, no = int(rng.choice([1, 7, 300, 5000, 70_000])), int(rng.choice([1, 9, 400, 4097, 50_000]))
    mdt = np.float32 if rng.random() < 0.5 else np.float64
    si = (rng.standard_normal(ni) + 0.7).astype(mdt)
    so = rng.standard_normal(no).astype(mdt)
    if rng.random() < 0.3:
        si, so = np.round(si, 1), np.round(so, 1)  # heavy ties
    if rng.random() < 0.3:
        si, so = (1 / (1 + np.exp(-si))).astype(mdt), (1 / (1 + np.exp(-so))).astype(mdt)  # inside [0, 1]: no sigmoid applied
    gm3 = _hip.ood_metrics(dev(si), dev(so)).cpu().numpy()
    em3 = np.array(oracle.auroc_fpr95_aupr(si, so), dtype=np.float64)
    check("metrics", (ni, no, mdt.__name__), float(np.abs(gm3 - em3).max()), 3e-6)

    # ---- metrics on skewed sets: a tight cluster (one bucket of the split: the single-workgroup radix path) + outliers ----
    if t % 4 == 1:
        nc = int(rng.choice([3000, 40_000, 300_000]))
        centre, width = float(rng.choice([0.5, 3.0, -200.0])), float(rng.choice([1e-12, 1e-9, 1e-4]))
        si = np.concatenate([centre + width * rng.random(nc), rng.standard_normal(5) * 50])
        so = np.concatenate([centre + width * (rng.random(nc // 2) - 0.3), rng.standard_normal(3) * 50])
        if rng.random() < 0.3:
            si[:3], so[:2] = np.inf, -np.inf
        # the sigmoid is taken on the host here (scores inside [0, 1] go through none on the device): for scores packed
        # within 1e-12 the squashed values collapse into a few hundred ties, and WHICH scores tie then hangs on the last
        # bit of exp - device and host libm differ there by an ulp (1e-5 in the metrics; the LSD sort of rounds 2-3
        # returns the same numbers: tools/debug/metrics_clustered.py).  What is checked is the sort.
        with np.errstate(over="ignore"):
            si, so = 1.0 / (1.0 + np.exp(-si)), 1.0 / (1.0 + np.exp(-so))
        gm3 = _hip.ood_metrics(dev(si), dev(so)).cpu().numpy()
        em3 = np.array(oracle.auroc_fpr95_aupr(si, so), dtype=np.float64)
        check("metrics clustered", (nc, centre, width), float(np.abs(gm3 - em3).max()), 3e-6)

    # ---- round 4: pred_h / mi, ASH-S of long rows / conv maps, GEN on probabilities, the other KDE kernels, MD on few
    #      wide rows (column-split + replay = the one launch's bits), roi_align folded into the sampler (= the two calls) ----
    cu, mcu, nu = int(rng.choice([2, 10, 16, 17, 43, 64, 65, 100, 257, 1000, 1025])), int(rng.choice([2, 5, 16, 32])), int(rng.choice([1, 7, 64, 300]))
    lu = (rng.standard_normal((nu * mcu, cu)) * float(rng.choice([0.5, 2.0, 6.0]))).astype(np.float32)
    ph, mi, pr = _hip.mcd_uncertainty(dev(lu), mcu, True)
    eph, emi = oracle.predictive_uncertainty(lu, mcu)
    e = np.exp(lu - lu.max(1, keepdims=True))
    check("pred_h", (nu, mcu, cu), rel(ph.cpu().numpy(), eph), 3e-6)
    check("mi", (nu, mcu, cu), float(np.nanmax(np.abs(mi.cpu().numpy() - emi))), 3e-6)
    check("mcd softmax", (nu, mcu, cu), rel(pr.cpu().numpy(), e / e.sum(1, keepdims=True)), 3e-6)
    dl2, pl2 = int(rng.choice([4097, 5000, 6272, 9000])), int(rng.choice([0, 65, 90, 100]))
    xl2 = np.maximum(rng.standard_normal((int(rng.choice([1, 5])), dl2)), 0).astype(np.float32) + np.float32(0.01)
    if rng.random() < 0.5:
        xl2[:, 7:40] = xl2[:, 6:7]  # ties at or around the threshold
    check("ash_s long rows", (xl2.shape, pl2), rel(_hip.ash_s(dev(xl2), pl2).cpu().numpy(), oracle.ash_s_defined(xl2, pl2)), 2e-6)
    shp = (int(rng.choice([1, 3])), int(rng.choice([8, 96])), int(rng.choice([4, 7])), int(rng.choice([4, 7])))
    xc4 = (np.abs(rng.standard_normal(shp)) + 0.01).astype(np.float32)
    pc4 = int(rng.choice([50, 65, 90]))
    tx = dev(xc4.copy())
    yc = _hip.ash_s_conv(tx, pc4, True)
    ey, ep = oracle.ash_s_conv_defined(xc4, pc4)
    check("ash_s conv", (shp, pc4), rel(yc.cpu().numpy(), ey), 2e-6)
    check("ash_s conv pruned in place", (shp, pc4), 0.0 if np.array_equal(tx.cpu().numpy(), ep) else 1.0, 0.5)
    pg = e / e.sum(1, keepdims=True)
    mg2 = int(rng.choice([1, 3, 10, 100]))
    gp = _hip.gen_entropy(dev(pg.astype(np.float32)), 0.1, mg2).cpu().numpy().astype(np.float64)
    o32 = oracle.generalized_entropy(pg.astype(np.float32), 0.1, mg2).astype(np.float64)
    o64 = oracle.generalized_entropy(pg.astype(np.float32).astype(np.float64), 0.1, mg2)
    okp = (np.abs(gp - o32) <= 1e-5 * np.maximum(1.0, np.abs(o32))) | (np.abs(gp - o64) <= np.maximum(1e-5, 2.0 * np.abs(o32 - o64)))
    check("gen on probabilities", (pg.shape, mg2), 0.0 if bool(okp.all()) else float(np.abs(gp - o32).max()), 0.5e-5)
    kern = str(rng.choice(["tophat", "epanechnikov", "exponential", "linear", "cosine"]))
    dk2, hk = int(rng.choice([1, 2, 3, 5])), float(rng.choice([0.7, 1.5, 3.0]))
    trk, xk = rng.standard_normal((int(rng.choice([50, 700])), dk2)), rng.standard_normal((int(rng.choice([1, 90])), dk2)) * 2.0
    gk = _hip.kde_score_kernel(dev(trk), dev(xk), hk, kern).cpu().numpy()
    ek = oracle.kde_score_kernel(trk, xk, hk, kern)
    both = np.isfinite(ek)
    check("kde " + kern, (trk.shape, xk.shape, hk),
          (rel(gk[both], ek[both]) if both.any() else 0.0) + (0.0 if np.array_equal(np.isneginf(gk), np.isneginf(ek)) else 1.0), 1e-9)
    if t % 5 == 0:
        nf, nr = int(rng.choice([257, 300, 1024, 2048])), int(rng.choice([1, 8, 17, 200, 600]))
        am = rng.standard_normal((nf, nf))
        pk = _hip.pack_weights(dev(am @ am.T / nf + np.eye(nf)))
        xm, mm = dev(rng.standard_normal((nr, nf)).astype(np.float32)), dev(rng.standard_normal(nf).astype(np.float32))
        one = torch.empty(nr, dtype=torch.float64, device="cuda")
        assert _hip.load_library().runia_md_score_f32(xm.data_ptr(), mm.data_ptr(), pk.data_ptr(), one.data_ptr(), nr, nf,
                                                      torch.cuda.current_stream().cuda_stream) == 0
        check("md few rows = one launch", (nr, nf), 0.0 if torch.equal(_hip.md_score(xm, mm, pk), one) else 1.0, 0.5)
    if t % 5 == 2:
        osz, sr, nm = int(rng.choice([4, 7, 8])), int(rng.choice([1, 2])), int(rng.choice([9, 16, 32]))
        if osz == 4 and sr == 1:
            sr = 2
        if osz == 8 and sr == 1:
            sr = 2
        cr, hr, wr = int(rng.choice([3, 64, 130])), int(rng.choice([5, 20, 40])), int(rng.choice([6, 33]))
        fmr = torch.relu(dev(rng.standard_normal((2, cr, hr, wr)).astype(np.float32)))
        kr = int(rng.choice([1, 37]))
        xy = rng.uniform(-30, 16.0 * wr, size=(kr, 2)).astype(np.float32)
        bx = dev(np.concatenate([xy, xy + rng.uniform(1, 8.0 * wr, size=(kr, 2)).astype(np.float32)], axis=1))
        bi = dev(rng.integers(0, 2, size=kr).astype(np.int32))
        rd = dev(rng.random((kr, nm, osz, osz)).astype(np.float32))
        rois = _hip.roi_align(fmr, bx, osz, 1.0 / 16.0, sr, True, bi)
        h2 = _hip.mc_entropy(rois, rd, nm, 0.3, 2, 5)
        h1 = _hip.roi_mc_entropy(_hip.nchw_to_nhwc(fmr), bx, osz, 1.0 / 16.0, sr, True, rd, nm, 0.3, 2, 5, batch_idx=bi)
        same = torch.equal(torch.nan_to_num(h1, nan=-7.0), torch.nan_to_num(h2, nan=-7.0))
        check("roi folded = two calls", (osz, sr, nm, cr, hr, wr, kr), 0.0 if same else 1.0, 0.5)

    if t % 10 == 7:
        # row GEMM on batches of more than one round of 32-row tiles (round 4: whole rounds first, the rest as smaller units -
        # 16-row tiles, KDE: column-split units + replay): slices scored alone (16-row tiles) equal the batch bit for bit,
        # wherever the boundary falls on this device
        nb_, dd_, nn_ = int(rng.integers(33_000, 140_000)), int(rng.choice([64, 128, 256])), int(rng.choice([16, 64, 256]))
        nn_ = min(nn_, dd_)
        gg = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
        hb = torch.randn(nb_, dd_, dtype=torch.float64, device="cuda", generator=gg)
        cp = np.linalg.qr(rng.standard_normal((dd_, nn_)))[0]
        pct_, bias_, scale_ = _hip.pack_weights(dev(cp)), dev(rng.standard_normal(nn_)), dev(rng.random(nn_) + 0.5)
        yb = _hip.pca_transform(hb, pct_, bias_, scale_, nn_)
        ab = rng.standard_normal((nn_, nn_))
        ppb, mmb = _hip.pack_weights(dev(ab @ ab.T / nn_ + np.eye(nn_))), dev(rng.standard_normal(nn_) * 0.1)
        sb = _hip.md_score(yb, mmb, ppb)
        trb = torch.randn(int(rng.choice([200, 700, 3000])), nn_, dtype=torch.float64, device="cuda", generator=gg)
        stb = _hip.kde_pack_train(trb)
        kb = _hip.kde_score_packed(stb, yb, 4.0)
        okb = True
        for lo in (0, 16_384 - 20, 32_768 - 20, 65_536 - 20, 98_304 - 20, nb_ - 40):
            if lo < 0 or lo + 40 > nb_:
                continue
            part = _hip.pca_transform(hb[lo:lo + 40].contiguous(), pct_, bias_, scale_, nn_)
            okb &= torch.equal(part, yb[lo:lo + 40]) and torch.equal(_hip.md_score(part, mmb, ppb), sb[lo:lo + 40])
            okb &= torch.equal(_hip.kde_score_packed(stb, part, 4.0), kb[lo:lo + 40])
        check("row gemm: slices = batch (last-round split)", (nb_, dd_, nn_, trb.shape[0]), 0.0 if okb else 1.0, 0.5)
        smp = rng.integers(0, nb_, size=48)
        y_e = (hb[smp].cpu().numpy() @ cp - bias_.cpu().numpy()) / scale_.cpu().numpy()
        check("row gemm: pca vs numpy (large batch)", (nb_, dd_, nn_), rel(yb[smp].cpu().numpy(), y_e), 1e-10)
        check("row gemm: kde vs oracle (large batch)", (nb_, nn_, trb.shape[0]),
              rel(kb[smp].cpu().numpy(), oracle.kde_score(trb.cpu().numpy(), yb[smp].cpu().numpy(), 4.0)), 1e-9)
        del hb, yb, kb, sb

    # ---- round 6: Cholesky / triangular inverse around the panel switch (128), order statistics of a flat array ----
    if t % 4 == 0:
        dc = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 191, 300, 513, 768, 831, 1000, 1100]))
        bc = int(rng.integers(1, 4))
        ac = rng.standard_normal((bc, dc, dc + int(rng.integers(1, 40))))
        cov = ac @ ac.transpose(0, 2, 1) / ac.shape[2] + float(rng.choice([1e-3, 0.05, 1.0])) * np.eye(dc)
        jit = float(rng.choice([0.0, 1e-3]))
        ref = np.linalg.cholesky(cov + jit * np.eye(dc))
        for dt_, tol_ in ((torch.float64, 1e-9), (torch.float32, 3e-3)):
            L_, info_ = _hip.cholesky(dev(cov, dt_), jit)
            ok_ = int(info_.abs().max()) == 0 and bool((torch.triu(L_, 1) == 0).all())
            check(f"cholesky {dt_}", (dc, bc, jit), rel(L_.cpu().numpy(), ref) if ok_ else 1.0, tol_ * max(1.0, np.abs(ref).max()))
        bad_ = cov.copy()
        jbad = int(rng.integers(0, dc))
        bad_[0, jbad, jbad] = -abs(bad_[0, jbad, jbad]) - 1.0  # the leading minor of order jbad + 1 cannot be positive definite
        info_b = _hip.cholesky(dev(bad_, torch.float64), 0.0)[1].cpu().numpy()
        check("cholesky info", (dc, bc, jbad), 0.0 if (info_b[0] == jbad + 1 and (info_b[1:] == 0).all()) else 1.0, 0.5)
        w_ = _hip.tril_inverse(dev(ref, torch.float64)).cpu().numpy()
        eye_err = max(float(np.max(np.abs(w_[b_] @ ref[b_] - np.eye(dc)))) for b_ in range(bc))
        check("tril inverse", (dc, bc), eye_err if np.allclose(np.triu(w_, 1), 0.0) else 1.0, 1e-9 * max(1.0, float(np.linalg.cond(ref[0]))))
        ns_ = int(rng.choice([1, 2, 255, 256, 257, 70_001, 1_000_003]))
        xs_ = (rng.standard_normal(ns_) * float(rng.choice([1e-30, 1.0, 1e20]))).astype(np.float32)
        if rng.random() < 0.5:
            xs_ = np.maximum(xs_, 0)
        if ns_ > 10 and rng.random() < 0.3:
            xs_[:: int(rng.integers(2, 9))] = xs_[1]
        ranks_ = sorted(set(int(r_) for r_ in rng.integers(0, ns_, size=4)) | {0, ns_ - 1})
        got_ = np.array(_hip.kth_smallest_flat(dev(xs_), ranks_), dtype=np.float32)
        check("radix select", (ns_,), 0.0 if np.array_equal(got_, np.sort(xs_)[ranks_]) else 1.0, 0.5)
        # pinvh on the device (Cholesky route / eigen route) against SciPy; the percentile of a large flat array against NumPy
        from scipy.linalg import pinvh as _pinvh
        from runia_core_amd.device_fit import percentile_flat, pinvh_device
        dp = int(rng.choice([40, 128, 129, 200, 333]))
        rank_ = dp if rng.random() < 0.6 else int(rng.integers(1, dp))
        xa_ = rng.standard_normal((dp, rank_ + (3 * dp if rank_ == dp else 0)))
        spd_ = xa_ @ xa_.T / xa_.shape[1] + (float(rng.choice([0.0, 1e-6, 0.1])) if rank_ == dp else 0.0) * np.eye(dp)
        got_p, ref_p = pinvh_device(dev(spd_)).cpu().numpy(), _pinvh(spd_)
        check("pinvh", (dp, rank_), float(np.max(np.abs(got_p - ref_p)) / np.abs(ref_p).max()), 1e-7)
        if t % 12 == 0:
            big_ = np.maximum(rng.standard_normal(int(rng.choice([1 << 22, (1 << 22) + 12345]))).astype(np.float32), 0 if rng.random() < 0.5 else -9)
            qq_ = float(rng.choice([90, 65, 99.5, 12.5]))
            check("percentile_flat", (big_.size, qq_), 0.0 if np.array_equal(percentile_flat(big_, qq_), np.percentile(big_, qq_)) else 1.0, 0.5)

    if (t + 1) % 10 == 0:
        print(f"round {t + 1}/{a.rounds}, mismatches so far: {bad}", flush=True)
print("fuzz done, mismatches:", bad)
sys.exit(1 if bad else 0)
