#!/usr/bin/env python3
"""Round-2 fixtures, generated from the reference's own source files (build container only).

Same by-path import recipe as ``tools/make_goldens.py`` (bare namespace packages, package ``__init__``
files never run), extended to the two files round 1 could not load:

* ``runia_core/feature_extraction/abstract_classes.py`` (``MCSamplerModule``) - needs ``torchvision.ops.nms``
  at import time only (an inert placeholder) and ``dropblock.DropBlock2D`` at RUN time.  ``dropblock==0.3.0``
  (/root/reference/requirements.txt:2) is absent from the image, so the layer - and only the layer - is the
  ~20-line published algorithm restated below (``_DropBlock2D``).  Everything else that runs is the
  reference's own code: the ``ModuleList`` order, the loop of ``forward``, ``get_mean_or_fullmean_ls_sample``,
  the ``reshape`` / ``cat`` (feature_extraction/abstract_classes.py:81-101, feature_extraction/utils.py:70-92).
* ``runia_core/llm_uncertainty/scores.py`` (``eigen_score``, ``semantic_entropy``) - pure functions.

Sections (``--only a,b``): sampler, kde_hd, eigen, roi.  Only DATA is written (inputs, draws, outputs).

Usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tools/make_goldens_r2.py [--only sampler]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _DropBlock2D(torch.nn.Module):
    """dropblock==0.3.0 ``DropBlock2D`` restated from its published algorithm (the package is absent):
    gamma = p / bs^2; seed mask = rand(B, H, W) < gamma drawn on the CPU default generator; block mask =
    1 - max_pool2d(seed, bs, stride 1, pad bs//2) (last row / column cropped for even bs); x * bm * numel / sum."""

    def __init__(self, drop_prob, block_size):
        super().__init__()
        self.drop_prob = drop_prob
        self.block_size = block_size

    def forward(self, x):
        assert x.dim() == 4
        if not self.training or self.drop_prob == 0.0:
            return x
        gamma = self.drop_prob / (self.block_size**2)
        seed = (torch.rand(x.shape[0], *x.shape[2:]) < gamma).float().to(x.device)
        bm = F.max_pool2d(seed[:, None], kernel_size=(self.block_size, self.block_size), stride=(1, 1),
                          padding=self.block_size // 2)
        if self.block_size % 2 == 0:
            bm = bm[:, :, :-1, :-1]
        bm = 1 - bm.squeeze(1)
        out = x * bm[:, None, :, :]
        return out * bm.numel() / bm.sum()


def _namespaces():
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    ns("runia_core", f"{REF}/runia_core")
    for sub in ("inference", "feature_extraction", "evaluation", "llm_uncertainty"):
        ns(f"runia_core.{sub}", f"{REF}/runia_core/{sub}")
    om = types.ModuleType("omegaconf")
    om.DictConfig = dict
    sys.modules["omegaconf"] = om
    sys.modules["faiss"] = types.ModuleType("faiss")
    db = types.ModuleType("dropblock")
    db.DropBlock2D = _DropBlock2D
    sys.modules["dropblock"] = db
    pm = types.ModuleType("pacmap")
    pm.PaCMAP = type("PaCMAP", (), {})
    sys.modules["pacmap"] = pm
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.__path__ = []
        ops = types.ModuleType("torchvision.ops")

        def _absent(*a, **k):  # import-time names only; never called by the sections below
            raise RuntimeError("torchvision is absent")

        ops.nms = _absent
        ops.roi_align = _absent
        tv.ops = ops
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = ops


# ------------------------------------------------------------------------------------------------
def gen_sampler():
    """MCSamplerModule.forward outputs of the REFERENCE'S OWN code for seeded inputs."""
    import runia_core.feature_extraction.abstract_classes as fac

    cases = {}
    specs = [  # name, C, H, W, n_mc, block_size, drop_prob, layer_type, seed
        ("c4x4_bs2", 48, 4, 4, 16, 2, 0.5, "Conv", 11),
        ("c4x4_bs2_mc32", 40, 4, 4, 32, 2, 0.4, "Conv", 12),
        ("c7x7_bs3", 24, 7, 7, 16, 3, 0.4, "Conv", 13),
        ("c8x8_bs8", 32, 8, 8, 16, 8, 0.5, "Conv", 14),
        ("c8x8_bs4", 32, 8, 8, 16, 4, 0.3, "Conv", 15),
        ("c2x2_bs1", 64, 2, 2, 16, 1, 0.3, "Conv", 116),
        ("c2x2_dead", 16, 2, 2, 16, 1, 0.3, "Conv", 16),   # one layer drops the whole map: 0 * numel / 0 = NaN upstream
        ("c4x4_bs3_mc8", 36, 4, 4, 8, 3, 0.5, "Conv", 17),
        ("c5x6_bs2", 20, 5, 6, 12, 2, 0.35, "Conv", 18),   # non-square map, generic kernel
        ("c4x4_p0", 16, 4, 4, 16, 2, 0.0, "Conv", 19),     # drop_prob 0: identity layers
        ("fc4x4_bs2", 12, 4, 4, 16, 2, 0.5, "FC", 20),
        ("rpn7x7_bs3", 6, 7, 7, 8, 3, 0.4, "RPN", 21),
    ]
    for name, c, h, w, n_mc, bs, p, lt, seed in specs:
        g = np.random.default_rng(seed)
        x = np.maximum(g.standard_normal((1, c, h, w)), 0).astype(np.float32) * (0.25 + g.random((1, c, 1, 1)).astype(np.float32))
        sampler = fac.MCSamplerModule(mc_samples=n_mc, block_size=bs, drop_prob=p, layer_type=lt)
        sampler.train()
        # draws: the stream the reference consumes (one torch.rand(1, H, W) per drop layer, ModuleList order)
        torch.manual_seed(seed)
        draws = torch.cat([torch.rand(1, h, w) for _ in range(n_mc)]).numpy()
        bm = []
        for s in range(n_mc):  # keep only fixtures in which no layer drops the whole map (0/0 upstream)
            seed_mask = torch.from_numpy((draws[s : s + 1] < p / bs**2).astype(np.float32))
            pooled = F.max_pool2d(seed_mask[:, None], (bs, bs), (1, 1), bs // 2)
            if bs % 2 == 0:
                pooled = pooled[:, :, :-1, :-1]
            bm.append(float((1 - pooled).sum()))
        assert min(bm) > 0 or p == 0.0 or name.endswith("_dead"), (name, bm)
        torch.manual_seed(seed)
        with torch.no_grad():
            out = sampler(torch.from_numpy(x)).numpy()
        cases[f"{name}_x"] = x
        cases[f"{name}_draws"] = draws
        cases[f"{name}_out"] = out
        cases[f"{name}_params"] = np.array([n_mc, bs, p, {"Conv": 0, "FC": 1, "RPN": 2}[lt], seed], dtype=np.float64)
        print(f"  sampler {name}: out {out.shape}, min mask sum {min(bm):.0f}")
    # eval mode: the drop layers are the identity
    sampler = fac.MCSamplerModule(mc_samples=4, block_size=2, drop_prob=0.5, layer_type="Conv")
    sampler.eval()
    x = np.maximum(np.random.default_rng(30).standard_normal((1, 8, 4, 4)), 0).astype(np.float32)
    with torch.no_grad():
        cases["eval_x"], cases["eval_out"] = x, sampler(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(OUT, "ref_sampler.npz"), **cases)


def kde_hd_inputs(d, seed):
    """Seeded whitened-like embeddings (what PCA(whiten=True) hands LaRED): train / InD test ~ N(0, I_d) with a mild
    per-dimension scale, OOD = shifted and inflated.  Regenerated from the seed by the tests (only the reference's
    scores are stored)."""
    g = np.random.default_rng(seed)
    scale = 0.8 + 0.4 * g.random(d)
    train = g.standard_normal((3000, d)) * scale
    ind = g.standard_normal((600, d)) * scale
    ood = g.standard_normal((600, d)) * scale * 1.15 + 0.35
    return train, ind, ood


def gen_kde_hd():
    """LaRED above the dimension where sklearn's tree KDE stops returning the density (D >~ 20): the REFERENCE'S
    scores (KDELatentSpace -> sklearn KernelDensity.score_samples) for InD and OOD sets at D = 16 (converged), 64, 256."""
    import runia_core.inference.postprocessors as pp

    cases = {}
    for d, seed in ((16, 416), (64, 464), (256, 4256)):
        train, ind, ood = kde_hd_inputs(d, seed)
        p = pp.KDELatentSpace()
        p.setup(train)
        s_ind, s_ood = p.postprocess(ind), p.postprocess(ood)
        cases[f"d{d}_seed"] = np.array(seed)
        cases[f"d{d}_checksum"] = np.array([np.abs(train).sum(), np.abs(ind).sum(), np.abs(ood).sum()])
        cases[f"d{d}_ref_ind"], cases[f"d{d}_ref_ood"] = s_ind, s_ood
        print(f"  kde_hd D={d}: ref ind mean {s_ind.mean():.3f} ood mean {s_ood.mean():.3f}")
    np.savez_compressed(os.path.join(OUT, "ref_kde_hd.npz"), **cases)


def _roi_align_torch(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """torchvision.ops.roi_align restated from its published algorithm (torchvision is absent): float32, bins average
    sampling_ratio^2 bilinear samples, aligned shifts by -0.5, out-of-map samples (more than one pixel) contribute 0."""
    import math

    x = input.detach().cpu().numpy().astype(np.float32)
    bx = (boxes[0] if isinstance(boxes, (list, tuple)) else boxes).detach().cpu().numpy().astype(np.float32)
    _, c, h, w = x.shape
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
    f = np.float32
    out = np.zeros((bx.shape[0], c, ph, pw), np.float32)
    off = f(0.5) if aligned else f(0.0)

    def bil(yy, xx):
        if yy < -1.0 or yy > h or xx < -1.0 or xx > w:
            return np.zeros(c, np.float32)
        yy, xx = max(yy, f(0)), max(xx, f(0))
        yl, xl = int(yy), int(xx)
        if yl >= h - 1:
            yh = yl = h - 1; yy = f(yl)
        else:
            yh = yl + 1
        if xl >= w - 1:
            xh = xl = w - 1; xx = f(xl)
        else:
            xh = xl + 1
        ly, lx = f(yy - f(yl)), f(xx - f(xl))
        hy, hx = f(f(1) - ly), f(f(1) - lx)
        return ((f(hy * hx) * x[0, :, yl, xl] + f(hy * lx) * x[0, :, yl, xh]) + f(ly * hx) * x[0, :, yh, xl]) + f(ly * lx) * x[0, :, yh, xh]

    for k, b in enumerate(bx):
        x1, y1, x2, y2 = (f(f(v * f(spatial_scale)) - off) for v in b)
        rw, rh = f(x2 - x1), f(y2 - y1)
        if not aligned:
            rw, rh = max(rw, f(1)), max(rh, f(1))
        bh, bw = f(rh / f(ph)), f(rw / f(pw))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        cnt = f(max(gh * gw, 1))
        for i in range(ph):
            for j in range(pw):
                acc = np.zeros(c, np.float32)
                for iy in range(gh):
                    yy = f(f(y1 + f(f(i) * bh)) + f(f(f(iy) + f(0.5)) * bh) / f(gh))
                    for ix in range(gw):
                        xx = f(f(x1 + f(f(j) * bw)) + f(f(f(ix) + f(0.5)) * bw) / f(gw))
                        acc = (acc + bil(yy, xx)).astype(np.float32)
                out[k, :, i, j] = acc / cnt
    return torch.from_numpy(out)


def _get_h(x, k=1, norm="max", min_dist=0.0):
    """entropy_estimators==0.0.1 continuous.get_h restated from its published algorithm (the package is absent; pinned by
    the reference's entropy goldens in tests/test_oracle_goldens.py)."""
    from scipy.spatial import cKDTree
    from scipy.special import digamma

    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    n, d = x.shape
    assert norm == "max"
    dist = cKDTree(x).query(x, k + 1, eps=0, p=np.inf)[0][:, -1].copy()
    dist[dist < min_dist] = min_dist
    return -digamma(k) + digamma(n) + (d / float(n)) * np.sum(np.log(2 * dist))


def gen_roi():
    """The reference's own per-ROI glue (_dropblock_rois_get_entropy, _reduce_features_to_rois of
    feature_extraction/object_level.py, loaded by path) on seeded feature maps and boxes, with the three absent third-party
    pieces it calls restated (torchvision.ops.roi_align, dropblock.DropBlock2D, entropy_estimators get_h)."""
    ee = types.ModuleType("entropy_estimators")
    cont = types.ModuleType("entropy_estimators.continuous")
    cont.get_h = _get_h
    ee.continuous = cont
    sys.modules["entropy_estimators"], sys.modules["entropy_estimators.continuous"] = ee, cont
    sys.modules["torchvision.ops"].roi_align = _roi_align_torch
    import runia_core.feature_extraction.abstract_classes as fac
    import runia_core.feature_extraction.object_level as fol

    cases = {}
    specs = [  # name, [(C, Hf, Wf)], output sizes, img (H, W), K boxes, sampling_ratio, n_mc, bs, p, seed
        ("p7", [(24, 25, 40)], (7,), (200, 320), 6, 2, 16, 3, 0.4, 71),
        ("p4x2", [(16, 50, 80), (8, 25, 40)], (4, 4), (400, 640), 5, 2, 16, 2, 0.5, 72),
        ("p8_adaptive", [(12, 32, 32)], (8,), (256, 256), 4, -1, 12, 4, 0.3, 73),
    ]
    for name, maps, osz, img, k, sr, n_mc, bs, p, seed in specs:
        g = np.random.default_rng(seed)
        fms = [np.maximum(g.standard_normal((1, c, h, w)), 0).astype(np.float32) for (c, h, w) in maps]
        x1 = g.uniform(-10, img[1] * 0.6, k); y1 = g.uniform(-10, img[0] * 0.6, k)
        bw = g.uniform(20, img[1] * 0.5, k); bh = g.uniform(20, img[0] * 0.5, k)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        sampler = fac.MCSamplerModule(mc_samples=n_mc, block_size=bs, drop_prob=p, layer_type="Conv")
        sampler.train()
        ph = osz[0]
        torch.manual_seed(seed)
        draws = torch.cat([torch.rand(1, ph, ph) for _ in range(k * n_mc)]).reshape(k, n_mc, ph, ph).numpy()
        torch.manual_seed(seed)
        ent = fol._dropblock_rois_get_entropy([torch.from_numpy(f) for f in fms], osz, torch.from_numpy(boxes), img, sr,
                                              len(fms), n_mc, sampler).numpy()
        means, stds = fol._reduce_features_to_rois([torch.from_numpy(f) for f in fms], osz, torch.from_numpy(boxes), img, sr,
                                                    len(fms), k, return_stds=True)
        for i, f in enumerate(fms):
            cases[f"{name}_fm{i}"] = f
        cases[f"{name}_boxes"], cases[f"{name}_draws"] = boxes, draws
        cases[f"{name}_entropy"] = ent
        cases[f"{name}_means"] = torch.cat(means).numpy()
        cases[f"{name}_stds"] = torch.cat(stds).numpy()
        cases[f"{name}_params"] = np.array([len(fms), osz[0], img[0], img[1], sr, n_mc, bs, p, seed], dtype=np.float64)
        print(f"  roi {name}: entropy {ent.shape} finite {np.isfinite(ent).all()}")
    np.savez_compressed(os.path.join(OUT, "ref_roi.npz"), **cases)


SECTIONS = {"sampler": gen_sampler, "kde_hd": gen_kde_hd, "roi": gen_roi}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(SECTIONS))
    args = ap.parse_args()
    _namespaces()
    os.makedirs(OUT, exist_ok=True)
    for name in args.only.split(","):
        print(f"[{name}]")
        SECTIONS[name]()
    print("fixtures written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
