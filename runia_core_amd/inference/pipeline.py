"""Device-resident LaREM row pipeline (additive API).

``LaRExInference.get_score`` in the reference handles one image per call and goes
device -> host between the sampler and the entropy stage
(``runia_core/inference/image_level.py:96-120``, ``evaluation/entropy.py:58``).
``LaREMPipeline`` is the batched form of exactly the same chain
``mc_sampler -> get_dl_h_z(.)[1] -> apply_pca_transform -> MDLatentSpace.postprocess``
with every intermediate kept in HBM; batch-1 results are identical to the per-image API.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from torch import Tensor

from .. import _hip
from ..dimensionality_reduction import DevicePCA, device_pca_for
from ..evaluation.entropy import MIN_DIST, neighbors_for

__all__ = ["LaREMPipeline", "AsyncScores", "PreparedDraws"]


class PreparedDraws:
    """Keep-flag table of one batch (K0's output), built ahead of its batch on the pipeline's side stream."""

    def __init__(self, table: Tensor, ready: "torch.cuda.Event", slot: int, shape: tuple):
        self.table, self.ready, self.slot, self.shape = table, ready, slot, shape


class AsyncScores:
    """Scores of one batch still in flight on the pipeline's side streams."""

    def __init__(self, scores: Tensor, done: "torch.cuda.Event"):
        self.scores = scores
        self.done = done

    def wait(self) -> Tensor:
        """Make the current stream wait for the batch (no host sync) and return the device scores."""
        torch.cuda.current_stream().wait_event(self.done)
        return self.scores

    def result(self) -> np.ndarray:
        self.done.synchronize()
        return _hip.to_host(self.scores)


class LaREMPipeline:
    """Fitted state on the device + ``score_*`` entry points.

    Args:
        postprocessor: a set-up ``MDLatentSpace`` (LaREM)
        pca_transform: fitted sklearn ``PCA`` / ``DevicePCA`` or ``None``
        mcd_samples_nro: MC samples per image
        drop_block_prob, drop_block_size: DropBlock parameters of the sampler
    """

    def __init__(self, postprocessor, pca_transform, mcd_samples_nro: int, drop_block_prob: float = 0.0,
                 drop_block_size: int = 1):
        self.postprocessor = postprocessor
        self.pca: Optional[DevicePCA] = device_pca_for(pca_transform) if pca_transform is not None else None
        self.n_mc = int(mcd_samples_nro)
        self.k = neighbors_for(self.n_mc)
        self.drop_prob = float(drop_block_prob)
        self.block_size = int(drop_block_size)
        # Row blocks pipelined over two HIP streams in score_latents.  Measured at N = 10 000 (bench workload):
        # 1 block 0.355 ms, 2 blocks 0.424 ms, 4 blocks 0.477 ms per step - the blocks are too small to pay for
        # the extra launches, so pipelining is opt-in (useful from ~10^5 rows per block).
        self.overlap_chunks = 1
        self._side_streams = None
        # Fold PCA transform + centring + the factor of the precision matrix into one contraction at first use
        # (score = -||M h + c||^2, exact algebra in f64; `runia_proj_sq_score_f64`).  Set False to keep the two-stage
        # K2 (`runia_pca_md_score_f64`), which also materialises the projection in LDS.
        self.fold_weights = True
        self._folded = None
        self._folded_fp = None
        self._k0 = None  # side stream + two table buffers of prepare_draws

    # -- K0 ahead of its batch --------------------------------------------------------
    def prepare_draws(self, rand, n: int, h: int, w: int,
                      inputs_ready: Optional["torch.cuda.Event"] = None) -> Optional[PreparedDraws]:
        """Build the keep-flag table of a COMING batch (its draws ``rand``: an ``(n, n_mc, h, w)`` tensor or
        ``CounterDraws``) on a side stream, under whatever the main stream is running: K0 is one wave's latency chain
        per image (~9 us per 10 000 images that leave the chip idle when it runs in line).  Hand the result to
        ``score_latents(..., prepared=...)``.  Two table buffers alternate; a buffer is rewritten only after the sampler
        + entropy launch that read it.  ``inputs_ready``: event after which ``rand`` is valid; default: everything queued
        so far on the caller's stream (safe, but the table launch then starts only after the batch being scored - pass
        the event to get the overlap).  Returns None when the map shape has no fused kernel or n > 65 535."""
        if not (0 < n <= 65535 and _hip.mc_entropy_supported(h, w, self.n_mc, self.k)):
            return None
        if self._k0 is None:
            self._k0 = {"stream": torch.cuda.Stream(), "tables": [None, None], "free": [None, None], "turn": 0}
        k0 = self._k0
        slot = k0["turn"]
        k0["turn"] ^= 1
        nbytes = max(16, int(_hip.load_library().runia_mc_entropy_workspace_bytes(n, h, w, self.n_mc)))
        if k0["tables"][slot] is None or k0["tables"][slot].numel() < nbytes:
            k0["tables"][slot] = torch.empty(nbytes, dtype=torch.uint8, device=_hip.require_gpu())
        s = k0["stream"]
        if inputs_ready is not None:
            s.wait_event(inputs_ready)
        elif not isinstance(rand, _hip.CounterDraws) and rand is not None:
            s.wait_stream(torch.cuda.current_stream())        # the draws are valid once the caller's stream got here
        if k0["free"][slot] is not None:
            s.wait_event(k0["free"][slot])
        with torch.cuda.stream(s):
            _hip.mc_mask_table(rand if isinstance(rand, _hip.CounterDraws) else (None if rand is None else rand.contiguous()),
                               n, h, w, self.n_mc, self.drop_prob if rand is not None else 0.0, self.block_size,
                               out=k0["tables"][slot])
            ready = s.record_event()
        if isinstance(rand, Tensor):
            rand.record_stream(s)
        return PreparedDraws(k0["tables"][slot], ready, slot, (n, h, w))

    # -- stages ---------------------------------------------------------------------
    def stack(self, latents: Tensor, rand: Optional[Tensor]) -> Tensor:
        """``(N, C, H, W)`` f32 -> MC samples ``(N * n_mc, C)`` f32."""
        return _hip.mc_stack(latents, rand, self.n_mc, self.drop_prob if rand is not None else 0.0, self.block_size)

    def entropy(self, z: Tensor) -> Tensor:
        return _hip.kl_entropy_per_dim(z, self.n_mc, self.k, MIN_DIST)

    def entropy_from_latents(self, latents: Tensor, rand: Optional[Tensor], kernel_events: Optional[list] = None,
                             zero_fill: Optional[Tensor] = None, prepared: Optional[PreparedDraws] = None) -> Tensor:
        """Sampler + per-dimension entropy without materialising the MC samples when the map shape is supported
        (``runia_mc_entropy_f32``: keep-flag table launch + sampler/entropy launch), else the two unfused kernels.
        ``prepared``: the batch's table from :meth:`prepare_draws` (``rand`` is then not read)."""
        n, _, h, w = latents.shape
        if prepared is not None:
            if prepared.shape != (n, h, w):
                raise ValueError(f"prepared draws are for a batch of shape {prepared.shape}, got {(n, h, w)}")
            torch.cuda.current_stream().wait_event(prepared.ready)
            out = _hip.mc_entropy(latents, None, self.n_mc, self.drop_prob, self.block_size, self.k, MIN_DIST,
                                  kernel_events=kernel_events, zero_fill=zero_fill, table=prepared.table)
            self._k0["free"][prepared.slot] = torch.cuda.current_stream().record_event()
            return out
        if _hip.mc_entropy_supported(h, w, self.n_mc, self.k):
            return _hip.mc_entropy(latents, rand, self.n_mc, self.drop_prob if rand is not None else 0.0,
                                   self.block_size, self.k, MIN_DIST, kernel_events=kernel_events, zero_fill=zero_fill)
        if zero_fill is not None:
            zero_fill.zero_()
        return self.entropy(self.stack(latents, rand))

    def _md_state(self):
        """(mean f64 [n], packed P) of a set-up ``MDLatentSpace`` or ``None`` for other postprocessors."""
        pp = self.postprocessor
        if pp is None or not hasattr(pp, "_device_state") or not hasattr(pp, "feats_mean"):
            return None
        st = pp._device_state()
        return pp._mean(torch.float64), st["packed_p"]

    def _folded_state(self):
        """(packed M^T, c, r) with M = W diag(1/scale) C, c = W (-bias/scale - mu), precision = W^T W; None when the
        precision matrix is not positive semi-definite to rounding (then the two-stage kernel is used)."""
        pp = self.postprocessor
        fp = (_hip.array_fingerprint(pp.precision), _hip.array_fingerprint(pp.feats_mean),
              None if self.pca is None else id(self.pca))
        if self._folded is None or self._folded_fp != fp:
            self._folded_fp = fp
            # Everything below runs on the device with the library's own kernels (Jacobi eigen-solver, f64 products) and
            # elementwise torch ops: no host LAPACK / BLAS call, whose last bits follow operand alignment and thread
            # count - every rank of a sharded job folds the same fitted arrays into the same weights.
            prec = _hip.to_device(np.asarray(pp.precision, dtype=np.float64), torch.float64)
            lam, vec = _hip.eigh(prec)                                            # ascending, eigenvectors as columns
            top = float(lam.abs().max()) if lam.numel() else 0.0
            if top == 0.0 or float(lam.min()) < -1e-10 * top:
                self._folded = False
                return None
            keep = lam > top * max(prec.shape) * np.finfo(np.float64).eps
            w = (torch.sqrt(lam[keep])[:, None] * vec[:, keep].t()).contiguous()  # (r, n): precision = w.T @ w
            mu = _hip.to_device(np.asarray(pp.feats_mean, dtype=np.float64).ravel(), torch.float64)
            if self.pca is not None:
                comp = _hip.to_device(self.pca.components_host, torch.float64)
                scale = self.pca.scale if self.pca.scale is not None else torch.ones_like(self.pca.bias)
                a = (comp / scale[:, None]).contiguous()                         # (n, D)
                b = -self.pca.bias / scale - mu
                m = _hip.matmul_f64(w, a)                                         # (r, D)
            else:
                b = -mu
                m = w
            c = _hip.matmul_f64(w, b.reshape(-1, 1).contiguous()).reshape(-1).contiguous()
            self._folded = (_hip.pack_weights(m.t().contiguous()), c, int(m.shape[0]))
        return self._folded or None

    def score_entropies(self, h: Tensor) -> Tensor:
        """PCA transform + postprocessor.  LaREM (``MDLatentSpace``) on f64 rows takes the fused
        ``runia_pca_md_score_f64`` launch; anything else goes stage by stage."""
        md = self._md_state()
        if md is not None and h.dtype == torch.float64:
            mean, packed_p = md
            if self.pca is not None and h.shape[1] != self.pca.n_features:
                raise ValueError(f"X has {h.shape[1]} features, but PCA is expecting {self.pca.n_features} features as input.")
            if self.fold_weights:
                folded = self._folded_state()
                if folded is not None:
                    return _hip.proj_sq_score(h, *folded)
            if self.pca is not None:
                if h.shape[1] != self.pca.n_features:
                    raise ValueError(f"X has {h.shape[1]} features, but PCA is expecting {self.pca.n_features} features as input.")
                return _hip.pca_md_score(h, self.pca.packed_ct, self.pca.bias, self.pca.scale, mean, packed_p,
                                         self.pca.n_components)
            return _hip.pca_md_score(h, None, None, None, mean, packed_p, h.shape[1])
        y = self.pca.transform_device(h) if self.pca is not None else h
        return self.postprocessor.postprocess_device(y)

    def _score_h_into(self, h: Tensor, out: Tensor, mean: Tensor, packed_p: Tensor) -> None:
        """LaREM score of entropy rows ``h`` written into ``out`` (folded single contraction or two-stage K2)."""
        folded = self._folded_state() if self.fold_weights else None
        if folded is not None:
            _hip.proj_sq_score(h, *folded, out=out)
        elif self.pca is not None:
            _hip.pca_md_score(h, self.pca.packed_ct, self.pca.bias, self.pca.scale, mean, packed_p,
                              self.pca.n_components, out=out)
        else:
            _hip.pca_md_score(h, None, None, None, mean, packed_p, h.shape[1], out=out)

    # -- chains ---------------------------------------------------------------------
    def score_samples(self, z: Tensor) -> Tensor:
        """Pre-stacked MC samples ``(N * n_mc, D)`` f32 (device) -> scores ``(N,)`` f64 (device)."""
        return self.score_entropies(self.entropy(z))

    def score_latents(self, latents: Tensor, rand, chunks: Optional[int] = None,
                      k1_events: Optional[list] = None, prepared: Optional[PreparedDraws] = None) -> Tensor:
        """Hooked activations ``(N, C, H, W)`` + uniform draws ``(N, n_mc, H, W)`` -> scores ``(N,)``.
        ``rand`` may also be a ``_hip.CounterDraws(seed, first_image)``: the draws are then made inside the keep-flag
        kernel by the counter generator (throughput mode; nothing is read from memory for them).

        Large LaREM batches are cut into ``chunks`` row blocks pipelined over two HIP streams: the sampler +
        entropy kernel (vector ALUs) of block i+1 runs beside the PCA + LaREM kernel (matrix cores) of block i.
        Rows are independent, so the result is identical to the single-launch form.  ``k1_events`` (optional list)
        receives one (start, end) event pair per sampler + entropy launch (the keep-flag table launch before it is
        left out), recorded on the stream the kernel is launched on.  ``prepared``: this batch's keep-flag table from
        :meth:`prepare_draws` (built ahead on a side stream; same bits as the in-line table launch)."""
        n, _, hh, ww = latents.shape
        chunks = self.overlap_chunks if chunks is None else int(chunks)
        md = self._md_state()
        fused = _hip.mc_entropy_supported(hh, ww, self.n_mc, self.k) and md is not None
        counter = isinstance(rand, _hip.CounterDraws)
        if prepared is not None and not fused:
            prepared = None
        if prepared is not None or not fused or chunks <= 1 or n < 2048 * chunks or (rand is not None and not counter and rand.dim() != 4):
            folded = self._folded_state() if (fused and self.fold_weights) else None
            if folded is not None and (self.pca is None or latents.shape[1] == self.pca.n_features):
                # K1 clears the score vector on its way, K2' adds the two column halves of each row into it: no
                # workspace, no combine launch (bit-identical to the store form)
                scores = torch.empty((n,), dtype=torch.float64, device=latents.device)
                h = self.entropy_from_latents(latents, rand, k1_events, zero_fill=scores, prepared=prepared)
                return _hip.proj_sq_accumulate(h, *folded, out=scores)
            return self.score_entropies(self.entropy_from_latents(latents, rand, k1_events if fused else None,
                                                                  prepared=prepared))
        latents = latents.contiguous()
        if rand is not None and not counter:
            rand = rand.contiguous()
        mean, packed_p = md
        main = torch.cuda.current_stream()
        if self._side_streams is None:
            self._side_streams = (torch.cuda.Stream(), torch.cuda.Stream())
        s_k1, s_k2 = self._side_streams
        h = torch.empty((n, latents.shape[1]), dtype=torch.float64, device=latents.device)
        scores = torch.empty((n,), dtype=torch.float64, device=latents.device)
        start = main.record_event()
        s_k1.wait_event(start)
        s_k2.wait_event(start)
        per = -(-n // chunks)
        drop = self.drop_prob if rand is not None else 0.0
        for a in range(0, n, per):
            b = min(a + per, n)
            with torch.cuda.stream(s_k1):
                r_ab = None if rand is None else (rand._replace(first_image=rand.first_image + a) if counter else rand[a:b])
                _hip.mc_entropy(latents[a:b], r_ab, self.n_mc, drop, self.block_size,
                                self.k, MIN_DIST, out=h[a:b], kernel_events=k1_events)
                ready = s_k1.record_event()
            s_k2.wait_event(ready)
            with torch.cuda.stream(s_k2):
                self._score_h_into(h[a:b], scores[a:b], mean, packed_p)
        main.wait_event(s_k1.record_event())
        main.wait_event(s_k2.record_event())
        return scores

    def score_latents_async(self, latents: Tensor, rand: Optional[Tensor], k1_events: Optional[list] = None,
                            inputs_ready: Optional["torch.cuda.Event"] = None) -> AsyncScores:
        """Pipelined form for streams of batches: the sampler + entropy kernel (vector ALUs) runs on one HIP
        stream and the PCA + LaREM kernel (matrix cores) on another, so batch i+1's K1 overlaps batch i's K2.
        Entropy buffers form a ring of two; a batch's K1 waits for the K2 that last read its buffer.
        ``inputs_ready``: event after which ``latents`` / ``rand`` are valid (default: everything queued so far
        on the caller's stream - note that this includes any ``AsyncScores.wait()`` the caller did, which would
        serialise consecutive batches).  Returns immediately with an :class:`AsyncScores` handle."""
        n, c, hh, ww = latents.shape
        md = self._md_state()
        if not (_hip.mc_entropy_supported(hh, ww, self.n_mc, self.k) and md is not None):
            s = self.score_latents(latents, rand)
            return AsyncScores(s, torch.cuda.current_stream().record_event())
        mean, packed_p = md
        if self._side_streams is None:
            self._side_streams = (torch.cuda.Stream(), torch.cuda.Stream())
        s_k1, s_k2 = self._side_streams
        ring = getattr(self, "_ring", None)
        if ring is None or ring["shape"] != (n, c) or ring["device"] != latents.device:
            ring = {"shape": (n, c), "device": latents.device, "slot": 0,
                    "h": [torch.empty((n, c), dtype=torch.float64, device=latents.device) for _ in range(2)],
                    "free": [None, None]}
            self._ring = ring
        slot = ring["slot"]
        ring["slot"] = slot ^ 1
        h = ring["h"][slot]
        scores = torch.empty((n,), dtype=torch.float64, device=latents.device)
        s_k1.wait_event(inputs_ready if inputs_ready is not None else torch.cuda.current_stream().record_event())
        if ring["free"][slot] is not None:
            s_k1.wait_event(ring["free"][slot])
        drop = self.drop_prob if rand is not None else 0.0
        with torch.cuda.stream(s_k1):
            _hip.mc_entropy(latents, rand, self.n_mc, drop, self.block_size, self.k, MIN_DIST, out=h,
                            kernel_events=k1_events)
            ready = s_k1.record_event()
        s_k2.wait_event(ready)
        with torch.cuda.stream(s_k2):
            self._score_h_into(h, scores, mean, packed_p)
            done = s_k2.record_event()
        ring["free"][slot] = done
        # keep the caller's tensors alive for the side streams (caching-allocator stream safety)
        latents.record_stream(s_k1)
        if isinstance(rand, Tensor):
            rand.record_stream(s_k1)
        scores.record_stream(s_k2)
        return AsyncScores(scores, done)

    @property
    def k2_stream(self):
        """Stream on which the scores of ``score_latents_async`` are produced (consumers may enqueue there)."""
        if self._side_streams is None:
            self._side_streams = (torch.cuda.Stream(), torch.cuda.Stream())
        return self._side_streams[1]

    def score_samples_host(self, z: np.ndarray) -> np.ndarray:
        return _hip.to_host(self.score_samples(_hip.to_device(z, torch.float32)))
