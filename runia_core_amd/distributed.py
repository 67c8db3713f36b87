"""Row sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; every stage of the scoring path is
row-local given the fitted state, so the test rows are cut into contiguous blocks of
``ceil(N / world)`` rows, the fitted state is replicated (broadcast once after ``setup``),
each rank scores its block with the HIP kernels, and ONE ``all_gather`` of the padded score
shards returns the full ``(N,)`` vector on every rank.  No collective runs
inside the data path.

Backend ``"nccl"`` is RCCL on ROCm; ``"gloo"`` is supported for tests and for rehearsing the
N > 1 control flow with several ranks on one GPU (device tensors are staged through the host).
"""
from __future__ import annotations

import io
import pickle
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "gather_scores", "sharded_scores", "broadcast_fitted", "ShardedPostprocessor", "OneShotGather"]


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block ``[start, stop)`` of rank ``rank``: blocks of ``ceil(N/world)`` rows, the tail
    ranks may be short or empty."""
    per = -(-n_rows // world) if n_rows > 0 else 0
    start = min(rank * per, n_rows)
    return start, min(start + per, n_rows)


def _active() -> bool:
    return dist.is_available() and dist.is_initialized()


def _world(group) -> Tuple[int, int]:
    if not _active():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _backend(group) -> Optional[str]:
    return str(dist.get_backend(group)) if _active() else None


def gather_scores(local: torch.Tensor, n_rows: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One all_gather of the per-rank score shards (padded to ``ceil(N/world)``) -> ``(N,)`` on every rank, on the
    device of ``local``.  ``out``: optional preallocated ``(world * ceil(N/world),)`` buffer of the shard's dtype and
    device (a steady-state caller passes the same buffer every call)."""
    world, _ = _world(group)
    if not _active():
        return local
    if local.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"gather_scores: score shards must be float32 or float64 on every rank, got {local.dtype}")
    per = -(-n_rows // world) if n_rows > 0 else 0
    if local.numel() == per and local.is_contiguous():
        buf = local  # even split: no padding copy
    else:
        buf = torch.zeros(per, dtype=local.dtype, device=local.device)
        buf[: local.numel()] = local
    if _backend(group) == "nccl":
        if not local.is_cuda:
            raise TypeError("gather_scores: the nccl (RCCL) backend gathers device tensors; got a CPU shard")
        if out is None:
            out = torch.empty(world * per, dtype=local.dtype, device=local.device)
        else:
            assert out.shape == (world * per,) and out.dtype == local.dtype and out.device == local.device
        dist.all_gather_into_tensor(out, buf, group=group)
        return out[:n_rows]
    # gloo: host tensors (device shards are staged through the host; rehearsal / tests only)
    host = buf.cpu() if buf.is_cuda else buf
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    full = torch.cat(parts)[:n_rows]
    return full.to(local.device) if local.is_cuda else full


def sharded_scores(score_fn: Callable, rows, group=None) -> torch.Tensor:
    """Score ``rows`` (anything sliceable along dim 0, identical on every rank) with ``score_fn`` applied to this
    rank's block only; returns the full score vector on every rank."""
    world, rank = _world(group)
    n = len(rows)
    a, b = shard_bounds(n, world, rank)
    local = score_fn(rows[a:b])
    if isinstance(local, np.ndarray):
        local = torch.from_numpy(local)
    return gather_scores(local.reshape(-1), n, group)


# ------------------------------------------------------------------------------------------------------------------
# fitted state: small parts pickled, arrays sent as tensors
# ------------------------------------------------------------------------------------------------------------------
_OOB_MIN_BYTES = 1 << 16


def _torch_dtype_of(np_dtype) -> Optional[torch.dtype]:
    try:
        return torch.from_numpy(np.empty(0, dtype=np_dtype)).dtype
    except TypeError:
        return None


class _ArrayLiftingPickler(pickle.Pickler):
    """Pickles an object graph but leaves every large ndarray / tensor out of the byte stream (``persistent_id``);
    the arrays travel as tensors through ``dist.broadcast`` instead of through pickle + a byte tensor."""

    def __init__(self, file, min_bytes: int):
        super().__init__(file, protocol=pickle.HIGHEST_PROTOCOL)
        self.min_bytes = min_bytes
        self.arrays: List[torch.Tensor] = []
        self.meta: List[tuple] = []
        self._seen = {}

    def persistent_id(self, o):
        if isinstance(o, np.ndarray):
            if o.nbytes < self.min_bytes or o.dtype.hasobject or _torch_dtype_of(o.dtype) is None:
                return None
            key = id(o)
            if key not in self._seen:
                self._seen[key] = len(self.arrays)
                # the memory layout is part of the state: derived quantities (folded weights, mean @ components.T ...)
                # come from host BLAS calls whose summation order follows the strides, and every rank must derive
                # the same bits.  Fortran-ordered arrays (sklearn's components_) travel as their transpose.
                if o.flags.f_contiguous and not o.flags.c_contiguous:
                    self.arrays.append(torch.from_numpy(o.T))
                    self.meta.append(("ndF", tuple(o.T.shape), o.dtype.str))
                else:
                    self.arrays.append(torch.from_numpy(np.ascontiguousarray(o)))
                    self.meta.append(("nd", tuple(o.shape), o.dtype.str))
            return self._seen[key]
        if isinstance(o, torch.Tensor) and not isinstance(o, torch.nn.Parameter):
            if o.numel() * o.element_size() < self.min_bytes or o.is_sparse or o.requires_grad:
                return None
            key = id(o)
            if key not in self._seen:
                self._seen[key] = len(self.arrays)
                self.arrays.append(o.detach().contiguous())
                self.meta.append(("t", tuple(o.shape), str(o.dtype).replace("torch.", "")))
            return self._seen[key]
        return None


class _ArrayPlacingUnpickler(pickle.Unpickler):
    def __init__(self, file, arrays):
        super().__init__(file)
        self._arrays = arrays

    def persistent_load(self, pid):
        return self._arrays[pid]


def broadcast_fitted(obj, src: int = 0, group=None, min_tensor_bytes: int = _OOB_MIN_BYTES):
    """Replicate a fitted state object from rank ``src`` to every rank (setup-time, once).

    The object graph (postprocessors, sklearn ``PCA`` objects, dicts / lists of them ...) is pickled WITHOUT its large
    arrays: every ndarray / tensor of at least ``min_tensor_bytes`` is lifted out and sent with ``dist.broadcast`` as
    a tensor (on the device for RCCL: host -> HBM -> xGMI -> HBM -> host, no pickle pass over the 410 MB bank or the
    33.5 MB precision of cfg3); the receivers rebuild the same graph around the arrays they received.  Device-side caches
    of the postprocessors are not part of the state (their ``__getstate__`` leaves them out) and are rebuilt on first use.
    Returns ``obj`` itself on ``src`` and the replica elsewhere."""
    world, rank = _world(group)
    if world == 1:
        return obj
    nccl = _backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    head = [None, None]
    arrays: List[torch.Tensor] = []
    if rank == src:
        buf = io.BytesIO()
        p = _ArrayLiftingPickler(buf, min_tensor_bytes)
        p.dump(obj)
        head, arrays = [buf.getvalue(), p.meta], p.arrays
    dist.broadcast_object_list(head, src=src, group=group)
    blob, meta = head
    received = []
    for i, (kind, shape, dt) in enumerate(meta):
        tdt = getattr(torch, dt) if kind == "t" else _torch_dtype_of(np.dtype(dt))
        if rank == src:
            t = arrays[i].to(dev)
        else:
            t = torch.empty(shape, dtype=tdt, device=dev)
        if t.numel():
            dist.broadcast(t, src=src, group=group)
        if rank != src:
            t = t.cpu() if t.is_cuda else t
            received.append(t if kind == "t" else (t.numpy().T if kind == "ndF" else t.numpy()))
        del t
    if rank == src:
        return obj
    return _ArrayPlacingUnpickler(io.BytesIO(blob), received).load()


class ShardedPostprocessor:
    """Wrap a set-up postprocessor: scores this rank's block of the rows and gathers the score vector.

    * ``postprocess(test_data, **kw)`` - the reference's host signature: every rank passes the same host rows, gets the
      full ``(N,)`` ndarray back.
    * ``postprocess_device(test_data)`` - every rank holds the same device rows; device scores ``(N,)`` out.
    * ``postprocess_shard(local_rows, n_rows)`` - every rank passes ONLY its own block (a device tensor of the rows
      ``shard_bounds(n_rows, world, rank)``), e.g. produced on that GPU by the backbone; device scores ``(N,)`` out.
      Nothing but the score shards crosses the links.

    Extra keyword arguments of ``postprocess`` are forwarded (and sliced when they are per-row arrays of the same
    length, e.g. ``pred_labels``)."""

    def __init__(self, postprocessor, group=None, device: Optional[torch.device] = None):
        self.postprocessor = postprocessor
        self.group = group
        self.device = device

    def __getattr__(self, name):
        return getattr(self.postprocessor, name)

    def postprocess_shard(self, local_rows: torch.Tensor, n_rows: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        world, rank = _world(self.group)
        a, b = shard_bounds(n_rows, world, rank)
        if local_rows.shape[0] != b - a:
            raise ValueError(f"postprocess_shard: rank {rank} of {world} owns rows [{a}, {b}) of {n_rows}, "
                             f"got a block of {local_rows.shape[0]} rows")
        # An empty tail shard still goes through the postprocessor (its kernels return at once for N == 0) so that the
        # shard carries the dtype the scorer returns: every rank must enter the one all_gather with the same element size.
        local = self.postprocessor.postprocess_device(local_rows).reshape(-1)
        return gather_scores(local, n_rows, self.group, out=out)

    def postprocess_device(self, test_data: torch.Tensor) -> torch.Tensor:
        world, rank = _world(self.group)
        n = test_data.shape[0]
        a, b = shard_bounds(n, world, rank)
        return self.postprocess_shard(test_data[a:b], n)

    def postprocess(self, test_data, **kwargs) -> np.ndarray:
        world, rank = _world(self.group)
        n = len(test_data)
        a, b = shard_bounds(n, world, rank)
        kw = {k: (v[a:b] if hasattr(v, "__len__") and not isinstance(v, str) and len(v) == n else v)
              for k, v in kwargs.items()}
        local = np.ascontiguousarray(self.postprocessor.postprocess(test_data[a:b], **kw)).reshape(-1)
        local = torch.from_numpy(local)
        if _backend(self.group) == "nccl":
            local = local.to(self.device or torch.device("cuda", torch.cuda.current_device()))
        return gather_scores(local, n, self.group).cpu().numpy()

    __call__ = postprocess


class OneShotGather:
    """One-shot all-gather of equal score shards over xGMI (``csrc/p2p.hip``; SURVEY section 5's fallback for a
    latency-bound gather): every rank writes its shard straight into every peer's receive buffer (mapped through HIP IPC)
    and raises a flag; a second small launch waits for the flags of the step and copies the gathered vector out.  Two
    launches on the caller's stream, no host synchronisation, no ring of ``world - 1`` hops.

    Opt-in alternative to ``gather_scores`` (the RCCL ``all_gather_into_tensor``); same result.  All ranks construct it
    collectively (the IPC handles travel through ``all_gather_object`` of the default / given group - any backend) and
    call it the same number of times.  The output buffers alternate between two slots: a gathered vector stays valid until
    the call after the next one.  ``check()`` reports a wait that timed out (a peer that never arrived)."""

    def __init__(self, max_shard_elems: int, dtype: torch.dtype = torch.float64, group=None, timeout_ms: int = 2000):
        import ctypes

        from . import _hip

        self._lib = _hip.load_library()
        _hip.require_gpu()
        self.world, self.rank = _world(group)
        if not (1 <= self.world <= 16):
            raise ValueError("OneShotGather supports up to 16 ranks of one node")
        self.dtype, self.group, self.timeout_ms = dtype, group, int(timeout_ms)
        self.elem = torch.empty((), dtype=dtype).element_size()
        self.capacity = max(1, int(max_shard_elems)) * self.elem
        # Construction is collective, and a rank that fails (no IPC support, out of memory) must not leave the others
        # waiting in a collective: every rank always takes part in both exchanges and all ranks raise together.
        self._own, self._opened, self._peers = None, [], (ctypes.c_void_p * self.world)()
        raw, err = None, None
        try:
            buf = ctypes.c_void_p()
            _hip._check(self._lib.runia_p2p_alloc(self.world, self.capacity, ctypes.byref(buf)), "runia_p2p_alloc")
            self._own = buf.value
            handle = ctypes.create_string_buffer(64)
            _hip._check(self._lib.runia_p2p_export(self._own, handle), "runia_p2p_export")
            raw = bytes(handle.raw)
        except Exception as e:  # noqa: BLE001 - reported to every rank below
            err = repr(e)
        handles = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(handles, raw, group=group)
        else:
            handles[0] = raw
        if err is None and all(h is not None for h in handles):
            try:
                for r in range(self.world):
                    if r == self.rank:
                        self._peers[r] = self._own
                        continue
                    p = ctypes.c_void_p()
                    _hip._check(self._lib.runia_p2p_open(ctypes.create_string_buffer(handles[r], 64), ctypes.byref(p)),
                                "runia_p2p_open")
                    self._peers[r] = p.value
                    self._opened.append(p.value)
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        elif err is None:
            err = "a peer could not allocate or export its buffer"
        errs = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(errs, err, group=group)  # doubles as the barrier: every mapping exists before a write
        else:
            errs[0] = err
        self._seq = 0
        self._out = [None, None]
        if any(e is not None for e in errs):
            self._release()
            raise _hip.RuniaHipError(f"OneShotGather could not be set up on every rank: {[e for e in errs if e][:2]}")

    def _release(self) -> None:
        for p in self._opened:
            self._lib.runia_p2p_close(p)
        if self._own is not None:
            self._lib.runia_p2p_free(self._own)
        self._own, self._opened = None, []

    def __call__(self, local: torch.Tensor, n_rows: int) -> torch.Tensor:
        from . import _hip

        per = -(-n_rows // self.world) if n_rows > 0 else 0
        if local.dtype != self.dtype or not local.is_cuda:
            raise TypeError(f"OneShotGather({self.dtype}): got a {local.dtype} shard on {local.device}")
        if per * self.elem > self.capacity:
            raise ValueError(f"shard of {per} elements exceeds the capacity this gather was built with")
        if per == 0:
            return local.new_empty(0)
        if local.numel() == per and local.is_contiguous():
            shard = local
        else:
            shard = torch.zeros(per, dtype=self.dtype, device=local.device)
            shard[: local.numel()] = local
        self._seq += 1
        slot = self._seq & 1
        if self._out[slot] is None or self._out[slot].numel() != self.world * per:
            self._out[slot] = torch.empty(self.world * per, dtype=self.dtype, device=local.device)
        out = self._out[slot]
        _hip._check(self._lib.runia_p2p_all_gather(shard.data_ptr(), per * self.elem, out.data_ptr(), self._peers, self.world,
                                                   self.rank, self.capacity, self._seq, self.timeout_ms, _hip._stream()),
                    "runia_p2p_all_gather")
        return out[:n_rows]

    def check(self) -> None:
        """Synchronises; raises if any wait so far gave up on a peer."""
        import ctypes

        from . import _hip

        status = ctypes.c_int(0)
        _hip._check(self._lib.runia_p2p_status(self._own, ctypes.byref(status)), "runia_p2p_status")
        if status.value & 2:
            raise _hip.RuniaHipError("OneShotGather: slot-reuse assertion failed - a slot was overwritten before the peer had "
                                     "copied the step before last out of it (runia_p2p_debug)")
        if status.value != 0:
            raise _hip.RuniaHipError("OneShotGather: a wait timed out - a peer never delivered its shard")

    @staticmethod
    def set_debug(on: bool) -> bool:
        """Switch the slot-reuse assertion of the gather launches on / off (every rank alike); returns the previous setting."""
        from . import _hip

        return bool(_hip.load_library().runia_p2p_debug(1 if on else 0))

    def __del__(self):  # best effort for a gather that was never closed: no collective here, just the local resources
        try:
            if getattr(self, "_own", None) is not None:
                torch.cuda.synchronize()
                self._release()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def close(self) -> None:
        """Collective: every rank calls it (a barrier makes sure nobody unmaps a buffer a peer may still be writing)."""
        if self._own is None:
            return
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier(group=self.group)  # nobody unmaps a buffer a peer may still be writing
        self._release()
