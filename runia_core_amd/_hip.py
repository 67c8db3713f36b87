"""ctypes binding of ``librunia_hip.so`` (C ABI in ``include/runia_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every
numerical stage of the scoring path is a kernel of the shared library.  There is
NO CPU fallback: without the library or without a GPU the wrappers raise.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p
from typing import NamedTuple, Optional, Union

import numpy as np
import torch  # imported before the library so that both share one HIP runtime

from . import config as _config

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librunia_hip.so")
_lib: Optional[ctypes.CDLL] = None

# name -> (restype, argtypes); mirrors include/runia_hip.h one to one
_SIGNATURES = {
    "runia_abi_version": (c_int, []),
    "runia_error_string": (c_char_p, [c_int]),
    "runia_device_count": (c_int, []),
    "runia_clock_probe": (c_int, [c_void_p, c_int, c_void_p]),
    "runia_time_next_launch": (c_int, [c_void_p, c_void_p]),
    "runia_mc_stack_f32": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_double, c_int, c_void_p],
    ),
    "runia_mc_drop_flat_f32": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_double, c_int, c_void_p],
    ),
    "runia_map_reduce_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "runia_kl_entropy_per_dim_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_double, c_void_p]),
    "runia_kl_entropy_joint_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_double, c_void_p]),
    "runia_kl_entropy_both_fused": (c_int, [c_int, c_int64, c_int]),
    "runia_kl_entropy_both_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_double, c_void_p]),
    "runia_packed_weights_bytes": (c_size_t, [c_int64, c_int64]),
    "runia_pack_weights_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "runia_pca_transform_f64": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p],
    ),
    "runia_pca_transform_f32in": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p],
    ),
    "runia_md_score_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "runia_md_score_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "runia_md_score_f32x_f64mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "runia_md_score_tril_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_md_score_tril_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_md_score_tril_f32x_f64mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_md_score_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "runia_md_score_ws_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_md_score_ws_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_md_score_ws_f32x_f64mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_mahalanobis_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "runia_mahalanobis_workspace_bytes_classes": (c_size_t, [c_int64, c_int64, c_int]),
    "runia_mahalanobis_score_f32": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int, c_void_p],
    ),
    "runia_mahalanobis_score_f64": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int, c_void_p],
    ),
    "runia_row_lse_msp_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "runia_l2_normalize_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "runia_kde_score_kernel_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_double, c_int, c_void_p]),
    "runia_knn_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "runia_knn_piece_products": (c_int, [c_int64, c_int64, c_int64]),
    "runia_knn_bank_state_bytes": (c_size_t, [c_int64, c_int64]),
    "runia_knn_prepare_bank_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_knn_prepared_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "runia_knn_kth_prepared_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_int64, c_int64,
                                           c_int64, c_int, c_void_p]),
    "runia_knn_kth_f32": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int64, c_int, c_void_p],
    ),
    "runia_kde_score_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p]),
    "runia_row_sqnorm_f64": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "runia_kde_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "runia_kde_score_packed_f64": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int64, c_double, c_void_p],
    ),
    "runia_mc_entropy_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "runia_mc_entropy_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "runia_mc_mask_table_f32": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_double, c_int, c_void_p],
    ),
    "runia_mc_stack_table_f32": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_int, c_double, c_int,
         c_void_p],
    ),
    "runia_mc_entropy_from_table_f32": (
        c_int,
        [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int,
         c_double, c_void_p],
    ),
    "runia_mc_entropy_f32": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int,
         c_int, c_double, c_int, c_int, c_double, c_void_p],
    ),
    "runia_mc_draws_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_uint64, c_int64, c_void_p]),
    "runia_mc_mask_table_counter_f32": (
        c_int,
        [c_uint64, c_int64, c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p],
    ),
    "runia_mc_entropy_counter_f32": (
        c_int,
        [c_void_p, c_uint64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int,
         c_int, c_double, c_int, c_int, c_double, c_int, c_void_p],
    ),
    "runia_select_hist_f32": (c_int, [c_void_p, c_void_p, c_int64, ctypes.c_uint32, ctypes.c_uint32, c_int, c_void_p]),
    "runia_cholesky_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p]),
    "runia_cholesky_f64": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p]),
    "runia_gmm_log_prob_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "runia_gmm_log_prob_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64,
                                       c_int64, c_int, c_void_p]),
    "runia_ood_metrics_workspace_bytes": (c_size_t, [c_int64]),
    "runia_ood_metrics_f64": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "runia_ood_metrics_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "runia_ood_clf_curve_f64": (
        c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "runia_ood_clf_curve_f32": (
        c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "runia_eigh_workspace_bytes": (c_size_t, [c_int64]),
    "runia_eigh_init_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "runia_eigh_sweep_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "runia_eigh_block_padded": (c_int64, [c_int64]),
    "runia_eigh_block_workspace_bytes": (c_size_t, [c_int64]),
    "runia_eigh_block_init_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "runia_eigh_block_sweep_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "runia_matmul_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "runia_p2p_buffer_bytes": (c_size_t, [c_int, c_size_t]),
    "runia_p2p_alloc": (c_int, [c_int, c_size_t, ctypes.POINTER(c_void_p)]),
    "runia_p2p_free": (c_int, [c_void_p]),
    "runia_p2p_export": (c_int, [c_void_p, c_void_p]),
    "runia_p2p_open": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "runia_p2p_close": (c_int, [c_void_p]),
    "runia_p2p_all_gather": (
        c_int, [c_void_p, c_size_t, c_void_p, ctypes.POINTER(c_void_p), c_int, c_int, c_size_t, c_uint64, c_int, c_void_p]),
    "runia_p2p_status": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "runia_p2p_debug": (c_int, [c_int]),
    "runia_centred_gram_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p]),
    "runia_roi_align_f32": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_double, c_int, c_int,
         c_void_p],
    ),
    "runia_nchw_to_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "runia_roi_mc_entropy_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "runia_roi_mc_entropy_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "runia_roi_mc_entropy_f32": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int, c_int,
         c_int, c_int, c_int, c_double, c_int, c_int, c_int, c_double, c_int, c_int, c_double, c_void_p],
    ),
    "runia_linear_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "runia_ash_s_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "runia_gen_score_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_double, c_void_p]),
    "runia_gen_entropy_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_double, c_void_p]),
    "runia_mcd_uncertainty_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "runia_ash_s_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "runia_tril_inverse_f64": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "runia_proj_norm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "runia_proj_norm_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "runia_proj_sq_workspace_bytes": (c_size_t, [c_int64]),
    "runia_proj_sq_accumulate_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "runia_proj_sq_score_f64": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int64, c_void_p],
    ),
    "runia_covariance_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "runia_covariance_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_covariance_f32in": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_void_p]),
    "runia_pca_md_score_f64": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
         c_void_p],
    ),
}


class RuniaHipError(RuntimeError):
    pass


class CounterDraws(NamedTuple):
    """DropBlock draws made inside the keep-flag kernel by the counter generator (Philox4x32-10, csrc/philox.hpp):
    image i of the batch uses image id ``first_image + i`` of the stream keyed by ``seed``.  ``redraw_dead_layers``:
    a drop layer that removes the whole map (NaN upstream as well) draws again from the image's next counter block."""

    seed: int
    first_image: int = 0
    redraw_dead_layers: bool = False


def library_path() -> str:
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    """Load the shared library and declare every C-ABI signature.  Works without a GPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            # a missing build artefact is built, never replaced: there is no other implementation to fall back to
            import subprocess

            try:
                subprocess.run(["make", "-C", os.path.join(os.path.dirname(_LIB_PATH), "csrc"), "-j8", "-s"], check=True,
                               stdout=subprocess.DEVNULL)
            except Exception:
                pass
        if not os.path.exists(_LIB_PATH):
            raise RuniaHipError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  runia_core_amd has no CPU fallback."
            )
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def require_gpu() -> torch.device:
    """The scoring path is HIP-only.  Fail loudly when it cannot run."""
    lib = load_library()
    if not torch.cuda.is_available() or lib.runia_device_count() < 1:
        raise RuniaHipError(
            "runia_core_amd: no HIP device visible (torch.cuda.is_available() is False). "
            "The scoring hot path runs only as HIP kernels on MI355X; there is no CPU fallback."
        )
    return torch.device("cuda", torch.cuda.current_device())


def array_fingerprint(a):
    """Cheap identity of a fitted array (address, shape, dtype): reassigning or refitting gives a new array, so caches of
    device copies keyed on it are rebuilt instead of going stale.  (In-place edits of the same buffer are not seen.)"""
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return ("t", a.data_ptr(), tuple(a.shape), str(a.dtype))
    a = np.asarray(a)
    return ("n", a.__array_interface__["data"][0], a.shape, a.dtype.str)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().runia_error_string(rc).decode()
        raise RuniaHipError(f"{what} failed: {msg} (code {rc})")


def _cuda_tensors(value):
    """CUDA tensors inside an argument (the argument itself, or the members of a tuple / list such as a packed state)."""
    if getattr(value, "is_cuda", False) is True:
        yield value
    elif isinstance(value, (tuple, list)):
        for v in value:
            yield from _cuda_tensors(v)


def resolve_device(args, kwargs, exempt=()):
    """The one device a wrapper call runs on: that of its CUDA tensor arguments (``exempt``: names the wrapper moves to
    the first tensor's device itself).  Tensors on different devices raise ``RuniaHipError`` - a kernel launched with
    pointers of two GPUs would fault or, worse, read peer memory silently.  No CUDA tensor -> None (the current device).
    (On the path of every wrapper call - 1 500 per harness sweep: plain loops, no intermediate lists.)"""
    dev, first = None, None
    i = -1
    for value in args:
        i += 1
        if value is None or i in exempt:
            continue
        if type(value) is torch.Tensor:
            if not value.is_cuda:
                continue
            d = value.device
            if dev is None:
                dev, first = d, i
            elif d != dev:
                _raise_two_devices(dev, first, d, i)
        elif isinstance(value, (tuple, list)) or getattr(value, "is_cuda", False) is True:  # packed states, tensor subclasses
            for t in _cuda_tensors(value):
                if dev is None:
                    dev, first = t.device, i
                elif t.device != dev:
                    _raise_two_devices(dev, first, t.device, i)
    for name, value in kwargs.items():
        if value is None or name in exempt:
            continue
        for t in _cuda_tensors(value):
            if dev is None:
                dev, first = t.device, name
            elif t.device != dev:
                _raise_two_devices(dev, first, t.device, name)
    return dev


def _raise_two_devices(dev, first, other, name):
    raise RuniaHipError(f"tensor arguments sit on different devices ({dev} for argument {first!r}, {other} for "
                        f"argument {name!r}): move them to one GPU before the call")


def _device_guard(*exempt):
    """Run the wrapper with the arguments' GPU as the current device: the launch stream (``_stream``), the workspaces
    and ``require_gpu()`` then all belong to the device the operands live on, whatever ``torch.cuda.current_device()``
    was (the reference's users do ``model.to("cuda:1")``, inference/abstract_classes.py:250-255)."""
    import functools
    import inspect

    def wrap(fn):
        names = list(inspect.signature(fn).parameters)
        skip = {names.index(e) for e in exempt if e in names} | set(exempt)

        @functools.wraps(fn)
        def guarded(*args, **kwargs):
            dev = resolve_device(args, kwargs, skip)
            if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
                return fn(*args, **kwargs)
            with torch.cuda.device(dev):
                return fn(*args, **kwargs)

        return guarded

    return wrap


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """The current device's current stream as the C ABI's ``runia_stream_t`` (torch's raw-stream query where this build has it:
    no ``torch.cuda.Stream`` object per wrapper call - 2 400 of them per harness sweep)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class _UploadCache(threading.local):
    depth = 0
    entries = None


_upload_cache = _UploadCache()
_UPLOAD_CACHE_MIN_BYTES = 1 << 20


@contextlib.contextmanager
def upload_cache():
    """While the context is open (re-entrant, per thread), ``to_device`` of a C-contiguous host ndarray of at least 1 MB is done once
    per (buffer address, shape, dtype, target dtype): later calls return the same device tensor.  For loops that hand the SAME
    host arrays to many setup / postprocess calls (``evaluation.baselines.calculate_all_baselines(device_resident=True)``: the
    training features would otherwise be uploaded by eight fits, each split by every baseline).  The arrays must not be written to
    while the context is open (the postprocessors never do: inputs are read-only by contract); the cache holds a reference to
    every array it has seen and drops everything when the outermost context closes.  The tensors are shared: read-only as well."""
    c = _upload_cache
    if c.depth == 0:
        c.entries = {}
    c.depth += 1
    try:
        yield
    finally:
        c.depth -= 1
        if c.depth == 0:
            c.entries = None


def to_device(a, dtype: torch.dtype) -> torch.Tensor:
    """Host ndarray / tensor -> contiguous device tensor of ``dtype`` (H2D copy if needed; see ``upload_cache``)."""
    dev = require_gpu()
    if isinstance(a, np.ndarray):
        c = _upload_cache
        if c.depth > 0 and a.flags.c_contiguous and a.nbytes >= _UPLOAD_CACHE_MIN_BYTES:
            key = (a.__array_interface__["data"][0], a.shape, a.dtype.str, dtype, str(dev))
            hit = c.entries.get(key)
            if hit is None:
                hit = (a, torch.from_numpy(a).to(device=dev, dtype=dtype, non_blocking=False).contiguous())
                c.entries[key] = hit
            return hit[1]
        t = torch.from_numpy(np.ascontiguousarray(a))
    elif isinstance(a, torch.Tensor):
        t = a.detach()
    else:
        t = torch.as_tensor(np.asarray(a))
    return t.to(device=dev, dtype=dtype, non_blocking=False).contiguous()


_PINNED_MIN, _PINNED_MAX = 1 << 16, 1 << 30


def to_host(t: torch.Tensor) -> np.ndarray:
    """Device tensor -> host ndarray (the ``.cpu().numpy()`` of every API that returns arrays, as the reference's do).
    Results between 64 KB and 1 GB land in page-locked memory from torch's caching host allocator and the ndarray is a
    view of it: a D2H copy into freshly allocated pageable memory runs at ~2 GB/s on this platform (first-touch page
    faults; the 41 MB of entropies of a 10 000-image batch took 19 ms, 200 x the kernels that made them), into a
    recycled pinned block at the link rate.  The block returns to the allocator's cache when the array is released."""
    if not t.is_cuda:
        return t.detach().numpy()
    t = t.detach().contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes < _PINNED_MIN or nbytes > _PINNED_MAX:
        return t.cpu().numpy()
    try:
        out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:  # locked-memory limit reached, fragmentation: the pageable copy is slower, never wrong
        return t.cpu().numpy()
    out.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return out.numpy()


# --------------------------------------------------------------------------------------
# stage wrappers: device tensors in, device tensors out, stream-ordered, no sync
# --------------------------------------------------------------------------------------
@_device_guard()
def mc_stack(x: torch.Tensor, rand: Optional[torch.Tensor], n_mc: int, drop_prob: float, block_size: int) -> torch.Tensor:
    """x [N,C,H,W] f32, rand [n_mc,H,W] (shared) or [N,n_mc,H,W] f32 -> [N*n_mc, C] f32."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    n, c, h, w = x.shape
    stride = 0
    if isinstance(rand, CounterDraws):
        rand = _explicit_counter_draws(rand, n, n_mc, h, w)
    if rand is not None:
        assert rand.is_cuda and rand.dtype == torch.float32
        rand = rand.contiguous()
        if rand.dim() == 4:
            assert rand.shape == (n, n_mc, h, w)
            stride = n_mc * h * w
        else:
            assert rand.shape == (n_mc, h, w)
    out = torch.empty((n * n_mc, c), dtype=torch.float32, device=x.device)
    table_path = n_mc >= 2 and bool(lib.runia_mc_entropy_supported(h, w, n_mc, 5)) and (x.data_ptr() % 16 == 0 or (h * w) % 4)
    if table_path:
        ws_bytes = int(lib.runia_mc_entropy_workspace_bytes(min(65535, n), h, w, n_mc))
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    done = 0
    while done < n:  # grid limit of the kernels
        m = min(65535, n - done)
        rp = None if rand is None else rand.data_ptr() + (done * stride * 4)
        if table_path:
            _check(
                lib.runia_mc_stack_table_f32(
                    x.data_ptr() + done * c * h * w * 4, rp, stride, out.data_ptr() + done * n_mc * c * 4, ws.data_ptr(),
                    ws_bytes, m, c, h, w, n_mc, float(drop_prob), int(block_size), _stream(),
                ),
                "runia_mc_stack_table_f32",
            )
            done += m
            continue
        _check(
            lib.runia_mc_stack_f32(
                x.data_ptr() + done * c * h * w * 4, rp, stride, out.data_ptr() + done * n_mc * c * 4,
                m, c, h, w, n_mc, float(drop_prob), int(block_size), _stream(),
            ),
            "runia_mc_stack_f32",
        )
        done += m
    return out


def _explicit_counter_draws(ticket: CounterDraws, n: int, n_mc: int, h: int, w: int) -> torch.Tensor:
    """Counter draws written out for the kernels that read draws from memory.  The redraw of fully dropped maps lives in
    the keep-flag kernel (fused table path: 2x2 / 4x4 / 7x7 / 8x8 maps): it cannot be honoured here, so it is refused
    rather than silently ignored."""
    if ticket.redraw_dead_layers:
        raise RuniaHipError("CounterDraws(redraw_dead_layers=True) is implemented by the keep-flag kernel of the fused "
                            "sampler + entropy path (maps of 2x2, 4x4, 7x7, 8x8); this call takes the explicit-draw kernels")
    return mc_draws(n, n_mc, h, w, ticket.seed, ticket.first_image)


def mc_draws(n: int, n_mc: int, h: int, w: int, seed: int, first_image: int = 0) -> torch.Tensor:
    """The counter generator's draws written out: [n, n_mc, h, w] f32 in [0, 1) (same values the counter entry points
    use inside the keep-flag kernel)."""
    lib = load_library()
    dev = require_gpu()
    out = torch.empty((n, n_mc, h, w), dtype=torch.float32, device=dev)
    _check(lib.runia_mc_draws_f32(out.data_ptr(), n, n_mc, h, w, int(seed) & (2**64 - 1), int(first_image), _stream()),
           "runia_mc_draws_f32")
    return out


@_device_guard()
def mc_drop_flat(x: torch.Tensor, rand: Optional[torch.Tensor], n_mc: int, drop_prob: float, block_size: int) -> torch.Tensor:
    """``layer_type="FC"/"RPN"`` form of the sampler: x [N,C,H,W] f32 -> [N*n_mc, C*H*W] f32 (no fullmean)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    n, c, h, w = x.shape
    stride = 0
    if isinstance(rand, CounterDraws):
        rand = _explicit_counter_draws(rand, n, n_mc, h, w)
    if rand is not None:
        assert rand.is_cuda and rand.dtype == torch.float32
        rand = rand.contiguous()
        if rand.dim() == 4:
            assert rand.shape == (n, n_mc, h, w)
            stride = n_mc * h * w
        else:
            assert rand.shape == (n_mc, h, w)
    e = c * h * w
    out = torch.empty((n * n_mc, e), dtype=torch.float32, device=x.device)
    done = 0
    while done < n:
        m = min(65535, n - done)
        rp = None if rand is None else rand.data_ptr() + done * stride * 4
        _check(
            lib.runia_mc_drop_flat_f32(x.data_ptr() + done * e * 4, rp, stride, out.data_ptr() + done * n_mc * e * 4,
                                       m, c, h, w, n_mc, float(drop_prob), int(block_size), _stream()),
            "runia_mc_drop_flat_f32",
        )
        done += m
    return out


@_device_guard()
def map_reduce(x: torch.Tensor, h: int, w: int, mode: str) -> torch.Tensor:
    """x [..., h*w] f32 seen as maps of h x w -> ``mode="mean"``: mean over w, [maps, h]; ``mode="std"``: std over the
    rows of the per-row stds, [maps] (the reductions of ``get_mean_or_fullmean_ls_sample(., "mean")`` and
    ``get_std_ls_sample`` upstream)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.numel() % (h * w) == 0 and mode in ("mean", "std")
    x = x.contiguous()
    maps = x.numel() // (h * w)
    out = torch.empty((maps, h) if mode == "mean" else (maps,), dtype=torch.float32, device=x.device)
    _check(lib.runia_map_reduce_f32(x.data_ptr(), out.data_ptr(), maps, h, w, 0 if mode == "mean" else 1, _stream()),
           "runia_map_reduce_f32")
    return out


@_device_guard()
def kl_entropy_per_dim(z: torch.Tensor, n_mc: int, k: int, min_dist: float = 1e-5) -> torch.Tensor:
    """z [N*n_mc, D] f32 -> h [N, D] f64."""
    lib = load_library()
    require_gpu()
    assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 2
    z = z.contiguous()
    n = z.shape[0] // n_mc
    d = z.shape[1]
    h = torch.empty((n, d), dtype=torch.float64, device=z.device)
    _check(
        lib.runia_kl_entropy_per_dim_f32(z.data_ptr(), h.data_ptr(), n, n_mc, d, k, min_dist, _stream()),
        "runia_kl_entropy_per_dim_f32",
    )
    return h


@_device_guard()
def kl_entropy_joint(z: torch.Tensor, n_mc: int, k: int, min_dist: float = 1e-5) -> torch.Tensor:
    """z [N*n_mc, D] f32 -> h_mvn [N] f64."""
    lib = load_library()
    require_gpu()
    assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 2
    z = z.contiguous()
    n = z.shape[0] // n_mc
    d = z.shape[1]
    h = torch.empty((n,), dtype=torch.float64, device=z.device)
    _check(
        lib.runia_kl_entropy_joint_f32(z.data_ptr(), h.data_ptr(), n, n_mc, d, k, min_dist, _stream()),
        "runia_kl_entropy_joint_f32",
    )
    return h


@_device_guard()
def kl_entropy_both(z: torch.Tensor, n_mc: int, k: int, min_dist: float = 1e-5):
    """z [N*n_mc, D] f32 -> (h_mvn [N] f64, h [N, D] f64): both outputs of ``get_dl_h_z`` from one read of the samples
    (``runia_kl_entropy_both_f32``; same bits as :func:`kl_entropy_joint` and :func:`kl_entropy_per_dim`)."""
    lib = load_library()
    require_gpu()
    assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 2
    z = z.contiguous()
    n = z.shape[0] // n_mc
    d = z.shape[1]
    h_mvn = torch.empty((n,), dtype=torch.float64, device=z.device)
    h = torch.empty((n, d), dtype=torch.float64, device=z.device)
    _check(lib.runia_kl_entropy_both_f32(z.data_ptr(), h_mvn.data_ptr(), h.data_ptr(), n, n_mc, d, k, min_dist, _stream()),
           "runia_kl_entropy_both_f32")
    return h_mvn, h


@_device_guard()
def pack_weights(b: torch.Tensor) -> torch.Tensor:
    """B [K, n] f64 (device) -> fragment-ordered copy for the f64 MFMA kernels."""
    lib = load_library()
    require_gpu()
    assert b.is_cuda and b.dtype == torch.float64 and b.dim() == 2
    b = b.contiguous()
    k, n = b.shape
    nbytes = lib.runia_packed_weights_bytes(k, n)
    packed = torch.empty((nbytes // 8,), dtype=torch.float64, device=b.device)
    _check(lib.runia_pack_weights_f64(b.data_ptr(), n, k, n, packed.data_ptr(), _stream()), "runia_pack_weights_f64")
    return packed


@_device_guard()
def pca_transform(x: torch.Tensor, packed_ct: torch.Tensor, bias: torch.Tensor, scale: Optional[torch.Tensor], n: int) -> torch.Tensor:
    """x [N, D] f64/f32 -> y [N, n] f64."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.float64)
    x = x.contiguous()
    nrow, d = x.shape
    y = torch.empty((nrow, n), dtype=torch.float64, device=x.device)
    fn = lib.runia_pca_transform_f32in if x.dtype == torch.float32 else lib.runia_pca_transform_f64
    _check(
        fn(x.data_ptr(), packed_ct.data_ptr(), bias.data_ptr(), _ptr(scale), y.data_ptr(), nrow, d, n,
           0 if scale is None else 1, _stream()),
        "runia_pca_transform",
    )
    return y


@_device_guard()
def md_score(x: torch.Tensor, mean: torch.Tensor, packed_p: torch.Tensor) -> torch.Tensor:
    """x [N, n] (f64 or f32), mean [n] (f64 or f32) -> score [N] f64 = -(x-mean) P (x-mean)^T,
    with ``x - mean`` formed under NumPy's dtype rules (f32 only when both are f32)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.float64)
    x = x.contiguous()
    nrow, n = x.shape
    if x.dtype == torch.float64:
        mean = mean.to(torch.float64)
        fn = lib.runia_md_score_ws_f64
    elif mean.dtype == torch.float32:
        fn = lib.runia_md_score_ws_f32
    else:
        mean = mean.to(torch.float64)
        fn = lib.runia_md_score_ws_f32x_f64mean
    mean = mean.contiguous()
    s = torch.empty((nrow,), dtype=torch.float64, device=x.device)
    # few rows of wide features: column blocks on separate workgroups + a replay launch (same bits, see runia_hip.h)
    ws_bytes = int(lib.runia_md_score_workspace_bytes(nrow, n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    _check(fn(x.data_ptr(), mean.data_ptr(), packed_p.data_ptr(), s.data_ptr(), _ptr(ws), ws_bytes, nrow, n, _stream()),
           "runia_md_score_ws")
    return s


@_device_guard()
def md_score_tril(x: torch.Tensor, mean: torch.Tensor, packed_wt: torch.Tensor) -> torch.Tensor:
    """``md_score`` from the triangular factor of the precision (``runia_md_score_tril_*``): precision = W^T W with W lower
    triangular, ``packed_wt = pack_weights(W^T)``; score [N] f64 = -|| W (x - mean) ||^2, same centring rules as ``md_score``."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.float64)
    x = x.contiguous()
    nrow, n = x.shape
    if x.dtype == torch.float64:
        mean = mean.to(torch.float64)
        fn = lib.runia_md_score_tril_f64
    elif mean.dtype == torch.float32:
        fn = lib.runia_md_score_tril_f32
    else:
        mean = mean.to(torch.float64)
        fn = lib.runia_md_score_tril_f32x_f64mean
    mean = mean.contiguous()
    s = torch.empty((nrow,), dtype=torch.float64, device=x.device)
    ws_bytes = int(lib.runia_md_score_workspace_bytes(nrow, n))  # few rows of wide features: column blocks + replay (same bits)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    _check(fn(x.data_ptr(), mean.data_ptr(), packed_wt.data_ptr(), s.data_ptr(), _ptr(ws), ws_bytes, nrow, n, _stream()),
           "runia_md_score_tril")
    return s


@_device_guard()
def mahalanobis_score(x: torch.Tensor, class_mean: torch.Tensor, packed_p: torch.Tensor, mu_p: torch.Tensor,
                      class_loop: bool = False, split: bool = True) -> torch.Tensor:
    """x [N, D], class_mean [C, D] (both f32 or both f64) -> score [N] f64.  ``class_loop=True`` hands over the small
    workspace only (more than 16 classes then take the per-class loop instead of the matrix-core form; tests).
    ``split=False`` (up to 16 classes; tests and measurements) hands over NO workspace: the entry point then keeps the
    one-launch form instead of the column-split launches - the same bits."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dim() == 2 and x.dtype == class_mean.dtype
    x = x.contiguous()
    class_mean = class_mean.contiguous()
    nrow, d = x.shape
    c = class_mean.shape[0]
    s = torch.empty((nrow,), dtype=torch.float64, device=x.device)
    ws_bytes = lib.runia_mahalanobis_workspace_bytes(nrow, d) if class_loop else lib.runia_mahalanobis_workspace_bytes_classes(nrow, d, c)
    ws = torch.empty((max(ws_bytes, 8) // 8,), dtype=torch.float64, device=x.device)
    no_ws = (not split) and c <= 16
    fn = lib.runia_mahalanobis_score_f32 if x.dtype == torch.float32 else lib.runia_mahalanobis_score_f64
    _check(
        fn(x.data_ptr(), class_mean.data_ptr(), packed_p.data_ptr(), mu_p.data_ptr(), s.data_ptr(),
           None if no_ws else ws.data_ptr(), 0 if no_ws else ws_bytes, nrow, d, c, _stream()),
        "runia_mahalanobis_score",
    )
    return s


@_device_guard()
def row_lse_msp(logits: torch.Tensor, want_lse: bool = True, want_msp: bool = False):
    """logits [N, C] f32 -> (lse [N] f32 | None, msp [N] f32 | None)."""
    lib = load_library()
    require_gpu()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2
    logits = logits.contiguous()
    n, c = logits.shape
    lse = torch.empty((n,), dtype=torch.float32, device=logits.device) if want_lse else None
    msp = torch.empty((n,), dtype=torch.float32, device=logits.device) if want_msp else None
    _check(lib.runia_row_lse_msp_f32(logits.data_ptr(), _ptr(lse), _ptr(msp), n, c, _stream()), "runia_row_lse_msp_f32")
    return lse, msp


@_device_guard()
def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    x = x.contiguous()
    y = torch.empty_like(x)
    _check(lib.runia_l2_normalize_f32(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], _stream()), "runia_l2_normalize_f32")
    return y


@_device_guard()
def knn_prepare_bank(bank: torch.Tensor) -> torch.Tensor:
    """Once per bank: squared row norms, their maximum and (banks the bf16 kernel can take) the bf16 pieces, as one
    device buffer for ``knn_kth(..., state=)``.  A deployed index scores many batches against the same bank; each call
    then skips the bank passes (50 000 x 2048: 0.4 ms)."""
    lib = load_library()
    require_gpu()
    assert bank.is_cuda and bank.dtype == torch.float32 and bank.dim() == 2 and bank.is_contiguous()
    m, d = bank.shape
    nbytes = int(lib.runia_knn_bank_state_bytes(m, d))
    state = torch.empty((max(nbytes, 16) + 15) // 16 * 4, dtype=torch.float32, device=bank.device)  # (16-byte granules)
    _check(lib.runia_knn_prepare_bank_f32(bank.data_ptr(), state.data_ptr(), nbytes, m, d, _stream()), "runia_knn_prepare_bank_f32")
    return state


@_device_guard()
def knn_kth(q: torch.Tensor, bank: torch.Tensor, k: int, state: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [N, D], bank [M, D] (both L2-normalised f32) -> -(k-th smallest squared L2) [N] f32.
    ``state``: ``knn_prepare_bank(bank)`` of the same bank (same scores, without the per-call bank passes)."""
    lib = load_library()
    require_gpu()
    assert q.is_cuda and bank.is_cuda and q.dtype == torch.float32 and bank.dtype == torch.float32
    q = q.contiguous()
    bank = bank.contiguous()
    n, d = q.shape
    m = bank.shape[0]
    s = torch.empty((n,), dtype=torch.float32, device=q.device)
    f32_only = not _config.knn_bf16_candidates  # (hand the entry point the f32 kernel's workspace: it keeps that kernel)
    qc = min(n, 8192, max(256, (1 << 31) // (4 * m)))  # the f32 kernel's chunk of distances (+ |q|^2 ...)
    if state is not None and m > 0 and n > 0:
        ws_bytes = (qc * m + qc) * 4 if f32_only else int(lib.runia_knn_prepared_workspace_bytes(n, m, d, k))
        ws = torch.empty((max(ws_bytes, 4) // 4,), dtype=torch.float32, device=q.device)
        _check(
            lib.runia_knn_kth_prepared_f32(q.data_ptr(), bank.data_ptr(), state.data_ptr(), state.numel() * 4, s.data_ptr(),
                                           ws.data_ptr(), ws_bytes, n, m, d, int(k), _stream()),
            "runia_knn_kth_prepared_f32",
        )
        return s
    ws_bytes = lib.runia_knn_workspace_bytes(n, m, d, k)
    if f32_only and lib.runia_knn_piece_products(n, m, d) > 0:
        # the f32 kernel's workspace (one chunk of distances, |q|^2, |b|^2, max |b|^2): the entry point then keeps that kernel
        ws_bytes = (qc * m + qc + m + 4) * 4
    ws = torch.empty((max(ws_bytes, 4) // 4,), dtype=torch.float32, device=q.device)
    _check(
        lib.runia_knn_kth_f32(q.data_ptr(), bank.data_ptr(), s.data_ptr(), ws.data_ptr(), ws_bytes, n, m, d, int(k), _stream()),
        "runia_knn_kth_f32",
    )
    return s


@_device_guard()
def row_sqnorm(x: torch.Tensor) -> torch.Tensor:
    """x [N, D] f64 -> squared row norms [N] f64."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float64 and x.dim() == 2
    x = x.contiguous()
    out = torch.empty((x.shape[0],), dtype=torch.float64, device=x.device)
    _check(lib.runia_row_sqnorm_f64(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _stream()), "runia_row_sqnorm_f64")
    return out


@_device_guard()
def kde_pack_train(train: torch.Tensor):
    """Setup-time state of ``kde_score_packed``: (pack(train_c^T), squared row norms of train_c, M, D, mean) with
    train_c = train - mean(train).  Distances are translation invariant; centring keeps |x|^2 + |t|^2 - 2 x.t
    well conditioned when the embeddings sit far from the origin."""
    assert train.is_cuda and train.dtype == torch.float64 and train.dim() == 2
    mean = train.mean(dim=0)
    tc = (train - mean).contiguous()
    return pack_weights(tc.t().contiguous()), row_sqnorm(tc), int(train.shape[0]), int(train.shape[1]), mean


@_device_guard()
def kde_score_packed(state, x: torch.Tensor, bandwidth: float = 1.0) -> torch.Tensor:
    """Gaussian-KDE log-density [N] f64 of x [N, D] f64 against a packed training set (matrix-core path)."""
    lib = load_library()
    require_gpu()
    packed, tn, m, d, mean = state
    assert x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.shape[1] == d
    x = (x - mean).contiguous()
    n = x.shape[0]
    out = torch.empty((n,), dtype=torch.float64, device=x.device)
    ws_bytes = int(lib.runia_kde_workspace_bytes(n, m))  # query norms (+ the values of the column-split launch on few rows)
    ws = torch.empty((max(ws_bytes // 8, 1),), dtype=torch.float64, device=x.device)
    _check(lib.runia_kde_score_packed_f64(packed.data_ptr(), tn.data_ptr(), x.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                          ws_bytes, m, n, d, float(bandwidth), _stream()),
           "runia_kde_score_packed_f64")
    return out


@_device_guard()
def kde_score(train: torch.Tensor, x: torch.Tensor, bandwidth: float = 1.0) -> torch.Tensor:
    """train [M, D] f64, x [N, D] f64 -> gaussian-KDE log-density [N] f64."""
    lib = load_library()
    require_gpu()
    assert train.is_cuda and x.is_cuda and train.dtype == torch.float64 and x.dtype == torch.float64
    train = train.contiguous()
    x = x.contiguous()
    m, d = train.shape
    n = x.shape[0]
    s = torch.empty((n,), dtype=torch.float64, device=x.device)
    _check(
        lib.runia_kde_score_f64(train.data_ptr(), x.data_ptr(), s.data_ptr(), m, n, d, float(bandwidth), _stream()),
        "runia_kde_score_f64",
    )
    return s


_TIMED_EVENT_POOL: dict = {}  # device index -> event pairs created (recorded once) on that device


def reserve_timed_events(n: int) -> None:
    """Create ``n`` event pairs for :func:`_timed_launch_events` now, on the current device (an event exists only once it has
    been recorded: two marker packets on the stream per pair), so that a bracketed launch inside a timed region costs no record of
    its own.  The pool is kept per device: an event belongs to the device it was first recorded on."""
    pool = _TIMED_EVENT_POOL.setdefault(torch.cuda.current_device(), [])
    for _ in range(int(n)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pool.append((e0, e1))


def _timed_launch_events():
    """An event pair attached to the NEXT timed launch site (``runia_time_next_launch``): the events then hold the kernel's
    own start / end timestamps (what rocprofv3's kernel trace reports).  A pair recorded around the launch on the stream also
    counts the dispatch gap behind the previous kernel: K1 read 117.3 us that way against 109.5 us in the trace of the same run.
    Use :func:`_timed_call` around the entry point: it disarms the pair if the call returns before its launch site."""
    pool = _TIMED_EVENT_POOL.setdefault(torch.cuda.current_device(), [])
    if not pool:
        reserve_timed_events(1)
    e0, e1 = pool.pop()
    _check(load_library().runia_time_next_launch(e0.cuda_event, e1.cuda_event), "runia_time_next_launch")
    return e0, e1


def _timed_call(fn, what: str):
    """Arm an event pair, run the entry point ``fn()`` (-> return code), and ALWAYS disarm afterwards: an entry point that
    returns before its timed launch site (unsupported shape, short workspace) must not leave the pair to the thread's next,
    unrelated launch.  Returns the pair; raises as ``_check`` does (the pair of a failed call never received timestamps and is
    dropped)."""
    e0, e1 = _timed_launch_events()
    try:
        rc = fn()
    finally:
        load_library().runia_time_next_launch(None, None)
    _check(rc, what)
    return e0, e1


def mc_entropy_supported(h: int, w: int, n_mc: int, k: int) -> bool:
    return bool(load_library().runia_mc_entropy_supported(int(h), int(w), int(n_mc), int(k)))


@_device_guard()
def mc_mask_table(rand: Union[torch.Tensor, CounterDraws, None], n: int, h: int, w: int, n_mc: int, drop_prob: float,
                  block_size: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K0 alone: the keep-flag table of a batch of ``n`` <= 65 535 images (``runia_mc_mask_table_f32`` / its counter
    form) into ``out`` (uint8 workspace of ``runia_mc_entropy_workspace_bytes``) on the current stream.  Pass the result
    as ``table=`` to :func:`mc_entropy`: a caller that knows the next batch's draws builds its table on a side stream
    under the previous batch's kernels (``LaREMPipeline.prepare_draws``)."""
    lib = load_library()
    dev = require_gpu()
    assert 0 < n <= 65535
    ws_bytes = int(lib.runia_mc_entropy_workspace_bytes(n, h, w, n_mc))
    if out is None:
        out = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    assert out.is_cuda and out.dtype == torch.uint8 and out.numel() >= ws_bytes
    if isinstance(rand, CounterDraws):
        _check(lib.runia_mc_mask_table_counter_f32(int(rand.seed) & (2**64 - 1), int(rand.first_image), out.data_ptr(),
                                                   ws_bytes, n, h, w, n_mc, float(drop_prob), int(block_size),
                                                   int(bool(rand.redraw_dead_layers)), _stream()),
               "runia_mc_mask_table_counter_f32")
        return out
    stride = 0
    if rand is not None:
        assert rand.is_cuda and rand.dtype == torch.float32 and rand.is_contiguous()
        if rand.dim() == 4:
            assert rand.shape == (n, n_mc, h, w)
            stride = n_mc * h * w
        else:
            assert rand.shape == (n_mc, h, w)
    _check(lib.runia_mc_mask_table_f32(_ptr(rand), stride, out.data_ptr(), ws_bytes, n, h, w, n_mc,
                                       float(drop_prob) if rand is not None else 0.0, int(block_size), _stream()),
           "runia_mc_mask_table_f32")
    return out


@_device_guard()
def mc_entropy(x: torch.Tensor, rand: Union[torch.Tensor, CounterDraws, None], n_mc: int, drop_prob: float, block_size: int, k: int,
               min_dist: float = 1e-5, want_samples: bool = False, out: Optional[torch.Tensor] = None,
               kernel_events: Optional[list] = None, zero_fill: Optional[torch.Tensor] = None,
               table: Optional[torch.Tensor] = None):
    """Fused sampler + entropy: x [N,C,H,W] f32 (+ draws) -> h [N, C] f64 (and optionally the MC samples).
    ``kernel_events``: if a list, (start, end) HIP event pairs attached to the dispatch of the sampler + entropy launch
    (the kernel's own start / end timestamps; the keep-flag table launch before it is left out) are appended - bench.py
    times the dominant kernel with it.
    ``zero_fill``: optional [N] f64 tensor cleared by the launch (the accumulator of ``proj_sq_accumulate``).
    ``table``: the batch's keep-flag table built earlier by :func:`mc_mask_table` (``rand`` is then ignored)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    n, c, hh, ww = x.shape
    if table is not None:
        h = torch.empty((n, c), dtype=torch.float64, device=x.device) if out is None else out
        assert h.is_cuda and h.dtype == torch.float64 and h.shape == (n, c) and h.is_contiguous()
        z = torch.empty((n * n_mc, c), dtype=torch.float32, device=x.device) if want_samples else None
        ws_bytes = int(lib.runia_mc_entropy_workspace_bytes(n, hh, ww, n_mc))
        assert table.is_cuda and table.numel() >= ws_bytes and n <= 65535
        if zero_fill is not None:
            assert zero_fill.is_cuda and zero_fill.dtype == torch.float64 and zero_fill.shape == (n,) and zero_fill.is_contiguous()
        call = lambda: lib.runia_mc_entropy_from_table_f32(x.data_ptr(), table.data_ptr(), ws_bytes, h.data_ptr(), _ptr(z),  # noqa: E731
                                                           _ptr(zero_fill), n, c, hh, ww, n_mc, int(k), float(min_dist), _stream())
        if kernel_events is not None:
            kernel_events.append(_timed_call(call, "runia_mc_entropy_from_table_f32"))
        else:
            _check(call(), "runia_mc_entropy_from_table_f32")
        return (h, z) if want_samples else h
    stride = 0
    counter = rand if isinstance(rand, CounterDraws) else None
    if counter is not None:
        rand = None
    if rand is not None:
        assert rand.is_cuda and rand.dtype == torch.float32
        rand = rand.contiguous()
        if rand.dim() == 4:
            assert rand.shape == (n, n_mc, hh, ww)
            stride = n_mc * hh * ww
        else:
            assert rand.shape == (n_mc, hh, ww)
    if out is None:
        h = torch.empty((n, c), dtype=torch.float64, device=x.device)
    else:
        assert out.is_cuda and out.dtype == torch.float64 and out.shape == (n, c) and out.is_contiguous()
        h = out
    z = torch.empty((n * n_mc, c), dtype=torch.float32, device=x.device) if want_samples else None
    if zero_fill is not None:
        assert zero_fill.is_cuda and zero_fill.dtype == torch.float64 and zero_fill.shape == (n,) and zero_fill.is_contiguous()
    ws_bytes = int(lib.runia_mc_entropy_workspace_bytes(min(65535, n), hh, ww, n_mc))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)  # stream-ordered: reused per slice
    done = 0
    while done < n:
        m = min(65535, n - done)
        rp = None if rand is None else rand.data_ptr() + done * stride * 4
        zp = None if z is None else z.data_ptr() + done * n_mc * c * 4
        zf = None if zero_fill is None else zero_fill.data_ptr() + done * 8
        if counter is not None and kernel_events is None:
            _check(
                lib.runia_mc_entropy_counter_f32(x.data_ptr() + done * c * hh * ww * 4, int(counter.seed) & (2**64 - 1),
                                                 int(counter.first_image) + done, h.data_ptr() + done * c * 8, zp, zf,
                                                 ws.data_ptr(), ws_bytes, m, c, hh, ww, n_mc, float(drop_prob),
                                                 int(block_size), int(k), float(min_dist),
                                                 int(bool(counter.redraw_dead_layers)), _stream()),
                "runia_mc_entropy_counter_f32",
            )
        elif kernel_events is None:
            _check(
                lib.runia_mc_entropy_f32(x.data_ptr() + done * c * hh * ww * 4, rp, stride, h.data_ptr() + done * c * 8,
                                         zp, zf, ws.data_ptr(), ws_bytes, m, c, hh, ww, n_mc, float(drop_prob),
                                         int(block_size), int(k), float(min_dist), _stream()),
                "runia_mc_entropy_f32",
            )
        else:
            if counter is not None:
                _check(
                    lib.runia_mc_mask_table_counter_f32(int(counter.seed) & (2**64 - 1), int(counter.first_image) + done,
                                                        ws.data_ptr(), ws_bytes, m, hh, ww, n_mc, float(drop_prob),
                                                        int(block_size), int(bool(counter.redraw_dead_layers)), _stream()),
                    "runia_mc_mask_table_counter_f32",
                )
            else:
                _check(
                    lib.runia_mc_mask_table_f32(rp, stride, ws.data_ptr(), ws_bytes, m, hh, ww, n_mc, float(drop_prob),
                                                int(block_size), _stream()),
                    "runia_mc_mask_table_f32",
                )
            kernel_events.append(_timed_call(
                lambda: lib.runia_mc_entropy_from_table_f32(x.data_ptr() + done * c * hh * ww * 4, ws.data_ptr(), ws_bytes,
                                                            h.data_ptr() + done * c * 8, zp, zf, m, c, hh, ww, n_mc, int(k),
                                                            float(min_dist), _stream()),
                "runia_mc_entropy_from_table_f32"))
        done += m
    return (h, z) if want_samples else h


@_device_guard()
def proj_sq_accumulate(h: torch.Tensor, packed_m: torch.Tensor, c: torch.Tensor, r: int, out: torch.Tensor) -> torch.Tensor:
    """``out`` [N] f64 += -|| M h + c ||^2 where ``out`` was zeroed earlier on the stream (``mc_entropy(zero_fill=out)``):
    the score of ``proj_sq_score`` bit for bit, without its workspace and combine launch."""
    lib = load_library()
    require_gpu()
    assert h.is_cuda and h.dtype == torch.float64 and h.dim() == 2
    h = h.contiguous()
    nrow, d = h.shape
    assert out.is_cuda and out.dtype == torch.float64 and out.shape == (nrow,) and out.is_contiguous()
    _check(lib.runia_proj_sq_accumulate_f64(h.data_ptr(), packed_m.data_ptr(), c.data_ptr(), out.data_ptr(), nrow, d, int(r),
                                            _stream()),
           "runia_proj_sq_accumulate_f64")
    return out


@_device_guard()
def pca_md_score(h: torch.Tensor, packed_ct: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                 scale: Optional[torch.Tensor], md_mean: torch.Tensor, packed_p: torch.Tensor, n: int,
                 want_projection: bool = False, out: Optional[torch.Tensor] = None):
    """Fused PCA transform + LaREM score: h [N, D] f64 -> score [N] f64 (projected rows stay on chip)."""
    lib = load_library()
    require_gpu()
    assert h.is_cuda and h.dtype == torch.float64 and h.dim() == 2
    h = h.contiguous()
    nrow, d = h.shape
    if out is None:
        s = torch.empty((nrow,), dtype=torch.float64, device=h.device)
    else:
        assert out.is_cuda and out.dtype == torch.float64 and out.shape == (nrow,) and out.is_contiguous()
        s = out
    y = torch.empty((nrow, n), dtype=torch.float64, device=h.device) if want_projection else None
    _check(
        lib.runia_pca_md_score_f64(h.data_ptr(), _ptr(packed_ct), _ptr(bias), _ptr(scale), md_mean.data_ptr(),
                                   packed_p.data_ptr(), s.data_ptr(), _ptr(y), nrow, d, n, _stream()),
        "runia_pca_md_score_f64",
    )
    return (s, y) if want_projection else s


@_device_guard()
def covariance(x: torch.Tensor):
    """x [N, D] f64/f32 (device) -> (mean [D] f64, cov [D, D] f64) = np.cov(x.T, bias=1) with its column means."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.float64)
    x = x.contiguous()
    n, d = x.shape
    mean = torch.empty((d,), dtype=torch.float64, device=x.device)
    cov = torch.empty((d, d), dtype=torch.float64, device=x.device)
    ws_bytes = lib.runia_covariance_workspace_bytes(n, d)
    ws = torch.empty((max(ws_bytes, 8) // 8,), dtype=torch.float64, device=x.device)
    fn = lib.runia_covariance_f32in if x.dtype == torch.float32 else lib.runia_covariance_f64
    _check(fn(x.data_ptr(), mean.data_ptr(), cov.data_ptr(), ws.data_ptr(), ws_bytes, n, d, _stream()), "runia_covariance")
    return mean, cov


@_device_guard()
def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], clip_max: float = float("inf")) -> torch.Tensor:
    """logits [N, C] = min(x, clip_max) @ w.T + bias  (x [N, D], w [C, D], all f32 on the device)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and w.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.shape[1] == w.shape[1]
    x, w = x.contiguous(), w.contiguous()
    n, d = x.shape
    c = w.shape[0]
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _check(lib.runia_linear_f32(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), n, d, c, float(clip_max), _stream()),
           "runia_linear_f32")
    return out


@_device_guard()
def ash_s(x: torch.Tensor, percentile: int) -> torch.Tensor:
    """ASH-S of 2-D activations (``ash_s_linear_layer``): rows of up to 4 096 features in registers (wave per row), longer
    rows through the radix-select kernel (workgroup per row)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    x = x.contiguous()
    y = torch.empty_like(x)
    if x.shape[1] <= 4096:
        _check(lib.runia_ash_s_f32(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], int(percentile), _stream()), "runia_ash_s_f32")
    else:
        _check(lib.runia_ash_s_rows_f32(x.data_ptr(), y.data_ptr(), None, x.shape[0], x.shape[1], int(percentile), 1, _stream()),
               "runia_ash_s_rows_f32")
    return y


@_device_guard()
def ash_s_conv(x: torch.Tensor, percentile: int, prune_in_place: bool = True) -> torch.Tensor:
    """ASH-S of (B, C, H, W) maps (``ash_s_conv_layer``): per sample the k largest of its C*H*W activations are kept and
    the sample is multiplied by exp(sum / kept sum).  ``prune_in_place``: ``x`` itself is left pruned, as the
    reference's ``view`` + ``scatter_`` leaves its argument (x must then be contiguous)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    assert x.is_contiguous() or not prune_in_place
    xc = x.contiguous()
    b = xc.shape[0]
    d = xc.numel() // max(b, 1)
    y = torch.empty_like(xc)
    _check(lib.runia_ash_s_rows_f32(xc.data_ptr(), y.data_ptr(), xc.data_ptr() if prune_in_place else None, b, d, int(percentile),
                                    0, _stream()), "runia_ash_s_rows_f32")
    return y


@_device_guard()
def gen_entropy(probs: torch.Tensor, gamma: float, m: int) -> torch.Tensor:
    """``generalized_entropy(probs, gamma, M)`` on rows that already are probabilities -> [N] f32."""
    lib = load_library()
    require_gpu()
    assert probs.is_cuda and probs.dtype == torch.float32 and probs.dim() == 2
    probs = probs.contiguous()
    s = torch.empty((probs.shape[0],), dtype=torch.float32, device=probs.device)
    _check(lib.runia_gen_entropy_f32(probs.data_ptr(), s.data_ptr(), probs.shape[0], probs.shape[1], int(m), float(gamma), _stream()),
           "runia_gen_entropy_f32")
    return s


@_device_guard()
def mcd_uncertainty(logits: torch.Tensor, n_mc: int, want_probs: bool = False):
    """logits [N * n_mc, C] f32 (an image's MC rows consecutive) -> (pred_h [N], mi [N], softmax rows or None): the
    predictive entropy of the mean distribution and the mutual information, one launch."""
    lib = load_library()
    require_gpu()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.shape[0] % n_mc == 0
    logits = logits.contiguous()
    n, c = logits.shape[0] // n_mc, logits.shape[1]
    ph = torch.empty((n,), dtype=torch.float32, device=logits.device)
    mi = torch.empty((n,), dtype=torch.float32, device=logits.device)
    probs = torch.empty_like(logits) if want_probs else None
    _check(lib.runia_mcd_uncertainty_f32(logits.data_ptr(), _ptr(probs), ph.data_ptr(), mi.data_ptr(), n, int(n_mc), c, _stream()),
           "runia_mcd_uncertainty_f32")
    return ph, mi, probs


@_device_guard()
def gen_score(logits: torch.Tensor, gamma: float, m: int) -> torch.Tensor:
    lib = load_library()
    require_gpu()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2
    logits = logits.contiguous()
    s = torch.empty((logits.shape[0],), dtype=torch.float32, device=logits.device)
    _check(lib.runia_gen_score_f32(logits.data_ptr(), s.data_ptr(), logits.shape[0], logits.shape[1], int(m), float(gamma), _stream()),
           "runia_gen_score_f32")
    return s


@_device_guard()
def proj_norm(x: torch.Tensor, u: torch.Tensor, packed_ns: torch.Tensor, n: int) -> torch.Tensor:
    """|| (x - u) @ NS ||_2 per row -> [N] f64 (x, u both f32 or both f64)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dim() == 2 and x.dtype == u.dtype and x.dtype in (torch.float32, torch.float64)
    x, u = x.contiguous(), u.contiguous()
    nrow, d = x.shape
    out = torch.empty((nrow,), dtype=torch.float64, device=x.device)
    fn = lib.runia_proj_norm_f32 if x.dtype == torch.float32 else lib.runia_proj_norm_f64
    _check(fn(x.data_ptr(), u.data_ptr(), packed_ns.data_ptr(), out.data_ptr(), nrow, d, int(n), _stream()), "runia_proj_norm")
    return out


@_device_guard()
def proj_sq_score(h: torch.Tensor, packed_m: torch.Tensor, c: torch.Tensor, r: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """score [N] = -|| M h + c ||^2 (h [N, D] f64, packed_m = pack(M.T), c [r])."""
    lib = load_library()
    require_gpu()
    assert h.is_cuda and h.dtype == torch.float64 and h.dim() == 2
    h = h.contiguous()
    nrow, d = h.shape
    s = torch.empty((nrow,), dtype=torch.float64, device=h.device) if out is None else out
    ws_bytes = int(lib.runia_proj_sq_workspace_bytes(nrow))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=h.device)
    _check(lib.runia_proj_sq_score_f64(h.data_ptr(), packed_m.data_ptr(), c.data_ptr(), s.data_ptr(), ws.data_ptr(),
                                       ws_bytes, nrow, d, int(r), _stream()),
           "runia_proj_sq_score_f64")
    return s


@_device_guard()
def ood_metrics(ind_scores: torch.Tensor, ood_scores: torch.Tensor) -> torch.Tensor:
    """Device scores (both f32 or both f64) -> device tensor [3] f64 = (auroc, fpr@95, aupr), InD = positive class
    (``get_auroc_results`` of the reference, evaluation/metrics.py:37-100).  Stream-ordered, no synchronisation."""
    lib = load_library()
    require_gpu()
    assert ind_scores.is_cuda and ood_scores.is_cuda and ind_scores.dtype == ood_scores.dtype
    assert ind_scores.dtype in (torch.float32, torch.float64)
    a, b = ind_scores.reshape(-1).contiguous(), ood_scores.reshape(-1).contiguous()
    n = a.numel() + b.numel()
    ws_bytes = int(lib.runia_ood_metrics_workspace_bytes(n))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=a.device)
    off = (-ws.data_ptr()) % 256
    out = torch.empty(3, dtype=torch.float64, device=a.device)
    fn = lib.runia_ood_metrics_f64 if a.dtype == torch.float64 else lib.runia_ood_metrics_f32
    _check(fn(a.data_ptr(), a.numel(), b.data_ptr(), b.numel(), out.data_ptr(), ws.data_ptr() + off, ws_bytes, _stream()),
           "runia_ood_metrics")
    return out


@_device_guard()
def ood_clf_curve(ind_scores: torch.Tensor, ood_scores: torch.Tensor):
    """Device scores (both f32 or both f64) -> ``(metrics [3] f64 device, tps [runs] int64 host, fps [runs] int64 host)``:
    torchmetrics' ``_binary_clf_curve`` (cumulative true / false positives at the end of every run of equal scores,
    descending) from the device sort + scans; only the compacted curve leaves the device (one synchronisation)."""
    lib = load_library()
    require_gpu()
    assert ind_scores.is_cuda and ood_scores.is_cuda and ind_scores.dtype == ood_scores.dtype
    assert ind_scores.dtype in (torch.float32, torch.float64)
    a, b = ind_scores.reshape(-1).contiguous(), ood_scores.reshape(-1).contiguous()
    n = a.numel() + b.numel()
    ws_bytes = int(lib.runia_ood_metrics_workspace_bytes(n))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=a.device)
    off = (-ws.data_ptr()) % 256
    out = torch.empty(3, dtype=torch.float64, device=a.device)
    curve = torch.empty((2, n), dtype=torch.int32, device=a.device)  # u32 counts < 2^31 (n is limited to 2^31 - 1)
    n_points = torch.zeros(1, dtype=torch.int64, device=a.device)
    fn = lib.runia_ood_clf_curve_f64 if a.dtype == torch.float64 else lib.runia_ood_clf_curve_f32
    _check(fn(a.data_ptr(), a.numel(), b.data_ptr(), b.numel(), out.data_ptr(), curve[0].data_ptr(), curve[1].data_ptr(),
              n_points.data_ptr(), ws.data_ptr() + off, ws_bytes, _stream()), "runia_ood_clf_curve")
    m = int(n_points.item())
    host = to_host(curve[:, :m].contiguous()).astype(np.int64)
    return out, host[0], host[1]


@_device_guard()
def eigh(a: torch.Tensor, max_sweeps: int = 30, blocked: bool = True, info: Optional[dict] = None):
    """Symmetric eigen-decomposition on the device: a [n, n] f64 -> (eigenvalues [n] ascending, eigenvectors [n, n] as
    columns), like ``numpy.linalg.eigh``.  Cyclic Jacobi sweeps until one applies no rotation; ``blocked`` (default):
    ``runia_eigh_block_sweep_f64`` - 64 x 64 sub-problems in LDS + matrix-core updates, (n/32 - 1) x 2 launches per sweep;
    ``blocked=False``: the scalar-rotation form ``runia_eigh_sweep_f64`` (2 (n - 1) launches per sweep).
    Setup-time: reads one counter back per sweep.  ``info`` (optional dict) receives ``sweeps`` and ``rotations``."""
    lib = load_library()
    require_gpu()
    assert a.is_cuda and a.dtype == torch.float64 and a.dim() == 2 and a.shape[0] == a.shape[1]
    n = a.shape[0]
    sym = (a + a.T) * 0.5  # exactly symmetric input
    count = torch.zeros(1, dtype=torch.int32, device=a.device)
    if blocked:
        big = int(lib.runia_eigh_block_padded(n))
        work = torch.zeros((big, big), dtype=torch.float64, device=a.device)
        work[:n, :n] = sym
        ws_bytes = int(lib.runia_eigh_block_workspace_bytes(n))
        init, sweep, size = lib.runia_eigh_block_init_f64, lib.runia_eigh_block_sweep_f64, big
    else:
        work = sym.contiguous()
        ws_bytes = int(lib.runia_eigh_workspace_bytes(n))
        init, sweep, size = lib.runia_eigh_init_f64, lib.runia_eigh_sweep_f64, n
    v = torch.empty_like(work)
    ws = torch.empty(ws_bytes + 16, dtype=torch.uint8, device=a.device)
    off = (-ws.data_ptr()) % 16
    _check(init(work.data_ptr(), v.data_ptr(), size, ws.data_ptr() + off, ws_bytes, _stream()), "runia_eigh_init")
    done = 0
    for sweep_no in range(1, max_sweeps + 1):
        _check(sweep(work.data_ptr(), v.data_ptr(), size, ws.data_ptr() + off, ws_bytes, count.data_ptr(), _stream()),
               "runia_eigh_sweep")
        total = int(count.item())
        if total == done:
            break
        done = total
    else:
        raise RuniaHipError(f"the Jacobi sweeps did not converge in {max_sweeps} sweeps (n = {n})")
    if info is not None:
        info["sweeps"] = info.get("sweeps", 0) + sweep_no
        info["rotations"] = info.get("rotations", 0) + done
        info["calls"] = info.get("calls", 0) + 1
    w = torch.diagonal(work)[:n].clone()
    # ascending order: n scalars, ranked on the host (the convergence loop has synchronised already; no device sort)
    order = torch.from_numpy(np.argsort(w.cpu().numpy(), kind="stable")).to(a.device)
    return w[order], v[:n, :n][:, order].contiguous()


@_device_guard()
def matmul_f64(a: torch.Tensor, b: torch.Tensor, transpose_b: bool = False) -> torch.Tensor:
    lib = load_library()
    require_gpu()
    assert a.is_cuda and b.is_cuda and a.dtype == torch.float64 and b.dtype == torch.float64
    a, b = a.contiguous(), b.contiguous()
    m, k = a.shape
    n = b.shape[0] if transpose_b else b.shape[1]
    assert (b.shape[1] if transpose_b else b.shape[0]) == k
    c = torch.empty((m, n), dtype=torch.float64, device=a.device)
    _check(lib.runia_matmul_f64(a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, 1 if transpose_b else 0, _stream()),
           "runia_matmul_f64")
    return c


@_device_guard()
def centred_gram(e: torch.Tensor, denom: float) -> torch.Tensor:
    """e [n, H] f32 -> Gram matrix [n, n] f64 of the column-centred rows, divided by ``denom``."""
    lib = load_library()
    require_gpu()
    assert e.is_cuda and e.dtype == torch.float32 and e.dim() == 2
    e = e.contiguous()
    n, h = e.shape
    g = torch.empty((n, n), dtype=torch.float64, device=e.device)
    _check(lib.runia_centred_gram_f32(e.data_ptr(), g.data_ptr(), n, h, float(denom), _stream()), "runia_centred_gram_f32")
    return g


@_device_guard("boxes", "batch_idx")
def roi_align(x: torch.Tensor, boxes: torch.Tensor, output_size, spatial_scale: float = 1.0, sampling_ratio: int = -1,
              aligned: bool = False, batch_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``torchvision.ops.roi_align``: x [B, C, H, W] f32, boxes [K, 4] f32 (xyxy) -> [K, C, PH, PW] f32 (device)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    boxes = boxes.to(device=x.device, dtype=torch.float32).contiguous()
    b, c, h, w = x.shape
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
    k = boxes.shape[0]
    if batch_idx is not None:
        batch_idx = batch_idx.to(device=x.device, dtype=torch.int32).contiguous()
    out = torch.empty((k, c, ph, pw), dtype=torch.float32, device=x.device)
    _check(lib.runia_roi_align_f32(x.data_ptr(), boxes.data_ptr(), _ptr(batch_idx), out.data_ptr(), k, b, c, h, w, int(ph),
                                   int(pw), float(spatial_scale), int(sampling_ratio), 1 if aligned else 0, _stream()),
           "runia_roi_align_f32")
    return out


@_device_guard()
def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """x [B, C, H, W] f32 -> [B, H, W, C] f32 (a contiguous copy in channels-last order; the ROI source of
    :func:`roi_mc_entropy`: 64 channels of a wave read a bilinear tap as one contiguous run)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
    _check(lib.runia_nchw_to_nhwc_f32(x.data_ptr(), out.data_ptr(), b, c, h * w, _stream()), "runia_nchw_to_nhwc_f32")
    return out


ROI_FUSED_MAX_IMAGE_BYTES = 1 << 30  # one image's feature map behind a 32-bit buffer, with room for the offsets of outside samples


def roi_mc_entropy_supported(ph: int, pw: int, n_mc: int, k: int, sampling_ratio: int) -> bool:
    return bool(load_library().runia_roi_mc_entropy_supported(int(ph), int(pw), int(n_mc), int(k), int(sampling_ratio)))


@_device_guard("boxes", "batch_idx", "rand")
def roi_mc_entropy(x_nhwc: torch.Tensor, boxes: torch.Tensor, output_size, spatial_scale: float, sampling_ratio: int,
                   aligned: bool, rand: Union[torch.Tensor, CounterDraws, None], n_mc: int, drop_prob: float, block_size: int, k: int,
                   min_dist: float = 1e-5, batch_idx: Optional[torch.Tensor] = None, return_samples: bool = False):
    """``roi_align`` -> per-ROI ``MCSamplerModule`` -> per-dimension entropy in ONE pass from the feature map (NHWC,
    :func:`nchw_to_nhwc`): x_nhwc [B, H, W, C], boxes [K, 4] xyxy, rand [K, n_mc, PH, PW] -> h [K, C] f64.  The
    (K, C, PH, PW) tensor of ``roi_align`` is never written; same bits as ``roi_align`` + :func:`mc_entropy`.  Calls of
    more than 65 535 ROIs are cut in slices."""
    lib = load_library()
    require_gpu()
    assert x_nhwc.is_cuda and x_nhwc.dtype == torch.float32 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    b, hh, ww, c = x_nhwc.shape
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
    boxes = boxes.to(device=x_nhwc.device, dtype=torch.float32).contiguous()
    kk = boxes.shape[0]
    if batch_idx is not None:
        batch_idx = batch_idx.to(device=x_nhwc.device, dtype=torch.int32).contiguous()
    if isinstance(rand, CounterDraws):
        rand = _explicit_counter_draws(rand, kk, n_mc, ph, pw)  # (same bits as the in-kernel generator)
    if rand is not None:
        rand = rand.to(device=x_nhwc.device, dtype=torch.float32).contiguous()
        assert rand.shape == (kk, n_mc, ph, pw)
    h = torch.empty((kk, c), dtype=torch.float64, device=x_nhwc.device)
    z = torch.empty((kk * n_mc, c), dtype=torch.float32, device=x_nhwc.device) if return_samples else None
    step = 65535
    for k0 in range(0, kk, step):
        n = min(step, kk - k0)
        ws_bytes = int(lib.runia_roi_mc_entropy_workspace_bytes(n, ph, pw, n_mc, sampling_ratio))
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x_nhwc.device)
        _check(lib.runia_roi_mc_entropy_f32(
            x_nhwc.data_ptr(), boxes[k0:].data_ptr(), None if batch_idx is None else batch_idx[k0:].data_ptr(),
            None if rand is None else rand[k0:].data_ptr(), n_mc * ph * pw, h[k0:].data_ptr(),
            None if z is None else z[k0 * n_mc:].data_ptr(), ws.data_ptr(), ws_bytes, n, b, c, hh, ww, int(ph), int(pw),
            float(spatial_scale), int(sampling_ratio), 1 if aligned else 0, int(n_mc), float(drop_prob), int(block_size), int(k),
            float(min_dist), _stream()), "runia_roi_mc_entropy_f32")
    return (h, z) if return_samples else h


KDE_KERNELS = ("gaussian", "tophat", "epanechnikov", "exponential", "linear", "cosine")


@_device_guard()
def kde_score_kernel(train: torch.Tensor, x: torch.Tensor, bandwidth: float, kernel: str) -> torch.Tensor:
    """log-density of ``x`` [N, D] under a kernel density estimate on ``train`` [M, D] (both f64) for any of sklearn's
    kernels (``KDE_KERNELS``) with sklearn's normalisation -> [N] f64."""
    lib = load_library()
    require_gpu()
    assert kernel in KDE_KERNELS, f"unknown kernel {kernel!r}"
    assert train.is_cuda and x.is_cuda and train.dtype == torch.float64 and x.dtype == torch.float64
    train, x = train.contiguous(), x.contiguous()
    s = torch.empty((x.shape[0],), dtype=torch.float64, device=x.device)
    _check(lib.runia_kde_score_kernel_f64(train.data_ptr(), x.data_ptr(), s.data_ptr(), train.shape[0], x.shape[0], train.shape[1],
                                          float(bandwidth), KDE_KERNELS.index(kernel), _stream()), "runia_kde_score_kernel_f64")
    return s


@_device_guard()
def clock_probe(chain: int = 8192, device: Optional[torch.device] = None) -> torch.Tensor:
    """Queue one clock probe (``runia_clock_probe``) on the current stream -> device tensor [4] int64
    (shader-clock ticks, 100 MHz ticks, FMAs in the chain, 0).  Read it with :func:`clock_ghz` after a synchronisation."""
    lib = load_library()
    dev = require_gpu() if device is None else torch.device(device)
    with torch.cuda.device(dev):  # the probe runs on `dev`'s stream and reads `dev`'s clock, whatever device is current
        out = torch.zeros(4, dtype=torch.int64, device=dev)
        _check(lib.runia_clock_probe(out.data_ptr(), int(chain), _stream()), "runia_clock_probe")
    return out


def clock_ghz(probe: torch.Tensor) -> dict:
    """Host reading of a finished :func:`clock_probe`: the clock held (GHz), the probe's length and the cycles one
    dependent f32 FMA took."""
    t, r, n, _ = (int(v) for v in probe.cpu().tolist())
    ns = r * 10.0
    return {"ghz": round(t / ns, 4) if ns > 0 else None, "probe_us": round(ns / 1e3, 2),
            "cycles_per_dependent_fma": round(t / n, 3) if n else None}


@_device_guard()
def tril_inverse(tril: torch.Tensor) -> torch.Tensor:
    """tril [B, D, D] f64 lower-triangular (device) -> their inverses [B, D, D] (``runia_tril_inverse_f64``)."""
    lib = load_library()
    require_gpu()
    assert tril.is_cuda and tril.dtype == torch.float64 and tril.dim() == 3 and tril.shape[1] == tril.shape[2]
    tril = tril.contiguous()
    out = torch.empty_like(tril)
    _check(lib.runia_tril_inverse_f64(tril.data_ptr(), out.data_ptr(), tril.shape[0], tril.shape[1], _stream()), "runia_tril_inverse_f64")
    return out


_GMM_WORKSPACE_CAP = 1 << 30  # bytes of per-tile sums kept at once (262 144 rows x 2048 x 10 classes: 336 MB)


@_device_guard()
def gmm_log_prob(x: torch.Tensor, means: torch.Tensor, w_tril: torch.Tensor, consts: torch.Tensor, want_log_prob: bool = True,
                 want_lse: bool = False):
    """Class-wise Gaussian log densities with the inverse Cholesky factors (``runia_gmm_log_prob_f32``): x [N, D] f32, means [C, D] f32,
    w_tril [C, D, D] f32 lower triangular (= L_c^-1), consts [C] f64 -> ``(log_prob [N, C] f32 or None, lse [N] f32 or None)``."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    assert means.dtype == torch.float32 and w_tril.dtype == torch.float32 and consts.dtype == torch.float64
    n, d = x.shape
    c = means.shape[0]
    assert tuple(means.shape) == (c, d) and tuple(w_tril.shape) == (c, d, d) and tuple(consts.shape) == (c,)
    assert want_log_prob or want_lse
    x, means, w_tril, consts = x.contiguous(), means.contiguous(), w_tril.contiguous(), consts.contiguous()
    lp = torch.empty((n, c), dtype=torch.float32, device=x.device) if want_log_prob else None
    lse = torch.empty((n,), dtype=torch.float32, device=x.device) if want_lse else None
    if n == 0:
        return lp, lse
    need = int(lib.runia_gmm_log_prob_workspace_bytes(n, d, c))
    ws_bytes = min(need, max(_GMM_WORKSPACE_CAP, need // max(1, n) * 128))  # the entry point scores the rows in chunks that fit
    ws = torch.empty(((ws_bytes + 7) // 8,), dtype=torch.float64, device=x.device)
    _check(lib.runia_gmm_log_prob_f32(x.data_ptr(), means.data_ptr(), w_tril.data_ptr(), consts.data_ptr(), _ptr(lp), _ptr(lse),
                                      ws.data_ptr(), ws.numel() * 8, n, d, c, _stream()), "runia_gmm_log_prob_f32")
    return lp, lse


@_device_guard()
def kth_smallest_flat(x: torch.Tensor, ranks) -> list:
    """Exact order statistics of a float32 device tensor read as one flat array: ``sorted(x.flatten())[k]`` for every ``k`` of
    ``ranks`` (0-based), as Python floats holding float32 values.  Radix select: three histogram passes per rank
    (``runia_select_hist_f32``), one 8 KB read-back per pass.  Setup-time (synchronises).  The array must not contain NaNs (their
    keys sort above +inf; NumPy's partition puts them last as well, but its percentile then returns NaN: the caller checks)."""
    lib = load_library()
    require_gpu()
    assert x.is_cuda and x.dtype == torch.float32
    x = x.contiguous()
    n = x.numel()
    hist = torch.empty((2048,), dtype=torch.int32, device=x.device)
    out = []
    for k in ranks:
        k = int(k)
        assert 0 <= k < n
        prefix, mask = 0, 0
        for shift, width in ((21, 11), (10, 11), (0, 10)):
            _check(lib.runia_select_hist_f32(x.data_ptr(), hist.data_ptr(), n, prefix, mask, shift, _stream()), "runia_select_hist_f32")
            h = hist.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            cum = np.cumsum(h)
            b = int(np.searchsorted(cum, k, side="right"))
            assert b < 2048 and h[b] > 0, "radix select lost the rank (NaNs in the array?)"
            k -= int(cum[b - 1]) if b else 0
            prefix |= (b & ((1 << width) - 1)) << shift  # (the last pass's 11-bit digit repeats bit 10, already in the prefix)
            mask |= ((1 << width) - 1) << shift
        bits = (prefix & 0x7FFFFFFF) if (prefix & 0x80000000) else (~prefix & 0xFFFFFFFF)
        out.append(float(np.array([bits], dtype=np.uint32).view(np.float32)[0]))
    return out


@_device_guard()
def cholesky(a: torch.Tensor, jitter: float = 0.0):
    """a [B, D, D] (or [D, D]) f32 / f64 symmetric on the device -> ``(L, info)``: the lower Cholesky factors of ``a + jitter * I``
    (a new tensor; zeros above the diagonal) and ``info`` [B] int32 on the device - 0, or j + 1 where column j's pivot was not
    positive (``runia_cholesky_*``)."""
    lib = load_library()
    require_gpu()
    assert a.is_cuda and a.dtype in (torch.float32, torch.float64) and a.dim() in (2, 3) and a.shape[-1] == a.shape[-2]
    m = a.reshape(-1, a.shape[-1], a.shape[-1]).contiguous().clone()
    info = torch.empty((m.shape[0],), dtype=torch.int32, device=a.device)
    fn = lib.runia_cholesky_f32 if a.dtype == torch.float32 else lib.runia_cholesky_f64
    _check(fn(m.data_ptr(), info.data_ptr(), m.shape[0], m.shape[1], float(jitter), _stream()), "runia_cholesky")
    return m.reshape(a.shape), info
