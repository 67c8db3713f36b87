"""Entropy of MC-dropout latent samples on the GPU.

Same signatures and return shapes as the reference's
``runia_core/evaluation/entropy.py`` (``get_dl_h_z`` :41-93,
``single_image_entropy_calculation`` :20-38); the k-d-tree loops over
``entropy_estimators.continuous.get_h`` are replaced by
``runia_kl_entropy_both_f32`` (one read of the samples for both outputs of ``get_dl_h_z``) and
``runia_kl_entropy_per_dim_f32`` / ``runia_kl_entropy_joint_f32``.
"""
from __future__ import annotations

from typing import Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import _hip

__all__ = ["get_dl_h_z", "single_image_entropy_calculation", "get_dl_h_z_device", "neighbors_for"]

MIN_DIST = 1e-5  # reference passes min_dist=1e-5 at every call site


def neighbors_for(mcd_samples_nro: int) -> int:
    """k = 5, or n-1 when there are 5 samples or fewer (reference evaluation/entropy.py:66)."""
    return 5 if mcd_samples_nro > 5 else mcd_samples_nro - 1


def single_image_entropy_calculation(sample: np.ndarray, neighbors: int) -> np.ndarray:
    """Per-dimension entropy of one image's ``(n_mc, D)`` MC samples -> ``(D,)`` f64."""
    z = _hip.to_device(np.asarray(sample), torch.float32)
    h = _hip.kl_entropy_per_dim(z, z.shape[0], int(neighbors), MIN_DIST)
    return _hip.to_host(h[0])


def get_dl_h_z_device(z: Tensor, mcd_samples_nro: int, joint: bool = True):
    """Device-resident form: ``z (N*n_mc, D)`` f32 cuda -> (h_mvn (N,) f64 | None, h_z (N, D) f64)."""
    k = neighbors_for(mcd_samples_nro)
    if joint:  # both outputs from one read of the samples (runia_kl_entropy_both_f32; same bits as the two kernels)
        return _hip.kl_entropy_both(z, mcd_samples_nro, k, MIN_DIST)
    return None, _hip.kl_entropy_per_dim(z, mcd_samples_nro, k, MIN_DIST)


def get_dl_h_z(
    dl_z_samples: Union[Tensor, np.ndarray], mcd_samples_nro: int = 32, parallel_run: bool = False
) -> Tuple[np.ndarray, np.ndarray]:
    """Entropy of the latent vector from MC-dropout samples.

    Args:
        dl_z_samples: ``(N * mcd_samples_nro, D)`` samples, image-major (Tensor on any device or ndarray)
        mcd_samples_nro: MC samples per image
        parallel_run: accepted for compatibility; every image is processed in parallel on the GPU

    Returns:
        ``(N, 1)`` joint entropy and ``(N, D)`` per-dimension entropy, float64 ndarrays
    """
    if isinstance(dl_z_samples, Tensor):
        n_rows = dl_z_samples.shape[0]
    else:
        dl_z_samples = np.asarray(dl_z_samples)
        n_rows = dl_z_samples.shape[0]
        if n_rows % mcd_samples_nro != 0:
            # np.split in the reference refuses an uneven division
            raise ValueError("array split does not result in an equal division")
    z = _hip.to_device(dl_z_samples, torch.float32)
    n_img = n_rows // mcd_samples_nro
    z = z[: n_img * mcd_samples_nro]
    h_mvn, h_z = get_dl_h_z_device(z, mcd_samples_nro, joint=True)
    return _hip.to_host(h_mvn).reshape(-1, 1), _hip.to_host(h_z)
