"""runia_core_amd — MI355X-native post-hoc OOD scoring engine behind the
runia_core ``Postprocessor`` / ``LaRExInference`` / ``MCSamplerModule`` API.

Python owns the public classes and the frozen backbone (PyTorch-ROCm); every
numerical stage of the scoring hot path is a hand-written gfx950 kernel in
``librunia_hip.so`` (C ABI: ``include/runia_hip.h``).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import _hip  # noqa: F401
from .dimensionality_reduction import apply_pca_ds, apply_pca_ds_split, apply_pca_transform  # noqa: F401
from .evaluation import get_dl_h_z, single_image_entropy_calculation  # noqa: F401
from .feature_extraction import Hook, MCSamplerModule, get_mean_or_fullmean_ls_sample  # noqa: F401
from .inference import (  # noqa: F401
    LaRDInference,
    LaRExInference,
    OodPostprocessor,
    Postprocessor,
    get_baselines_thresholds,
    postprocessor_input_dict,
    postprocessors_dict,
    record_time,
)
