from .abstract_classes import MCSamplerModule  # noqa: F401
from .utils import Hook, apply_dropout, get_mean_or_fullmean_ls_sample, get_std_ls_sample, get_variance_ls_sample  # noqa: F401
from .image_level import FastMCDSamplesExtractor  # noqa: F401
from .object_level import _dropblock_rois_get_entropy, _reduce_features_to_rois, roi_align  # noqa: F401
