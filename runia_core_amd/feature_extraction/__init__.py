from .abstract_classes import MCSamplerModule  # noqa: F401
from .utils import Hook, get_mean_or_fullmean_ls_sample  # noqa: F401
