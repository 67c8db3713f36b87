from .entropy import get_dl_h_z, single_image_entropy_calculation  # noqa: F401
from .metrics import get_auroc_results  # noqa: F401
