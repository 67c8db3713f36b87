from .entropy import get_dl_h_z, single_image_entropy_calculation  # noqa: F401
from .metrics import get_auroc_results  # noqa: F401
from .metrics import log_evaluate_postprocessors, select_and_log_best_larex  # noqa: F401
from .latent_space import log_evaluate_larex  # noqa: F401
from .baselines import baseline_name_dict, calculate_all_baselines, get_labels_from_logits, remove_latent_features  # noqa: F401
