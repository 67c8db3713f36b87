from .abstract_classes import (  # noqa: F401
    InferenceModule,
    ObjectDetectionInference,
    OodPostprocessor,
    Postprocessor,
    ProbabilisticInferenceModule,
    get_baselines_thresholds,
    get_method_threshold,
    record_time,
)
from .funcs import gmm_fit, mahalanobis_postprocess, mahalanobis_preprocess, normalizer  # noqa: F401
from .image_level import LaRDInference, LaRExInference  # noqa: F401
from .pipeline import LaREMPipeline  # noqa: F401
from .postprocessors import (  # noqa: F401
    ASH,
    DDU,
    DICE,
    GEN,
    KNN,
    MSP,
    DICEReAct,
    ReAct,
    ViM,
    DetectorKDE,
    Energy,
    FlatL2Bank,
    GMMLatentSpace,
    KDELatentSpace,
    KNNLatentSpace,
    Mahalanobis,
    MDLatentSpace,
    cMDLatentSpace,
    postprocessor_input_dict,
    postprocessors_dict,
    register_postprocessor,
)
