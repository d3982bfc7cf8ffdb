from .sliding_window import sliding_window_inference, window_starts, importance_map  # noqa: F401
from .convex_adam_utils import (minmax, extract_features, load_model, MINDSSC, apply_avg_pool3d,  # noqa: F401
                                smooth_merged_features, correlate, stage1_inputs)
from .instance_optimization import merge_features  # noqa: F401
