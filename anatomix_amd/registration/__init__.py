from .sliding_window import sliding_window_inference, window_starts, importance_map  # noqa: F401
from .convex_adam_utils import minmax, extract_features, load_model  # noqa: F401
