from .depth_post_proc import get_depth_prediction  # noqa: F401
from .instance_post_proc import get_instance_predictions  # noqa: F401
from .panoptic_post_proc import get_panoptic_prediction  # noqa: F401

__all__ = ["get_panoptic_prediction", "get_instance_predictions", "get_depth_prediction"]
