from .metadata import MetadataCatalog  # noqa: F401
from .synthetic import synthetic_batch  # noqa: F401
from .target_generator import PanopticDeepLabTargetGenerator  # noqa: F401
