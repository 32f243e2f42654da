from .metadata import MetadataCatalog  # noqa: F401
from .synthetic import synthetic_batch  # noqa: F401
