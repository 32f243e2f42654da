from .depth_evaluation import DepthEvaluator  # noqa: F401

__all__ = ["DepthEvaluator"]
