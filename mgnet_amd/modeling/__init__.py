from .loss import MultiViewPhotometricLoss  # noqa: F401

__all__ = ["MultiViewPhotometricLoss"]
