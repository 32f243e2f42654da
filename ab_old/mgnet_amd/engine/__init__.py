from .reducer import GradReducer
from .trainer import Trainer

__all__ = ["GradReducer", "Trainer"]
