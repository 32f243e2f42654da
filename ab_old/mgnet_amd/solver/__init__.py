from .build import WarmupPolyLR, build_lr_scheduler, build_optimizer, get_mgnet_optimizer_params, get_module_parameters

__all__ = ["get_mgnet_optimizer_params", "get_module_parameters", "build_optimizer", "build_lr_scheduler", "WarmupPolyLR"]
