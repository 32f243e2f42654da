"""Optimizer / schedule of the MGNet recipe -- mirror of mgnet/solver/build.py:9-116 (parameter groups) and
tools/train_net.py:100-154 (Adam + full-model gradient clipping, WarmupPolyLR)."""
import itertools
import math
from typing import Any, Dict, List, Optional

import torch

from ..modeling.layers import InPlaceABNSync

_MODULES = ("backbone", "global_context", "sem_seg_head", "ins_embed_head", "depth_head", "pose_net")
_NORMS = (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm, torch.nn.modules.instancenorm._InstanceNorm,
          torch.nn.LayerNorm, torch.nn.LocalResponseNorm, InPlaceABNSync)


def get_module_parameters(module, lr=None, weight_decay=None, weight_decay_norm=None, weight_decay_bias=None):
    """One param group per tensor; conv/linear weights, their biases and norm affine params get their own decay."""
    groups: List[Dict[str, Any]] = []
    for m in module.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.ConvTranspose2d)):
            groups.append(dict(params=[m.weight], lr=lr, weight_decay=weight_decay))
            if m.bias is not None:
                groups.append(dict(params=[m.bias], lr=lr, weight_decay=weight_decay_bias))
        elif isinstance(m, _NORMS):
            groups.append(dict(params=[m.weight], lr=lr, weight_decay=weight_decay_norm))
            groups.append(dict(params=[m.bias], lr=lr, weight_decay=weight_decay_norm))
    return groups


def get_mgnet_optimizer_params(model, base_lr, weight_decay: Optional[float] = 0.0, weight_decay_norm: Optional[float] = 0.0,
                               head_lr_factor: Optional[float] = 1.0, weight_decay_bias: Optional[float] = 0.0):
    """Per-submodule learning rates: every attribute whose NAME contains "head" trains with base_lr * head_lr_factor
    (sem_seg_head, ins_embed_head, depth_head -- not pose_net / global_context), build.py:41-58; log_vars last."""
    groups: List[Dict[str, Any]] = []
    for name in _MODULES:
        sub = getattr(model, name, None)
        if sub is None:
            continue
        lr = base_lr * head_lr_factor if "head" in name else base_lr
        groups.extend(get_module_parameters(sub, lr, weight_decay, weight_decay_norm, weight_decay_bias))
    if hasattr(model, "log_vars"):
        groups.append(dict(params=model.log_vars, weight_decay=0.0, multiply_lr=False))
    return groups


def _with_full_model_clipping(optim_cls, clip_value, norm_type=2.0):
    class FullModelGradientClippingOptimizer(optim_cls):  # train_net.py:129-133
        def step(self, closure=None):
            params = itertools.chain(*[g["params"] for g in self.param_groups])
            torch.nn.utils.clip_grad_norm_(params, clip_value, norm_type=norm_type)
            return super().step(closure=closure)

    return FullModelGradientClippingOptimizer


def build_optimizer(cfg, model, reducer=None):
    """Adam/AdamW/SGD over the MGNet param groups with clip_grad_norm_(all, 0.01).
    With a flat-bucket `reducer` on a CUDA model and OPTIMIZER == "ADAM": [HIP] fused clip+Adam (solver/fused_adam.py);
    otherwise (CPU host-logic tests, SGD/AdamW) the torch optimizers."""
    s = cfg.SOLVER
    groups = get_mgnet_optimizer_params(model, base_lr=s.BASE_LR, head_lr_factor=s.HEAD_LR_FACTOR,
                                        weight_decay=s.WEIGHT_DECAY, weight_decay_norm=s.WEIGHT_DECAY_NORM)
    clip = s.CLIP_GRADIENTS
    enable = clip.ENABLED and clip.CLIP_TYPE == "full_model" and clip.CLIP_VALUE > 0.0
    wrap = (lambda c: _with_full_model_clipping(c, clip.CLIP_VALUE, clip.NORM_TYPE)) if enable else (lambda c: c)
    if reducer is not None and s.OPTIMIZER in ("ADAM", "ADAMW", "SGD"):
        # [HIP] one fused clip + step over the reducer's flat buckets for each of the three optimizers tools/train_net.py:129-154 builds
        from .fused_adam import FusedAdam
        fp16 = getattr(model, "amp_dtype", None) == torch.float16   # the reference's AMP: fp16 needs GradScaler's dynamic loss scale
        extra = dict(weight_decay=0.01) if s.OPTIMIZER == "ADAMW" else {}   # (torch.optim.AdamW's default for groups without their own)
        return FusedAdam(groups, s.BASE_LR, reducer, max_grad_norm=clip.CLIP_VALUE if enable else 0.0,
                         loss_scale=float(s.AMP.LOSS_SCALE_INIT) if fp16 else None,
                         growth_interval=int(s.AMP.LOSS_SCALE_GROWTH_INTERVAL), kind=s.OPTIMIZER, momentum=s.MOMENTUM, nesterov=s.NESTEROV,
                         **extra)
    if s.OPTIMIZER == "SGD":
        return wrap(torch.optim.SGD)(groups, s.BASE_LR, momentum=s.MOMENTUM, nesterov=s.NESTEROV)
    if s.OPTIMIZER == "ADAM":
        return wrap(torch.optim.Adam)(groups, s.BASE_LR)
    if s.OPTIMIZER == "ADAMW":
        return wrap(torch.optim.AdamW)(groups, s.BASE_LR)
    raise NotImplementedError(f"no optimizer type {s.OPTIMIZER}")


class WarmupPolyLR(torch.optim.lr_scheduler._LRScheduler):
    """detectron2.projects.deeplab WarmupPolyLR (recalled): linear warm-up from warmup_factor, then (1-it/max)^power."""

    def __init__(self, optimizer, max_iters, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear",
                 last_epoch=-1, power=0.9, constant_ending=0.0):
        self.max_iters, self.warmup_factor, self.warmup_iters = max_iters, warmup_factor, warmup_iters
        self.warmup_method, self.power, self.constant_ending = warmup_method, power, constant_ending
        super().__init__(optimizer, last_epoch)

    def _warm(self, it):
        if it >= self.warmup_iters:
            return 1.0
        if self.warmup_method == "constant":
            return self.warmup_factor
        alpha = it / self.warmup_iters
        return self.warmup_factor * (1 - alpha) + alpha

    def get_lr(self):
        w = self._warm(self.last_epoch)
        poly = math.pow(1.0 - self.last_epoch / self.max_iters, self.power)
        if self.constant_ending > 0 and w == 1.0 and poly < self.constant_ending:
            return [b * self.constant_ending for b in self.base_lrs]
        return [b * w * poly for b in self.base_lrs]


def build_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    if s.LR_SCHEDULER_NAME != "WarmupPolyLR":
        raise NotImplementedError(s.LR_SCHEDULER_NAME)
    return WarmupPolyLR(optimizer, s.MAX_ITER, warmup_factor=s.WARMUP_FACTOR, warmup_iters=s.WARMUP_ITERS,
                        warmup_method=s.WARMUP_METHOD, power=s.POLY_LR_POWER, constant_ending=s.POLY_LR_CONSTANT_ENDING)
