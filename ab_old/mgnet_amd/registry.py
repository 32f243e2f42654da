"""Registries + the `@configurable` construction protocol + ShapeSpec.  Mirrors the detectron2 surface the
reference uses: `X_REGISTRY.register()`, `.get(name)`, `build_*`, `@configurable __init__` with `from_config`."""
import functools
import inspect
from collections import namedtuple

__all__ = ["Registry", "configurable", "ShapeSpec", "META_ARCH_REGISTRY", "BACKBONE_REGISTRY", "SEM_SEG_HEADS_REGISTRY",
           "INS_EMBED_HEADS_REGISTRY", "DEPTH_HEADS_REGISTRY", "build_model", "build_backbone", "build_sem_seg_head",
           "build_ins_embed_head", "build_depth_head"]


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do_register(o.__name__, o)
                return o
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map


def _called_with_cfg(*args, **kwargs):
    from .config import CfgNode
    if len(args) and isinstance(args[0], CfgNode):
        return True
    return isinstance(kwargs.get("cfg", None), CfgNode)


def configurable(init_func):
    """`@configurable def __init__(self, *, a, b)` + `@classmethod from_config(cls, cfg, ...) -> kwargs`:
    the object can be built either from explicit arguments or from a cfg."""
    assert init_func.__name__ == "__init__"

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        from_config = getattr(type(self), "from_config", None)
        if from_config is None or not inspect.ismethod(from_config):
            raise AttributeError("Class with @configurable must have a 'from_config' classmethod.")
        if _called_with_cfg(*args, **kwargs):
            explicit = {k: v for k, v in kwargs.items() if k != "cfg"}
            sig = inspect.signature(from_config)
            cfg_kwargs = from_config(*args, **{k: v for k, v in kwargs.items() if k in sig.parameters})
            cfg_kwargs.update({k: v for k, v in explicit.items() if k not in sig.parameters})
            init_func(self, **cfg_kwargs)
        else:
            init_func(self, *args, **kwargs)

    return wrapped


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
INS_EMBED_HEADS_REGISTRY = Registry("INS_EMBED_BRANCHES")   # mg_net.py:42
DEPTH_HEADS_REGISTRY = Registry("DEPTH_BRANCHES")           # mg_net.py:47


def build_model(cfg):
    import torch
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    return model.to(torch.device(cfg.MODEL.DEVICE))


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


def build_sem_seg_head(cfg, input_shape):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)


def build_ins_embed_head(cfg, input_shape):   # mg_net.py:613-618
    return INS_EMBED_HEADS_REGISTRY.get(cfg.MODEL.INS_EMBED_HEAD.NAME)(cfg, input_shape)


def build_depth_head(cfg, input_shape):       # mg_net.py:718-723
    return DEPTH_HEADS_REGISTRY.get(cfg.MODEL.DEPTH_HEAD.NAME)(cfg, input_shape)
