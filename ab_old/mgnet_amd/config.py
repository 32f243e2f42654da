"""Config boundary: a yacs-compatible CfgNode on PyYAML + the detectron2 default keys the MGNet yamls
touch + `add_mgnet_config`, so that configs/MGNet-*.yaml of the reference load unchanged.

Mirrors mgnet/config.py:6-138 (which extends detectron2's CfgNode).  detectron2/yacs are not available in this
environment; the semantics reproduced are: attribute access, `_BASE_` inheritance relative to the yaml file,
`merge_from_file` / `merge_from_list` with "non-existent key" errors, literal_eval of string values, freeze/defrost,
clone, dump.
"""
import ast
import copy
import os

import yaml

__all__ = ["CfgNode", "CN", "get_cfg", "add_mgnet_config"]

BASE_KEY = "_BASE_"


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # ---- attribute access -----------------------------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    # ---- immutability ---------------------------------------------------------------------------
    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        out.__dict__[CfgNode.IMMUTABLE] = self.is_frozen()
        return out

    # ---- merging --------------------------------------------------------------------------------
    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return CfgNode(v)
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @staticmethod
    def _coerce(new, old, key):
        """yacs type rule: the replacement must have the type of the default (tuple<->list allowed, int->float)."""
        if old is None or new is None or type(new) is type(old):
            return new
        if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
            return type(old)(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            v = self._decode(v)
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"Type mismatch for config key: {full}")
                self[k]._merge(v, path + [k])
            else:
                self[k] = self._coerce(v, self[k], full)

    @classmethod
    def load_yaml_with_base(cls, filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = cls.load_yaml_with_base(base)

            def rec(dst, src):
                for k, v in src.items():
                    if isinstance(v, dict) and isinstance(dst.get(k), dict):
                        rec(dst[k], v)
                    else:
                        dst[k] = v

            rec(merged, cfg)
            return merged
        return cfg

    def merge_from_file(self, filename):
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable")
        self._merge(self.load_yaml_with_base(filename), [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("Override list has odd length: {}; it must be a list of pairs".format(opts))
        for full, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent key: {full}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent key: {full}")
            node[parts[-1]] = self._coerce(self._decode(v), node[parts[-1]], full)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)) for k, v in self.items()}

    def dump(self, **kw):
        return yaml.safe_dump(self.to_dict(), **kw)


CN = CfgNode

# detectron2 defaults (recalled from detectron2/config/defaults.py; only the keys the MGNet yamls and code read --
# SURVEY Appendix C -- plus their siblings that mgnet/config.py extends)
_D2_DEFAULTS = {
    "VERSION": 2,
    "MODEL": {
        "DEVICE": "cuda", "META_ARCHITECTURE": "GeneralizedRCNN", "WEIGHTS": "",
        "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
        "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 2},
        "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN", "WIDTH_PER_GROUP": 64,
                    "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64,
                    "DEFORM_ON_PER_STAGE": [False, False, False, False], "DEFORM_MODULATED": False,
                    "DEFORM_NUM_GROUPS": 1},
        "SEM_SEG_HEAD": {"NAME": "SemSegFPNHead", "IN_FEATURES": ["p2", "p3", "p4", "p5"], "IGNORE_VALUE": 255,
                         "NUM_CLASSES": 54, "CONVS_DIM": 128, "COMMON_STRIDE": 4, "NORM": "GN", "LOSS_WEIGHT": 1.0},
    },
    "INPUT": {"MIN_SIZE_TRAIN": (800,), "MIN_SIZE_TRAIN_SAMPLING": "choice", "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800,
              "MAX_SIZE_TEST": 1333, "RANDOM_FLIP": "horizontal",
              "CROP": {"ENABLED": False, "TYPE": "relative_range", "SIZE": [0.9, 0.9]}, "FORMAT": "BGR",
              "MASK_FORMAT": "polygon"},
    "DATASETS": {"TRAIN": (), "TEST": ()},
    "DATALOADER": {"NUM_WORKERS": 4, "ASPECT_RATIO_GROUPING": True, "SAMPLER_TRAIN": "TrainingSampler",
                   "FILTER_EMPTY_ANNOTATIONS": True},
    "SOLVER": {"LR_SCHEDULER_NAME": "WarmupMultiStepLR", "MAX_ITER": 40000, "BASE_LR": 0.001, "MOMENTUM": 0.9,
               "NESTEROV": False, "WEIGHT_DECAY": 0.0001, "WEIGHT_DECAY_NORM": 0.0, "GAMMA": 0.1, "STEPS": (30000,),
               "WARMUP_FACTOR": 1.0 / 1000, "WARMUP_ITERS": 1000, "WARMUP_METHOD": "linear", "CHECKPOINT_PERIOD": 5000,
               "IMS_PER_BATCH": 16, "REFERENCE_WORLD_SIZE": 0, "BIAS_LR_FACTOR": 1.0, "WEIGHT_DECAY_BIAS": None,
               "CLIP_GRADIENTS": {"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0},
               "AMP": {"ENABLED": False, "DTYPE": "bfloat16", "LOSS_SCALE_INIT": 65536.0, "LOSS_SCALE_GROWTH_INTERVAL": 2000}},
    "TEST": {"EXPECTED_RESULTS": [], "EVAL_PERIOD": 0},
    "OUTPUT_DIR": "./output", "SEED": -1, "CUDNN_BENCHMARK": False, "VIS_PERIOD": 0,
}

# MGNet additions -- same keys and default values as mgnet/config.py:12-138, written as data instead of statements
_MGNET_KEYS = {
    "CUDNN_DETERMINISTIC": False, "COMMIT_ID": "", "WRITE_OUTPUT_TO_SUBDIR": True, "WITH_PANOPTIC": True,
    "WITH_DEPTH": True, "WITH_UNCERTAINTY": True, "VISUALIZE_EVALUATION": False,
    "SOLVER": {"OPTIMIZER": "ADAM", "LR_SCHEDULER_NAME": "WarmupPolyLR", "POLY_LR_POWER": 0.9,
               "POLY_LR_CONSTANT_ENDING": 0.0, "WARMUP_FACTOR": 0.1, "WARMUP_ITERS": 1000, "HEAD_LR_FACTOR": 10.0},
    "INPUT": {"TRAIN_DATASET_MAPPER": "mgnet.data.MGNetTrainDatasetMapper",
              "TEST_DATASET_MAPPER": "mgnet.data.MGNetTestDatasetMapper",
              "COLOR_JITTER": {"ENABLED": True, "BRIGHTNESS": 0.2, "CONTRAST": 0.2, "SATURATION": 0.2, "HUE": 0.05},
              "CROP": {"RANDOM_PAD_TO_CROP_SIZE": True},
              "GAUSSIAN_SIGMA": 8, "IGNORE_STUFF_IN_OFFSET": True, "SMALL_INSTANCE_AREA": 4096,
              "SMALL_INSTANCE_WEIGHT": 3, "IGNORE_CROWD_IN_SEMANTIC": False, "IGNORED_CATEGORIES_IN_DEPTH": []},
    "MODEL": {
        "SIZE_DIVISIBILITY": 32,
        "GCM": {"GCM_CHANNELS": 128, "INIT_METHOD": "xavier"},
        "SEM_SEG_HEAD": {"ARM_CHANNELS": [128, 128], "REFINE_CHANNELS": [128, 128], "FFM_CHANNELS": 256,
                         "HEAD_CHANNELS": 256, "INIT_METHOD": "xavier", "LOSS_TYPE": "ohem", "LOSS_TOP_K": 0.2,
                         "OHEM_THRESHOLD": 0.7, "OHEM_N_MIN": 100000},
        "INS_EMBED_HEAD": {"NAME": "MGNetInsEmbedHead", "IN_FEATURES": ["res3", "res4", "res5"], "COMMON_STRIDE": 8,
                           "ARM_CHANNELS": [128, 128], "REFINE_CHANNELS": [128, 128], "FFM_CHANNELS": 256,
                           "HEAD_CHANNELS": 256, "INIT_METHOD": "xavier", "CENTER_LOSS_WEIGHT": 200.0,
                           "OFFSET_LOSS_WEIGHT": 0.01},
        "DEPTH_HEAD": {"NAME": "MGNetSelfSupervisedDepthHead", "IN_FEATURES": ["res3", "res4", "res5"],
                       "COMMON_STRIDE": 8, "ARM_CHANNELS": [128, 128], "REFINE_CHANNELS": [128, 128],
                       "FFM_CHANNELS": 256, "HEAD_CHANNELS": 256, "INIT_METHOD": "default", "MSC_LOSS": True,
                       "SSIM_LOSS_WEIGHT": 0.85, "PHOTOMETRIC_LOSS_WEIGHT": 1.0, "SMOOTHING_LOSS_WEIGHT": 0.001,
                       "AUTOMASK_LOSS": True, "PHOTOMETRIC_REDUCE_OP": "min", "PADDING_MODE": "zeros"},
        "POST_PROCESSING": {"STUFF_AREA": 2048, "CENTER_THRESHOLD": 0.3, "NMS_KERNEL": 7, "USE_DGC_SCALING": True},
    },
    "TEST": {"AMP": {"ENABLED": True}, "MSC_FLIP_EVAL": False, "EVAL_SEMANTIC": True, "EVAL_INSTANCE": False,
             "MIN_DEPTH": 0.001, "MAX_DEPTH": 80.0},
}


def _overlay(node, tree):
    for k, v in tree.items():
        if isinstance(v, dict):
            if k not in node:
                node[k] = CfgNode()
            _overlay(node[k], v)
        else:
            node[k] = copy.deepcopy(v)


def get_cfg():
    """detectron2.config.get_cfg() equivalent (defaults subset)."""
    return CfgNode(copy.deepcopy(_D2_DEFAULTS))


def add_mgnet_config(cfg):
    """Add config for MGNet (same keys/defaults as mgnet/config.py:6-138)."""
    _overlay(cfg, _MGNET_KEYS)
    return cfg
