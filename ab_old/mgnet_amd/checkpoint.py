"""Checkpoint import/export in the reference's file formats, so weights move in both directions:

* `.pth` -- what detectron2's DetectionCheckpointer writes for the reference (tools/train_net.py:222-234):
  `torch.save({"model": state_dict, "optimizer": ..., "scheduler": ..., "iteration": n})`; the state-dict keys of
  mgnet_amd.modeling.MGNet equal the reference's, the optimizer entry has torch.optim.Adam's layout
  (`FusedAdam.state_dict`).
* `.pkl` -- ImageNet initialisation produced by tools/convert-torchvision-to-mgnet.py:8-50 (yaml `MODEL.WEIGHTS:
  "./weights/imagenet_weights.pkl"`): `{"model": {key: ndarray}, "__author__": "torchvision", "matching_heuristics": True}`
  with torchvision ResNet keys renamed (`convert_key`) under the prefixes `backbone.` / `pose_encoder.`; loaded with
  detectron2's suffix-matching heuristic (recalled: a checkpoint key is assigned to the model key it is the longest
  suffix of), which is how `pose_encoder.*` finds `pose_net.pose_encoder.*`.
"""
import os
import pickle
import re
from typing import Dict, List, NamedTuple

import numpy as np
import torch

__all__ = ["convert_key", "convert_torchvision_resnets", "Checkpointer", "Incompatible"]

# torchvision ResNet name -> MGNet name (convert-torchvision-to-mgnet.py:8-19), applied in this order
_RULES = [(re.compile(r"layer([1-4])"), lambda m: "res%d" % (int(m.group(1)) + 1)),
          (re.compile(r"bn([1-3])"), lambda m: "conv%s.norm" % m.group(1)),
          (re.compile(r"downsample\.0"), lambda m: "shortcut"),
          (re.compile(r"downsample\.1"), lambda m: "shortcut.norm")]


def convert_key(k: str, prefix: str = "") -> str:
    """`layerN` -> `res(N+1)`, `bnN` -> `convN.norm`, `downsample.0/1` -> `shortcut[.norm]`; everything outside the four
    stages (conv1, bn1, fc) lives under `stem.`; result is `<prefix>.<name>`."""
    if "layer" not in k:
        k = "stem." + k
    for pat, rep in _RULES:
        k = pat.sub(rep, k)
    return prefix + "." + k


def convert_torchvision_resnets(backbone_sd: Dict[str, torch.Tensor], pose_encoder_sd: Dict[str, torch.Tensor]) -> dict:
    """The dictionary tools/convert-torchvision-to-mgnet.py pickles from two torchvision ResNet state dicts: the second
    becomes the 9-channel pose encoder (its 3-channel stem weight repeated three times along Cin and divided by 3, :39-42)."""
    model = {}
    for k, v in backbone_sd.items():
        model[convert_key(k, "backbone")] = v.detach().cpu().numpy()
    for k, v in pose_encoder_sd.items():
        nk = convert_key(k, "pose_encoder")
        if nk == "pose_encoder.stem.conv1.weight":
            v = torch.cat([v] * 3, 1) / 3
        model[nk] = v.detach().cpu().numpy()
    return {"model": model, "__author__": "torchvision", "matching_heuristics": True}


class Incompatible(NamedTuple):
    missing_keys: List[str]          # model keys that received nothing
    unexpected_keys: List[str]       # checkpoint keys that matched no model key
    shape_mismatch: List[tuple]      # (model key, checkpoint shape, model shape): skipped


def _match(model_keys, ckpt_keys, heuristics):
    """model key -> checkpoint key.  Exact names first; with `heuristics` the checkpoint key that is the longest
    dot-aligned suffix of the model key."""
    ckpt = set(ckpt_keys)
    out = {}
    for mk in model_keys:
        if mk in ckpt:
            out[mk] = mk
        elif heuristics:
            parts = mk.split(".")
            for i in range(1, len(parts)):
                cand = ".".join(parts[i:])
                if cand in ckpt:
                    out[mk] = cand
                    break
    return out


class Checkpointer:
    """DetectionCheckpointer-shaped: `save(name, **extra)`, `load(path)`, `resume_or_load(path, resume=)`,
    `has_checkpoint()`, `get_checkpoint_file()`; checkpointables (optimizer, scheduler) by keyword like detectron2's."""

    def __init__(self, model, save_dir="", **checkpointables):
        self.model, self.save_dir, self.checkpointables = model, save_dir, dict(checkpointables)

    # ---- export ----------------------------------------------------------------------------------------------------
    def save(self, name, **extra):
        data = {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}
        for k, obj in self.checkpointables.items():
            data[k] = obj.state_dict()
        data.update(extra)
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, f"{name}.pth")
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(os.path.basename(path))
        return path

    # ---- import ----------------------------------------------------------------------------------------------------
    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
            return os.path.join(self.save_dir, f.read().strip())

    @staticmethod
    def _read(path):
        if path.endswith(".pkl"):
            with open(path, "rb") as f:
                data = pickle.load(f, encoding="latin1")
            if "model" not in data:   # a bare {key: array} pickle
                data = {"model": data, "matching_heuristics": True}
            return data
        data = torch.load(path, map_location="cpu", weights_only=False)
        return data if "model" in data else {"model": data}

    def load(self, path, checkpointables=None):
        """Loads the model (and the named checkpointables present in the file); returns the remaining entries
        (e.g. "iteration") and leaves the matching report in `self.last_incompatible`."""
        if not path:
            self.last_incompatible = Incompatible([], [], [])
            return {}
        data = self._read(path)
        self.last_incompatible = self._load_model(data.pop("model"), bool(data.pop("matching_heuristics", False)))
        for k in (self.checkpointables if checkpointables is None else checkpointables):
            if k in data:
                self.checkpointables[k].load_state_dict(data.pop(k))
        data.pop("__author__", None)
        return data

    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])

    def _load_model(self, ckpt, heuristics):
        ckpt = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in ckpt.items()}
        sd = self.model.state_dict()
        match = _match(sd.keys(), ckpt.keys(), heuristics)
        mismatch, used = [], set()
        with torch.no_grad():
            for mk, ck in match.items():
                used.add(ck)
                if tuple(ckpt[ck].shape) != tuple(sd[mk].shape):
                    mismatch.append((mk, tuple(ckpt[ck].shape), tuple(sd[mk].shape)))
                    continue
                sd[mk].copy_(ckpt[ck])   # in place: parameters may be views of the optimizer's flat buffers
        bad = {m[0] for m in mismatch}
        return Incompatible([k for k in sd if k not in match or k in bad], [k for k in ckpt if k not in used], mismatch)
