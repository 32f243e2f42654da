"""Generates tests/golden/checkpoint_keys.json: the reference's OWN key renaming (tools/convert-torchvision-to-mgnet.py,
`convert_key`, imported by file path; its `__main__` block does not run) applied to every state-dict key of a torchvision
ResNet-18 / ResNet-34.  Runs only where /root/reference exists:  python tests/golden/make_golden_checkpoint.py"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tools/convert-torchvision-to-mgnet.py"


def torchvision_resnet_keys(blocks=(2, 2, 2, 2)):
    """State-dict key names of torchvision.models.resnet18/34 (BasicBlock nets); torchvision itself is not installed."""
    bn = ["weight", "bias", "running_mean", "running_var", "num_batches_tracked"]
    keys = ["conv1.weight"] + [f"bn1.{s}" for s in bn]
    for li, n in enumerate(blocks, start=1):
        for b in range(n):
            for c in (1, 2):
                keys.append(f"layer{li}.{b}.conv{c}.weight")
                keys += [f"layer{li}.{b}.bn{c}.{s}" for s in bn]
            if li > 1 and b == 0:
                keys.append(f"layer{li}.{b}.downsample.0.weight")
                keys += [f"layer{li}.{b}.downsample.1.{s}" for s in bn]
    return keys + ["fc.weight", "fc.bias"]


def main():
    spec = importlib.util.spec_from_file_location("ref_convert", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, blocks in (("resnet18", (2, 2, 2, 2)), ("resnet34", (3, 4, 6, 3))):
        for prefix in ("backbone", "pose_encoder"):
            out[f"{name}/{prefix}"] = {k: mod.convert_key(k, prefix) for k in torchvision_resnet_keys(blocks)}
    with open(os.path.join(HERE, "checkpoint_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    main()
