#!/usr/bin/env python3
"""Pseudo-label generation (mirrors /tools/generate_pseudo_labels.py:67-137 of the reference).

    python tools/generate_pseudo_labels.py --config-file <MGNet-*.yaml> --input <dir of images> --output <gt dir>
           [--weights model.pth] [--synthetic N] [opts: KEY VALUE ...]

For every input frame: `model.eval()` forward (single scale, or multi-scale + flip with TEST.MSC_FLIP_EVAL True) -> panoptic
prediction on the device (csrc/postproc.hip) -> the dataset's `instanceIds` image with `mgn_pseudo_label_ids` (csrc/instances.hip:
stuff -> id_map[trainId], things -> id_map[trainId] * label_divisor + instance) -> uint16 PNG under the output directory, named like
the reference names them (`_leftImg8bit` -> `_gtFine_instanceIds`, under the frame's city directory).  The reference's dataset
registration / test mapper / DDP sharding / COCO-panoptic conversion are data plumbing outside SURVEY 8 and are not rebuilt: frames
come from a directory (any image PIL opens) or are synthetic (`--synthetic N`, the smoke path without data files)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def id_map_from_meta(meta, kitti=False):
    """generate_pseudo_labels.py:92-97: trainId -> id; KITTI leaves the ego vehicle out"""
    id_map = np.zeros(256, dtype=np.uint8)
    for cat in meta.categories:
        if cat["name"] == "ego vehicle" and kitti:
            continue
        if 0 <= cat["trainId"] < 256:
            id_map[cat["trainId"]] = cat["id"]
    return id_map


def output_path_for(file_name, gt_dir, cityscapes=True):
    """generate_pseudo_labels.py:120-131"""
    if cityscapes:
        out = os.path.basename(file_name)
        out = os.path.join(file_name.split("/")[-2] if "/" in file_name else "", out)
        out = out.replace("_leftImg8bit", "_gtFine_instanceIds")
        return os.path.join(gt_dir, os.path.splitext(out)[0] + ".png")
    return os.path.join(gt_dir, os.path.splitext(file_name.replace("image", "label"))[0] + ".png")


def main():
    from PIL import Image

    from mgnet_amd import _C, add_mgnet_config, get_cfg
    from mgnet_amd.checkpoint import Checkpointer
    from mgnet_amd.data.metadata import MetadataCatalog
    from mgnet_amd.registry import build_model

    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default=os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    ap.add_argument("--input", default=None, help="directory of frames (searched recursively)")
    ap.add_argument("--output", required=True, help="gt directory the instanceIds PNGs are written to")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic frames instead of --input")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("opts", nargs=argparse.REMAINDER)
    args = ap.parse_args()

    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(args.config_file)
    # (only the panoptic branch is needed; the depth branch's DGC rescaling would ask the mapper for camera matrices)
    cfg.merge_from_list(["MODEL.DEVICE", "cuda:0", "WITH_DEPTH", False] + list(args.opts))
    assert cfg.WITH_PANOPTIC, "WITH_PANOPTIC = True is required for pseudo label generation!"   # (:34)
    # (the reference's MGNet-*-PseudoLabelGeneration.yaml set SOLVER.AMP.ENABLED False: fp32 inference.  The config is respected: every
    #  convolution then runs as three bf16 MFMA passes with fp32 accumulation, ops._conv2d_fp32_split -- ~1e-5 of an fp32 convolution;
    #  `SOLVER.AMP.ENABLED True` on the command line selects the faster 16-bit trunk)
    model = build_model(cfg).eval()
    if args.weights:
        Checkpointer(model, save_dir=args.output).load(args.weights)
    name = cfg.DATASETS.TRAIN[0] if len(cfg.DATASETS.TRAIN) else "cityscapes"
    meta = MetadataCatalog.get(name)
    id_map = id_map_from_meta(meta, kitti="kitti" in name)

    if args.synthetic:
        g = torch.Generator().manual_seed(0)
        frames = [(f"synthetic/frame_{i:06d}_leftImg8bit.png", torch.randint(0, 256, (3, args.height, args.width), generator=g, dtype=torch.uint8))
                  for i in range(args.synthetic)]
    else:
        assert args.input, "--input or --synthetic"
        files = sorted(os.path.join(d, f) for d, _, fs in os.walk(args.input) for f in fs if f.lower().endswith((".png", ".jpg", ".jpeg")))
        frames = ((f, torch.from_numpy(np.asarray(Image.open(f).convert("RGB"))).permute(2, 0, 1).contiguous()) for f in files)
    n = 0
    for file_name, img in frames:
        with torch.no_grad():
            out = model([{"image": img.cuda(), "height": img.shape[1], "width": img.shape[2]}])[0]
        pan = out["panoptic_seg"][0].long().contiguous()
        ids = _C.pseudo_label_ids(pan, meta.label_divisor, id_map).cpu().numpy().view(np.uint16)
        path = output_path_for(file_name, args.output, cityscapes="kitti" not in name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(ids).save(path)
        n += 1
    print(f"wrote {n} instanceIds images under {args.output}")


if __name__ == "__main__":
    main()
