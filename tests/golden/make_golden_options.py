#!/usr/bin/env python3
"""Golden vectors for the NON-DEFAULT options of the reference's MultiViewPhotometricLoss (mgnet/modeling/loss.py:92-109, 131-144, 222-255):
`automask_loss=False` with `photometric_reduce_op` "min" and "mean", `padding_mode` "border" / "reflection" of the warp
(camera_utils.py:24-55 -> F.grid_sample), and `ssim_loss_weight=0` (loss.py:196-197: the photometric maps are then the 3-channel L1 maps --
"min" runs over channels AND sources and only exists WITH a reprojection mask (without one the default mask is built [B,3,H,W] and
cannot index the [B,1,H,W] minimum), "mean" only exists WITHOUT one (`loss[mask]` of a [B,3,H,W] map with a [B,1,H,W] mask): the other
two combinations raise IndexError in the reference, recorded as `<tag>.raises`).  Same recipe as make_golden.py (the reference's own mgnet.geometry +
mgnet.modeling.loss imported in the build container, CPU fp32); inputs are the cases of make_golden.py, only outputs are stored:
    tests/golden/reproj_options.npz   keys "<case>.<automask>.<reduce>[.<padding>][.ssim0].<loss_photometric | loss_smoothness | dphot_dinv<i> | dphot_dposes>"
(the smoothness term does not depend on the options; its gradient is pinned by reproj_<case>.npz)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

COMBOS = [(False, "min", "zeros", 0.85), (False, "mean", "zeros", 0.85), (True, "min", "border", 0.85), (True, "min", "reflection", 0.85),
          (False, "mean", "border", 0.85),
          (True, "min", "zeros", 0.0), (False, "min", "zeros", 0.0), (False, "mean", "zeros", 0.0), (True, "min", "border", 0.0)]
CASES = ["rand_small", "oob_clamp", "no_mask_odd"]


def run(L, c, automask, reduce_op, padding_mode="zeros", ssim_w=0.85):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    inv = [t(a).requires_grad_(True) for a in c["inv"]]
    poses = t(c["poses"]).requires_grad_(True)
    targets = {"image_orig": t(c["img"]), "image_prev_orig": t(c["prev"]), "image_next_orig": t(c["nxt"]), "camera_matrix": t(c["K"])}
    if c["mask"] is not None:
        targets["reprojection_mask"] = t(c["mask"])
    loss = L.MultiViewPhotometricLoss(ssim_w, 1.0, 0.001, automask, reduce_op, padding_mode)
    out = loss({"depth": inv, "poses": poses}, targets)
    gp = torch.autograd.grad(out["loss_photometric"], inv + [poses], allow_unused=True)
    res = {"loss_photometric": out["loss_photometric"].detach().numpy(), "loss_smoothness": out["loss_smoothness"].detach().numpy(),
           "dphot_dposes": gp[3].numpy()}
    for i in range(3):
        res[f"dphot_dinv{i}"] = gp[i].numpy()
    return res


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    _, L = MG.import_reference()
    blob = {}
    for name in CASES:
        c = MG.build_case(name)
        for automask, red, pad, sw in COMBOS:
            tag = f"{name}.{int(automask)}.{red}" + ("" if pad == "zeros" else "." + pad) + ("" if sw else ".ssim0")
            try:
                r = run(L, c, automask, red, pad, sw)
            except IndexError as e:
                assert sw == 0.0 and (red == "mean") == (c["mask"] is not None), (tag, e)
                blob[f"{tag}.raises"] = np.array(1)
                print(f"{name:12s} automask={automask} reduce={red} padding={pad} ssim_w={sw}: IndexError ({str(e)[:60]}...)")
                continue
            for k, v in r.items():
                blob[f"{tag}.{k}"] = v
            print(f"{name:12s} automask={automask} reduce={red} padding={pad} ssim_w={sw}: Lp={float(r['loss_photometric']):.7f}")
    path = os.path.join(HERE, "reproj_options.npz")
    np.savez_compressed(path, **blob)
    print(f"-> {path} {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
