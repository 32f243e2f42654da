#!/usr/bin/env python3
"""Golden vectors for the NON-DEFAULT options of the reference's MultiViewPhotometricLoss (mgnet/modeling/loss.py:92-109, 131-144, 222-255):
`automask_loss=False` with `photometric_reduce_op` "min" and "mean", and `padding_mode` "border" / "reflection" of the warp
(camera_utils.py:24-55 -> F.grid_sample).  Same recipe as make_golden.py (the reference's own mgnet.geometry +
mgnet.modeling.loss imported in the build container, CPU fp32); inputs are the cases of make_golden.py, only outputs are stored:
    tests/golden/reproj_options.npz   keys "<case>.<automask>.<reduce>[.<padding>].<loss_photometric | loss_smoothness | dphot_dinv<i> | dphot_dposes>"
(the smoothness term does not depend on the options; its gradient is pinned by reproj_<case>.npz)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

COMBOS = [(False, "min", "zeros"), (False, "mean", "zeros"), (True, "min", "border"), (True, "min", "reflection"), (False, "mean", "border")]
CASES = ["rand_small", "oob_clamp", "no_mask_odd"]


def run(L, c, automask, reduce_op, padding_mode="zeros"):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    inv = [t(a).requires_grad_(True) for a in c["inv"]]
    poses = t(c["poses"]).requires_grad_(True)
    targets = {"image_orig": t(c["img"]), "image_prev_orig": t(c["prev"]), "image_next_orig": t(c["nxt"]), "camera_matrix": t(c["K"])}
    if c["mask"] is not None:
        targets["reprojection_mask"] = t(c["mask"])
    loss = L.MultiViewPhotometricLoss(0.85, 1.0, 0.001, automask, reduce_op, padding_mode)
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
        for automask, red, pad in COMBOS:
            r = run(L, c, automask, red, pad)
            tag = f"{name}.{int(automask)}.{red}" + ("" if pad == "zeros" else "." + pad)
            for k, v in r.items():
                blob[f"{tag}.{k}"] = v
            print(f"{name:12s} automask={automask} reduce={red} padding={pad}: Lp={float(r['loss_photometric']):.7f}")
    path = os.path.join(HERE, "reproj_options.npz")
    np.savez_compressed(path, **blob)
    print(f"-> {path} {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
