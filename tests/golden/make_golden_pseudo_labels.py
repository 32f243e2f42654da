"""Generates tests/golden/pseudo_labels.npz by EXECUTING the id arithmetic of the reference's tools/generate_pseudo_labels.py
(the statements between its comment `# Remap train_ids to ids` and the output-path logic, read from /root/reference at generation
time and run unmodified on synthetic panoptic predictions; the script as a whole needs detectron2 and datasets and cannot be
imported).  Runs only where /root/reference exists:   python tests/golden/make_golden_pseudo_labels.py"""
import os
import textwrap
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tools/generate_pseudo_labels.py"


def reference_remap():
    lines = open(SRC).read().splitlines()
    a = next(i for i, l in enumerate(lines) if "# Remap train_ids to ids" in l) + 1
    b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith("output_path = _input"))
    code = compile(textwrap.dedent("\n".join(lines[a:b])), SRC, "exec")

    def remap(panoptic_prediction, label_divisor, id_map):
        # `id_map[...] * meta.label_divisor` is a uint8 array times the Python int 1000: value-based promotion (-> uint16, exact for
        # ids <= 65) under the NumPy < 2 the reference runs on, an OverflowError under NumPy >= 2 (NEP 50).  The harness's `meta`
        # stand-in therefore carries the divisor as np.int64 (exact integer arithmetic, the same values as the legacy promotion)
        ns = dict(panoptic_prediction=panoptic_prediction.copy(), meta=types.SimpleNamespace(label_divisor=np.int64(label_divisor)),
                  id_map=id_map, np=np)
        exec(code, ns)
        return ns["panoptic_prediction"].astype(np.uint16)     # (:134 Image.fromarray(panoptic_prediction.astype(np.uint16)))
    return remap


def main():
    remap = reference_remap()
    rs = np.random.RandomState(0)
    # Cityscapes: trainId -> id of the 19 evaluation classes (cityscapesscripts labels), things = trainIds 11..18
    ids = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]
    id_map = np.zeros(256, dtype=np.uint8)
    id_map[:19] = ids
    H, W, div = 48, 80, 1000
    blk = rs.randint(0, 19, size=(H // 8, W // 8))
    pan = (np.kron(blk, np.ones((8, 8), dtype=np.int64)) * div).astype(np.int64)        # every class as a stuff-style segment (instance 0)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(14):
        cy, cx, r = rs.randint(H), rs.randint(W), rs.randint(2, 9)
        pan[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = rs.randint(11, 19) * div + k + 1
    pan[rs.rand(H, W) < 0.05] = -1                                                       # void
    out = remap(pan, div, id_map)
    np.savez_compressed(os.path.join(HERE, "pseudo_labels.npz"), pan=pan, id_map=id_map, label_divisor=div, out=out)
    print("pseudo_labels.npz", out.shape, out.dtype, np.unique(out)[:12])


if __name__ == "__main__":
    main()
