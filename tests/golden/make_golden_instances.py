"""Generates tests/golden/instances_*.npz from the reference's OWN mgnet/postprocessing/instance_post_proc.py (imported
unmodified).  Runs only where /root/reference exists (the build container):

    python tests/golden/make_golden_instances.py

detectron2 is neither installed nor vendored: the harness supplies stand-ins for the two names that file imports from
`detectron2.structures` -- `Instances` (attribute bag) and `BitMasks.get_bounding_boxes` (published semantics: per mask
[x_min, y_min, x_max + 1, y_max + 1] of its non-zero pixels, zeros for an empty mask).  Parent packages are stubbed so that
`mgnet/__init__.py` is not executed.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mgnet"
THING_IDS = list(range(11, 19))
DIV = 1000


def import_reference():
    for name, path in (("mgnet", REF), ("mgnet.postprocessing", REF + "/postprocessing")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    class Boxes:
        def __init__(self, t):
            self.tensor = t

    class BitMasks:
        def __init__(self, t):
            self.tensor = t

        def get_bounding_boxes(self):
            boxes = torch.zeros(self.tensor.shape[0], 4, dtype=torch.float32)
            x_any, y_any = torch.any(self.tensor, dim=1), torch.any(self.tensor, dim=2)
            for idx in range(self.tensor.shape[0]):
                x, y = torch.where(x_any[idx, :])[0], torch.where(y_any[idx, :])[0]
                if len(x) > 0 and len(y) > 0:
                    boxes[idx, :] = torch.as_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
            return Boxes(boxes)

    class Instances:
        def __init__(self, image_size):
            self.image_size = image_size

    d2 = types.ModuleType("detectron2")
    st = types.ModuleType("detectron2.structures")
    st.BitMasks, st.Instances = BitMasks, Instances
    d2.structures = st
    sys.modules["detectron2"], sys.modules["detectron2.structures"] = d2, st
    import mgnet.postprocessing.instance_post_proc as I  # noqa
    return I


def case(seed, H, W, C=19, n_inst=9, tiny=False, no_things=False):
    rs = np.random.RandomState(seed)
    blk = rs.randint(0, 11, size=((H + 7) // 8, (W + 7) // 8))
    cls = np.kron(blk, np.ones((8, 8), dtype=np.int64))[:H, :W]
    pan = cls * DIV                                   # stuff segments: class * divisor
    pan[rs.rand(H, W) < 0.03] = -1                    # void
    yy, xx = np.mgrid[0:H, 0:W]
    if not no_things:
        for k in range(n_inst):
            c = THING_IDS[rs.randint(len(THING_IDS))]
            cy, cx = rs.randint(0, H), rs.randint(0, W)
            r = 1 if (tiny and k % 3 == 0) else rs.randint(2, max(3, min(H, W) // 3))
            m = (yy - cy) ** 2 * (1 + rs.rand()) + (xx - cx) ** 2 <= r * r
            if tiny and k % 3 == 0:
                m = (yy == cy) & (xx == cx)            # single-pixel instance
            pan[m] = c * DIV + k + 1
    sem = (rs.randn(C, H, W) * 2).astype(np.float32)
    lab = np.where(pan >= 0, pan // DIV, 0)
    sem[lab, yy, xx] += 3.0                           # the predicted class mostly wins
    heat = rs.rand(1, H, W).astype(np.float32)
    return sem, heat, pan.astype(np.int64)


def main():
    I = import_reference()
    cases = {"basic": case(1, 40, 64), "ragged_tiny": case(2, 37, 51, n_inst=12, tiny=True), "no_things": case(3, 24, 32, no_things=True),
             "many": case(4, 96, 160, n_inst=150)}
    for name, (sem, heat, pan) in cases.items():
        out = I.get_instance_predictions(torch.from_numpy(sem), torch.from_numpy(heat), torch.from_numpy(pan), THING_IDS, DIV)
        n = len(out)
        H, W = pan.shape
        np.savez_compressed(
            os.path.join(HERE, f"instances_{name}.npz"), sem=sem, heat=heat, pan=pan, thing_ids=np.array(THING_IDS), label_divisor=DIV,
            classes=np.array([int(i.pred_classes[0]) for i in out], dtype=np.int64),
            scores=np.array([float(i.scores[0]) for i in out], dtype=np.float32),
            boxes=np.stack([i.pred_boxes.tensor[0].numpy() for i in out]) if n else np.zeros((0, 4), np.float32),
            masks=np.packbits(np.stack([i.pred_masks[0].numpy() for i in out]).reshape(n, -1), axis=1) if n else np.zeros((0, 0), np.uint8))
        print(name, n, "instances")


if __name__ == "__main__":
    main()
