"""Synthetic Cityscapes-shaped training batches in the reference's per-sample dict format (the
producer in the reference is mgnet/data/dataset_mapper.py:129-259 + target_generator.py:54-158).  Generated directly on
the device; contents follow SURVEY 8(d)."""
import math

import torch


def synthetic_batch(B, H, W, device, seed=1234, with_panoptic=True, with_depth=True, num_classes=20):
    g = torch.Generator(device=device).manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, device=device, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W + 16, device=device, dtype=torch.float32), indexing="ij")
    base = torch.zeros(B, 3, H, W + 16, device=device)
    for _ in range(8):
        f = rnd(B, 3, 2) * 0.2 + 0.005
        base += torch.sin(f[..., 0, None, None] * xx + f[..., 1, None, None] * yy + rnd(B, 3, 1, 1) * 6.283)
    lo, hi = base.amin((2, 3), keepdim=True), base.amax((2, 3), keepdim=True)
    base = (0.95 * (base - lo) / (hi - lo) + 0.05 * rnd(*base.shape)).clamp(0, 1)
    u8 = lambda t: (t * 255).round().to(torch.uint8)
    orig = u8(base[..., 8:8 + W])
    prev = u8(torch.roll(base[..., 5:5 + W], 1, 2))
    nxt = u8(torch.roll(base[..., 11:11 + W], -1, 2))
    jit = (0.8 + 0.4 * rnd(B, 1, 1, 1))
    jitter = lambda t: (t.float() * jit).clamp(0, 255).to(torch.uint8)
    sx, sy = W / 2048.0, H / 1024.0
    K = torch.eye(4)
    K[0, 0], K[1, 1] = 2262.52 * sx, 2265.30 * sy
    K[0, 2], K[1, 2] = (1096.98 + 0.5) * sx - 0.5, (513.137 + 0.5) * sy - 0.5
    # Every entry of the per-frame dicts is a SLICE of one batched device buffer -- what a loader that collates into batch
    # buffers hands over, and what the device-side target generator produces (data/target_generator.py) -- so that
    # `MGNet._stack` / `ImageList.from_tensors` re-assemble the batch as a view instead of copying 13 tensors per step.
    cols = {"image": jitter(orig)}
    if with_depth:
        cols.update({"image_prev": jitter(prev), "image_next": jitter(nxt), "image_orig": orig, "image_prev_orig": prev,
                     "image_next_orig": nxt, "camera_matrix": K.to(device).repeat(B, 1, 1),
                     "reprojection_mask": rnd(B, H, W) < 0.9})
    if with_panoptic:
        sems, centers = [], []
        gy = torch.arange(H, device=device, dtype=torch.float32)[None, :, None]
        gx = torch.arange(W, device=device, dtype=torch.float32)[None, None, :]
        for b in range(B):
            blk = torch.randint(0, num_classes, (math.ceil(H / 32), math.ceil(W / 32)), device=device, generator=g)
            sem = blk.repeat_interleave(32, 0).repeat_interleave(32, 1)[:H, :W].contiguous().long()
            sem[rnd(H, W) < 0.02] = 255
            cy, cx = rnd(20) * H, rnd(20) * W
            sems.append(sem)
            centers.append(torch.exp(-((gy - cy[:, None, None]) ** 2 + (gx - cx[:, None, None]) ** 2) / (2 * 8.0 ** 2)).amax(0))
        ow = (rnd(B, 1, H, W) < 0.3).float()
        cols.update({"sem_seg": torch.stack(sems), "sem_seg_weights": torch.where(rnd(B, H, W) < 0.05, 3.0, 1.0),
                     "center": torch.stack(centers), "center_weights": (rnd(B, 1, H, W) < 0.7).float(),
                     "offset": (rnd(B, 2, H, W) * 128 - 64) * ow, "offset_weights": ow})
    cols = {k: v.contiguous() for k, v in cols.items()}
    batch = []
    for b in range(B):
        d = {"height": H, "width": W}
        d.update({k: v[b] for k, v in cols.items()})
        batch.append(d)
    return batch
