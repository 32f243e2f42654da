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
    batch = []
    for b in range(B):
        d = {"image": jitter(orig)[b], "height": H, "width": W}
        if with_depth:
            d.update({"image_prev": jitter(prev)[b], "image_next": jitter(nxt)[b], "image_orig": orig[b],
                      "image_prev_orig": prev[b], "image_next_orig": nxt[b], "camera_matrix": K.clone(),
                      "reprojection_mask": rnd(H, W) < 0.9})
        if with_panoptic:
            blk = torch.randint(0, num_classes, (math.ceil(H / 32), math.ceil(W / 32)), device=device, generator=g)
            sem = blk.repeat_interleave(32, 0).repeat_interleave(32, 1)[:H, :W].contiguous().long()
            sem[rnd(H, W) < 0.02] = 255
            cy, cx = rnd(20) * H, rnd(20) * W
            gy = torch.arange(H, device=device, dtype=torch.float32)[None, :, None]
            gx = torch.arange(W, device=device, dtype=torch.float32)[None, None, :]
            center = torch.exp(-((gy - cy[:, None, None]) ** 2 + (gx - cx[:, None, None]) ** 2) / (2 * 8.0 ** 2)).amax(0)
            ow = (rnd(1, H, W) < 0.3).float()
            d.update({"sem_seg": sem, "sem_seg_weights": torch.where(rnd(H, W) < 0.05, 3.0, 1.0),
                      "center": center, "center_weights": (rnd(1, H, W) < 0.7).float(),
                      "offset": (rnd(2, H, W) * 128 - 64) * ow, "offset_weights": ow})
        batch.append(d)
    return batch
