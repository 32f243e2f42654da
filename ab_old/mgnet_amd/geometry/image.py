"""mgnet/geometry/image.py:20-199 -- shape helpers, finite differences, pixel grids."""
from functools import lru_cache

import torch
import torch.nn.functional as F

__all__ = ["same_shape", "gradient_x", "gradient_y", "interpolate_image", "match_scales", "meshgrid", "image_grid"]


def same_shape(shape1, shape2):
    return len(shape1) == len(shape2) and all(a == b for a, b in zip(shape1, shape2))


def gradient_x(image):
    """[B,C,H,W] -> [B,C,H,W-1]: left minus right neighbour (image.py:42-55)"""
    return image[:, :, :, :-1] - image[:, :, :, 1:]


def gradient_y(image):
    """[B,C,H,W] -> [B,C,H-1,W]: upper minus lower neighbour (image.py:58-69)"""
    return image[:, :, :-1, :] - image[:, :, 1:, :]


def interpolate_image(image, shape, mode="bilinear", align_corners=True):
    shape = tuple(shape[-2:]) if len(shape) > 2 else tuple(shape)
    if same_shape(tuple(image.shape[-2:]), shape):
        return image
    return F.interpolate(image, size=shape, mode=mode, align_corners=align_corners)


def match_scales(image, targets, num_scales, mode="bilinear", align_corners=True):
    """one copy of `image` per scale, resized to targets[i]'s resolution (image.py:101-135; note the reference compares the
    image's (H,W) with the target's FULL shape, so equal resolutions still go through interpolate_image's own check)"""
    return [interpolate_image(image, targets[i].shape, mode=mode, align_corners=align_corners) for i in range(num_scales)]


@lru_cache(maxsize=None)
def meshgrid(B, H, W, dtype, device, normalized=False):
    """xs, ys [B,H,W]: pixel (or [-1,1]) coordinates (image.py:138-170)"""
    lo_x, hi_x, lo_y, hi_y = (-1, 1, -1, 1) if normalized else (0, W - 1, 0, H - 1)
    xs = torch.linspace(lo_x, hi_x, W, device=device, dtype=dtype)
    ys = torch.linspace(lo_y, hi_y, H, device=device, dtype=dtype)
    ys, xs = torch.meshgrid([ys, xs], indexing="ij")
    return xs.repeat([B, 1, 1]), ys.repeat([B, 1, 1])


@lru_cache(maxsize=None)
def image_grid(B, H, W, dtype, device, normalized=False):
    """[B,3,H,W] homogeneous pixel grid (u, v, 1) (image.py:173-199)"""
    xs, ys = meshgrid(B, H, W, dtype, device, normalized=normalized)
    return torch.stack([xs, ys, torch.ones_like(xs)], dim=1)
