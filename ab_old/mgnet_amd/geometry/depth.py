"""mgnet/geometry/depth.py:11-51 -- inverse depth helpers."""
import torch

from .image import gradient_x, gradient_y

__all__ = ["inv2depth", "calc_smoothness"]


def inv2depth(inv_depth):
    """1 / clamp(inv_depth, 1e-6), element-wise; lists/tuples map to lists (depth.py:11-15)"""
    if isinstance(inv_depth, (tuple, list)):
        return [inv2depth(x) for x in inv_depth]
    return 1.0 / inv_depth.clamp(min=1e-6)


def _mean_normalized(inv_depths):
    """inverse depth divided by its per-image mean (clamped at 1e-6) (depth.py:28-51)"""
    return [d / d.mean(2, True).mean(3, True).clamp(min=1e-6) for d in inv_depths]


def calc_smoothness(inv_depths, image, num_scales):
    """edge-aware first-order smoothness terms per scale: d(normalised inverse depth) * exp(-mean_c |d image|)
    -> (list of [B,1,H,W-1], list of [B,1,H-1,W])   (depth.py:18-25)"""
    norm = _mean_normalized(inv_depths)
    wx = torch.exp(-gradient_x(image).abs().mean(1, keepdim=True))
    wy = torch.exp(-gradient_y(image).abs().mean(1, keepdim=True))
    return ([gradient_x(norm[i]) * wx for i in range(num_scales)], [gradient_y(norm[i]) * wy for i in range(num_scales)])
