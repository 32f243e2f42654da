"""ImageList.from_tensors (detectron2.structures equivalent used at mg_net.py:250-345): zero-pad a list of
[..., H, W] tensors to a common size divisible by `size_divisibility` and stack them."""
import torch
import torch.nn.functional as F


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        assert len(tensors) > 0
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
        if all(s == (H, W) for s in sizes):
            whole = _as_one_batch(tensors)   # slices of one batched buffer (device-side target generation): no copy
            return ImageList(torch.stack(list(tensors), 0) if whole is None else whole, sizes)
        out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (H, W), pad_value)
        for t, o in zip(tensors, out):
            o[..., : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out, sizes)


def _as_one_batch(tensors):
    """If the tensors are the consecutive slices t[0], t[1], ... of ONE contiguous batched tensor, return that batch as a
    view (what `torch.stack` would produce, without the copy); else None."""
    t0 = tensors[0]
    if not t0.is_contiguous() or t0.numel() == 0:
        return None
    step, base = t0.numel(), t0.storage_offset()
    sp = t0.untyped_storage().data_ptr()
    for i, t in enumerate(tensors):
        if (t.shape != t0.shape or t.dtype != t0.dtype or t.device != t0.device or not t.is_contiguous()
                or t.untyped_storage().data_ptr() != sp or t.storage_offset() != base + i * step):
            return None
    return t0.as_strided((len(tensors),) + tuple(t0.shape), (step,) + tuple(t0.stride()), base)
