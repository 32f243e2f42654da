"""ImageList.from_tensors (detectron2.structures equivalent used at mg_net.py:250-345): zero-pad a list of
[..., H, W] tensors to a common size divisible by `size_divisibility` and stack them."""
import torch
import torch.nn.functional as F


class Boxes:
    """detectron2.structures.Boxes equivalent (the part instance_post_proc.py:68 produces): `tensor` [N, 4] = x1, y1, x2, y2."""

    def __init__(self, tensor):
        self.tensor = tensor.reshape(-1, 4)

    def __len__(self):
        return self.tensor.shape[0]

    @staticmethod
    def cat(boxes_list):
        return Boxes(torch.cat([b.tensor for b in boxes_list], 0))


class Instances:
    """detectron2.structures.Instances equivalent (what mg_net.py:394-402 returns under `"instances"`): an image size plus
    per-instance fields of equal length set as attributes (`pred_classes`, `pred_masks`, `scores`, `pred_boxes`)."""

    def __init__(self, image_size, **fields):
        object.__setattr__(self, "_image_size", tuple(image_size))
        object.__setattr__(self, "_fields", {})
        for k, v in fields.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        if self._fields:
            assert len(self) == len(value), f"Adding a field of length {len(value)} to Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists):
        assert len(instance_lists) > 0 and all(i.image_size == instance_lists[0].image_size for i in instance_lists)
        if len(instance_lists) == 1:
            return instance_lists[0]
        ret = Instances(instance_lists[0].image_size)
        for k in instance_lists[0]._fields:
            vals = [i.get(k) for i in instance_lists]
            v0 = vals[0]
            ret.set(k, torch.cat(vals, 0) if isinstance(v0, torch.Tensor) else type(v0).cat(vals))
        return ret


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
