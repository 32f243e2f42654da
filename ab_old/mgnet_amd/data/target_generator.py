"""Panoptic training targets, generated ON THE DEVICE -- host-side mirror of mgnet/data/target_generator.py (same class
name, constructor arguments and `__call__(panoptic, segments_info)` contract) over `mgn_panoptic_targets`
(csrc/targets.hip).  Only the label image (4 B/px as ids, 3 B/px as the RGB PNG) and a few hundred table entries cross
PCIe instead of 32 B/px of finished maps, and the host no longer scans the label image once per segment.

`generate_batch` is the form the training loop uses: B label images in, the batched target tensors of
MGNet.forward (mg_net.py:278-349) out, plus the class part of `reprojection_mask` (dataset_mapper.py:214-216).
"""
import numpy as np
import torch

from .. import _C

__all__ = ["PanopticDeepLabTargetGenerator"]


class PanopticDeepLabTargetGenerator(object):
    def __init__(self, ignore_label, thing_ids, sigma=8, ignore_stuff_in_offset=False, small_instance_area=0,
                 small_instance_weight=1, ignore_crowd_in_semantic=False, *, depth_ignore_ids=(), promotion="auto",
                 device="cuda"):
        """First seven arguments: target_generator.py:15-24.  Extras (keyword-only):
        depth_ignore_ids -- contiguous class ids masked out of the photometric loss (dataset_mapper.py:119-124); when
            given, the result carries "reprojection_mask";
        promotion -- how `center - coordinate` (:143-144) is evaluated: "legacy" = float32 (NumPy < 2), "nep50" = float64
            then rounded (NumPy >= 2), "auto" = whatever the installed NumPy would make the reference compute."""
        self.ignore_label = int(ignore_label)
        self.thing_ids = sorted(list(thing_ids))
        self.ignore_stuff_in_offset = ignore_stuff_in_offset
        self.small_instance_area = small_instance_area
        self.small_instance_weight = small_instance_weight
        self.ignore_crowd_in_semantic = ignore_crowd_in_semantic
        self.sigma = int(sigma)
        if sigma != self.sigma or not 1 <= self.sigma <= 64:
            raise ValueError("sigma must be an integer in 1..64")
        if not 0 <= self.ignore_label <= 255:
            raise ValueError("ignore_label must fit the uint8 semantic map (target_generator.py:84)")
        if promotion == "auto":
            promotion = "nep50" if int(np.__version__.split(".")[0]) >= 2 else "legacy"
        if promotion not in ("legacy", "nep50"):
            raise ValueError(f"promotion: {promotion!r}")
        self.promotion = promotion
        self.depth_ignore_ids = [int(i) for i in depth_ignore_ids]
        self.device = torch.device(device)
        # the Gaussian patch exactly as the reference builds it (:45-50), rounded to the heat map's float32
        size = 6 * self.sigma + 3
        x = np.arange(0, size, 1, float)
        y = x[:, np.newaxis]
        x0, y0 = 3 * self.sigma + 1, 3 * self.sigma + 1
        self.g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * self.sigma ** 2))
        self._g_dev = None

    # ---- host side: segment tables -------------------------------------------------------------------------------
    def _tables(self, segments_infos):
        B = len(segments_infos)
        n_max = max([len(s) for s in segments_infos] + [1])
        if n_max > _C.TARGETS_MAX_SEGMENTS:
            raise ValueError(f"{n_max} segments in one image (limit {_C.TARGETS_MAX_SEGMENTS})")
        cap = 64
        while cap < n_max:
            cap *= 2
        tab = np.zeros((2, B, cap), dtype=np.int32)
        cnt = np.zeros((B,), dtype=np.int32)
        perms = []
        for b, segs in enumerate(segments_infos):
            ids = np.array([s["id"] for s in segs], dtype=np.int64).reshape(-1)
            if len(ids) and (ids.min() < -2 ** 31 or ids.max() >= 2 ** 31):
                raise ValueError("segment ids must fit int32")
            if len(np.unique(ids)) != len(ids):
                raise ValueError("segments_info holds the same id twice")
            cat = np.array([s["category_id"] for s in segs], dtype=np.int64).reshape(-1)
            if len(cat) and (cat.min() < 0 or cat.max() > 255):
                raise ValueError("category_id must fit the uint8 semantic map (target_generator.py:84)")
            crowd = np.array([1 if s["iscrowd"] else 0 for s in segs], dtype=np.int64).reshape(-1)
            thing = np.isin(cat, self.thing_ids).astype(np.int64)
            order = np.argsort(ids, kind="stable")
            tab[0, b, :len(ids)] = ids[order]
            tab[1, b, :len(ids)] = (cat | (crowd << 8) | (thing << 9))[order]
            cnt[b] = len(ids)
            perms.append(order)
        return tab, cnt, cap, perms

    def _cfg(self, B, H, W, pan_rgb, cap):
        if not self.thing_ids:
            raise IndexError("thing_ids is empty (target_generator.py:146 indexes thing_ids[0])")
        cfg = _C.TargetsCfg(B, H, W, int(pan_rgb), self.ignore_label, self.sigma, int(self.thing_ids[0]),
                            int(bool(self.ignore_stuff_in_offset)), int(self.small_instance_area),
                            int(self.small_instance_weight), int(bool(self.ignore_crowd_in_semantic)),
                            int(self.promotion == "legacy"), cap)
        for c in self.depth_ignore_ids:
            if 0 <= c <= 255:
                cfg.depth_ignore_mask[c >> 5] |= 1 << (c & 31)
        return cfg

    def _labels_to_device(self, panoptic):
        t = torch.as_tensor(np.ascontiguousarray(panoptic)) if isinstance(panoptic, np.ndarray) else panoptic
        rgb = t.dim() >= 3 and t.shape[-1] == 3 and t.dtype == torch.uint8
        if not rgb and t.dtype != torch.int32:
            t = t.to(torch.int32)
        return t.to(self.device, non_blocking=True).contiguous(), rgb

    # ---- device side -----------------------------------------------------------------------------------------------
    def generate_batch(self, panoptic, segments_infos, with_center_points=False):
        """panoptic: [B,H,W] integer ids or [B,H,W,3] uint8 RGB label images (numpy or tensor, host or device);
        segments_infos: list of B `segments_info` lists.  Returns the batched target tensors on the device."""
        if self.device.type != "cuda":
            raise RuntimeError("PanopticDeepLabTargetGenerator runs on the GPU (no CPU fallback by design)")
        pan, rgb = self._labels_to_device(panoptic)
        B, H, W = pan.shape[:3]
        if len(segments_infos) != B:
            raise ValueError("one segments_info list per label image")
        tab, cnt, cap, perms = self._tables(segments_infos)
        cfg = self._cfg(B, H, W, rgb, cap)
        if self._g_dev is None:
            self._g_dev = torch.from_numpy(self.g.astype(np.float32).reshape(-1)).to(self.device)
        meta = torch.from_numpy(np.concatenate([tab.reshape(-1), cnt])).to(self.device, non_blocking=True)
        seg_ids, seg_attr = meta[:B * cap].view(B, cap), meta[B * cap:2 * B * cap].view(B, cap)
        seg_count = meta[2 * B * cap:]
        out = _C.panoptic_targets(cfg, pan, seg_ids, seg_attr, seg_count, self._g_dev,
                                  want_mask=bool(self.depth_ignore_ids), want_points=with_center_points)
        if with_center_points:   # (y, x) per thing instance, in segments_info order (:117-118) -- a small D2H copy + sync
            pts = out.pop("center_points").cpu().numpy()
            out.pop("seg_area")
            lists = []
            for b in range(B):
                inv = np.empty(len(perms[b]), dtype=np.int64)
                inv[perms[b]] = np.arange(len(perms[b]))
                rows = pts[b][inv] if len(inv) else pts[b][:0]
                lists.append([[float(r[0]), float(r[1])] for r in rows if not np.isnan(r[0])])
            out["center_points"] = lists
        return out

    def __call__(self, panoptic, segments_info):
        """Per-image form with the reference's return dict (target_generator.py:147-156); tensors live on the device."""
        pan = panoptic[None] if isinstance(panoptic, (np.ndarray, torch.Tensor)) else np.asarray(panoptic)[None]
        out = self.generate_batch(pan, [segments_info], with_center_points=True)
        ret = {k: (v[0] if isinstance(v, (torch.Tensor, list)) else v) for k, v in out.items()}
        return ret
