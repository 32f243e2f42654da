"""MGNet meta-architecture and its three heads -- host-side mirror of mgnet/modeling/mg_net.py for the TRAINING
path (mg_net.py:220-373) with the same registries, `@configurable`/`from_config` protocol, attribute names
(=> state-dict keys) and loss dict keys, and for single-scale INFERENCE (mg_net.py:375-425): per-image
`sem_seg_postprocess`, panoptic fusion and DGC depth rescaling through mgnet_amd.postprocessing (HIP), and multi-scale +
flip inference (mg_net.py:427-520, row f4: `TEST.MSC_FLIP_EVAL`)."""
import contextlib
import os
from typing import Dict, List

import torch
from torch import nn

from ..data.metadata import MetadataCatalog
from ..events import get_event_storage
from ..registry import (DEPTH_HEADS_REGISTRY, INS_EMBED_HEADS_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY,
                        ShapeSpec, build_backbone, build_depth_head, build_ins_embed_head, build_sem_seg_head,
                        configurable)
from ..structures import ImageList
from . import ops
from .layers import GlobalContextModule, MGNetDecoder, MGNetHead, PoseCNN
from .loss import DeepLabCE, MultiViewPhotometricLoss, OhemCE

__all__ = ["MGNet", "INS_EMBED_HEADS_REGISTRY", "build_ins_embed_head", "DEPTH_HEADS_REGISTRY", "build_depth_head",
           "MGNetSemSegHead", "MGNetInsEmbedHead", "MGNetSelfSupervisedDepthHead"]


def _amp_dtype(cfg):
    """SOLVER.AMP.ENABLED -> the 16-bit activation format of the trunk.  detectron2's AMPTrainer (torch.cuda.amp) is IEEE fp16 +
    GradScaler (tools/train_net.py:162); this stack defaults to bf16 (no loss scaling needed, same MFMA rate) and runs the
    reference's fp16 + dynamic loss scaling with SOLVER.AMP.DTYPE "float16" (a key added by mgnet_amd)."""
    if not cfg.SOLVER.AMP.ENABLED:
        return None
    name = str(cfg.SOLVER.AMP.get("DTYPE", "bfloat16")) if hasattr(cfg.SOLVER.AMP, "get") else "bfloat16"
    return {"bfloat16": torch.bfloat16, "bf16": torch.bfloat16, "float16": torch.float16, "fp16": torch.float16}[name]


def _decoder_kwargs(node, input_shape, feature_node=None):
    feats = (feature_node or node).IN_FEATURES
    return dict(input_shape={k: v for k, v in input_shape.items() if k in feats}, common_stride=node.COMMON_STRIDE,
                arm_channels=node.ARM_CHANNELS, refine_channels=node.REFINE_CHANNELS, ffm_channels=node.FFM_CHANNELS,
                head_channels=node.HEAD_CHANNELS, init_method=node.INIT_METHOD)


@META_ARCH_REGISTRY.register()
class MGNet(nn.Module):
    @configurable
    def __init__(self, *, size_divisibility, pixel_mean, pixel_std, backbone, global_context, sem_seg_head,
                 ins_embed_head, depth_head, pose_net, with_panoptic, with_depth, with_uncertainty, msc_flip_eval=False,
                 amp_dtype=None, predict_instances=False, instance_post_proc_func=None, panoptic_post_proc_func=None,
                 depth_post_proc_func=None, **unused_inference_kwargs):
        super().__init__()
        self.size_divisibility = size_divisibility
        self._mean01, self._std01 = [float(x) / 255.0 for x in pixel_mean], [float(x) / 255.0 for x in pixel_std]  # host copies
        self.register_buffer("pixel_mean", torch.tensor([x / 255.0 for x in pixel_mean]).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor([x / 255.0 for x in pixel_std]).view(-1, 1, 1), False)
        self.backbone = backbone
        self.bb_features = list(backbone.output_shape().keys())
        self.global_context = global_context
        self.sem_seg_head, self.ins_embed_head = sem_seg_head, ins_embed_head
        self.depth_head, self.pose_net = depth_head, pose_net
        self.with_panoptic, self.with_depth, self.with_uncertainty = with_panoptic, with_depth, with_uncertainty
        if with_uncertainty:  # mg_net.py:104-107
            self.register_parameter("log_vars", nn.Parameter(torch.zeros(5), requires_grad=True))
        self.msc_flip_eval = msc_flip_eval
        self.predict_instances = predict_instances
        self.instance_post_proc_func = instance_post_proc_func
        self.panoptic_post_proc_func, self.depth_post_proc_func = panoptic_post_proc_func, depth_post_proc_func
        self.amp_dtype = amp_dtype  # activation dtype of the conv trunk (None = fp32); SOLVER.AMP.ENABLED -> bf16

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        shapes = backbone.output_shape()
        gcm = GlobalContextModule(in_channels=list(shapes.values())[-1].channels, out_channels=cfg.MODEL.GCM.GCM_CHANNELS,
                                  init_method=cfg.MODEL.GCM.INIT_METHOD)
        sem = ins = dep = pose = None
        if cfg.WITH_PANOPTIC:
            sem, ins = build_sem_seg_head(cfg, shapes), build_ins_embed_head(cfg, shapes)
        if cfg.WITH_DEPTH:
            dep, pose = build_depth_head(cfg, shapes), PoseCNN(cfg)
        meta = MetadataCatalog.get(cfg.DATASETS.TRAIN[0] if len(cfg.DATASETS.TRAIN) else "cityscapes")  # mg_net.py:147
        pan_fn = dep_fn = ins_fn = None
        if cfg.TEST.EVAL_INSTANCE:   # mg_net.py:145-153
            from functools import partial

            from ..postprocessing import get_instance_predictions
            ins_fn = partial(get_instance_predictions, thing_ids=list(meta.thing_dataset_id_to_contiguous_id.values()),
                             label_divisor=meta.label_divisor)
        if cfg.WITH_PANOPTIC:   # mg_net.py:155-170
            from ..postprocessing import get_panoptic_prediction
            pp = cfg.MODEL.POST_PROCESSING
            pan_kw = dict(num_thing_classes=len(meta.thing_dataset_id_to_contiguous_id.values()),
                          last_stuff_id=max(meta.stuff_dataset_id_to_contiguous_id.values()), label_divisor=meta.label_divisor,
                          stuff_area=pp.STUFF_AREA, void_label=-1, threshold=pp.CENTER_THRESHOLD, nms_kernel=pp.NMS_KERNEL)

            def pan_fn(sem_seg, center_heatmap, offsets):
                return get_panoptic_prediction(sem_seg, center_heatmap, offsets, **pan_kw)
        if cfg.WITH_DEPTH:      # mg_net.py:172-192
            from ..postprocessing import get_depth_prediction
            road = next((c["trainId"] * meta.label_divisor for c in meta.categories if c["name"] == "road"), None)
            ignore = [c["trainId"] * meta.label_divisor for c in meta.categories if c["name"] in cfg.INPUT.IGNORED_CATEGORIES_IN_DEPTH]
            dep_kw = dict(use_dgc_scaling=cfg.MODEL.POST_PROCESSING.USE_DGC_SCALING, road_class_id=road, depth_filter_class_ids=ignore)

            def dep_fn(**kw):
                return get_depth_prediction(**kw, **dep_kw)
        return dict(size_divisibility=cfg.MODEL.SIZE_DIVISIBILITY, pixel_mean=cfg.MODEL.PIXEL_MEAN,
                    pixel_std=cfg.MODEL.PIXEL_STD, backbone=backbone, global_context=gcm, sem_seg_head=sem,
                    ins_embed_head=ins, depth_head=dep, pose_net=pose, with_panoptic=cfg.WITH_PANOPTIC,
                    with_depth=cfg.WITH_DEPTH, with_uncertainty=cfg.WITH_UNCERTAINTY, msc_flip_eval=cfg.TEST.MSC_FLIP_EVAL,
                    amp_dtype=_amp_dtype(cfg), predict_instances=cfg.TEST.EVAL_INSTANCE, instance_post_proc_func=ins_fn,
                    panoptic_post_proc_func=pan_fn, depth_post_proc_func=dep_fn)

    @property
    def device(self):
        return self.pixel_mean.device

    # ---- batching helpers (mg_net.py:250-345) --------------------------------------------------------------
    def _stack(self, batched_inputs, key, scale=None, rgbx=False):
        ts = [x[key].to(self.device) for x in batched_inputs]
        d = self.size_divisibility
        if (scale is not None and ts[0].is_cuda and ts[0].dtype == torch.uint8
                and (d <= 1 or (ts[0].shape[-2] % d == 0 and ts[0].shape[-1] % d == 0))):
            from .. import _C
            # rgbx (opt-in, MGN_RGBX=1): the context frames of the reprojection loss as ONE pixel-interleaved [B,H,W,4] batch (one
            # 16-byte gather per bilinear corner).  Measured: -11 % kernel time when the warp is incoherent (random depths and
            # poses: 2.29 -> 2.04 ms), +3 % when it is close to the identity (what the benchmark's freshly initialised heads
            # predict: 1.98 -> 2.04 ms, the dword gathers of adjacent lanes then share cache lines) -- so the planar default stays
            out = _C.u8_frames_to_f32_rgbx(ts, scale) if rgbx and os.environ.get("MGN_RGBX") else None
            if out is None:
                out = _C.u8_frames_to_f32(ts, scale)   # [HIP] stack + `/ 255` in one pass (no padding needed)
            if out is not None:
                return out
        t = ImageList.from_tensors(ts, self.size_divisibility).tensor
        # `.float() / 255` (mg_net.py:250,320-335) once on the stacked batch instead of per frame: same values (the zero
        # padding stays zero), 2 launches instead of 2 per frame
        # (true division of the uint8 batch promotes to fp32 inside ONE kernel: same values as .float() / scale)
        return t if scale is None else (t / scale if not t.is_floating_point() else t.float() / scale)

    def _orig_frames_u8(self, batched_inputs):
        """The un-jittered frames of the photometric loss (mg_net.py:320-335: `uint8.float() / 255`, no mean / std) as ONE uint8 RGBX
        batch made by one launch; the loss kernels divide by 255 in registers (exactly rounded, so the values are the reference's).
        None when the frames are not uint8 CUDA tensors of a size that needs no padding (then the fp32 path assembles them);
        MGN_FRAMES_F32=1 forces the fp32 path."""
        keys = ("image_orig", "image_prev_orig", "image_next_orig")
        f0 = batched_inputs[0].get("image_orig")
        d = self.size_divisibility
        if (f0 is None or not f0.is_cuda or f0.dtype != torch.uint8 or os.environ.get("MGN_FRAMES_F32") or os.environ.get("MGN_RGBX")
                or (d > 1 and (f0.shape[-2] % d or f0.shape[-1] % d)) or 3 * len(batched_inputs) > 48):
            return None
        from .. import _C
        frames = [x[k] for k in keys for x in batched_inputs]
        out = _C.u8_frames_to_rgbx(frames)
        if out is None:
            return None
        B = len(batched_inputs)
        return {k: out[j * B:(j + 1) * B] for j, k in enumerate(keys)}

    def _to_device_async(self, t, slot=None):
        """Small host tensor -> device without stalling the host (see _C.PinnedStager)."""
        if t.device == self.device or self.device.type != "cuda":
            return t.to(self.device)
        from .. import _C
        st = self.__dict__.get("_stager")
        if st is None:
            st = self.__dict__["_stager"] = _C.PinnedStager()
        return st.stage(t, self.device, slot, data=True)

    def _net_input(self, batched_inputs, key):
        x = (self._stack(batched_inputs, key, 255.0) - self.pixel_mean) / self.pixel_std
        if self.amp_dtype is not None:
            x = x.to(self.amp_dtype)
        return x.contiguous(memory_format=torch.channels_last) if x.is_cuda else x.contiguous()

    def _side_streams(self):
        """Two side streams for the independent branches of the training step (pose network | backbone, then the three heads with
        their losses): most launches of the step are short and under-fill the chip one at a time, and every dependent launch
        costs ~3 us of dispatch latency -- concurrent branches hide both.  The autograd engine replays each node on the stream
        of its forward and orders the streams itself.  MGNET_STREAMS=0 keeps everything on the current stream."""
        if not (self.training and self.pixel_mean.is_cuda) or os.environ.get("MGNET_STREAMS", "1") == "0" or (getattr(self, "_no_side_streams", False) and not os.environ.get("MGN_GRAPH_STREAMS")):
            return None
        st = self.__dict__.get("_streams")
        if st is None:
            # (a third one for the pose network: MGNET_POSE_STREAM=1 -- its backward then is not queued behind the instance head's)
            st = self.__dict__["_streams"] = [torch.cuda.Stream(self.device) for _ in range(3 if os.environ.get("MGNET_POSE_STREAM") == "1" else 2)]
        return st

    def interleaved_trunks(self):
        """True when the training forward issues pose encoder and backbone block by block beside each other (both are the same
        ResNet; MGNET_INTERLEAVE=0 restores the reference's order: pose network first, as a whole)"""
        if not getattr(self, "with_depth", False) or os.environ.get("MGNET_INTERLEAVE", "1") == "0" or getattr(self, "pose_net", None) is None:
            return False
        pe, bb = self.pose_net.pose_encoder, self.backbone
        return (getattr(pe, "stage_names", None) == getattr(bb, "stage_names", 0)
                and all(len(getattr(pe, n)) == len(getattr(bb, n)) for n in bb.stage_names))

    def forward(self, batched_inputs):
        inputs, outputs, targets = {}, {}, {}
        side = self._side_streams()
        main = torch.cuda.current_stream() if side else None

        def on(k):
            return torch.cuda.stream(side[k]) if side else contextlib.nullcontext()

        def tensors(obj):
            if isinstance(obj, torch.Tensor):
                yield obj
            elif isinstance(obj, dict):
                for v in obj.values():
                    yield from tensors(v)
            elif isinstance(obj, (list, tuple)):
                for v in obj:
                    yield from tensors(v)
            elif hasattr(obj, "tensors"):       # lazy wrappers of ops.py (LazyUpsample ...) list what they hold
                yield from tensors(obj.tensors())

        def handover(src, dst, *objs):
            """stream `dst` continues after what `src` has been given so far and will read `objs` (made on `src`)"""
            if side:
                dst.wait_stream(src)
                if not torch.cuda.is_current_stream_capturing():   # (a capture's private pool never recycles memory between its nodes)
                    for t in tensors(objs):
                        if t.is_cuda:
                            t.record_stream(dst)

        fused_prep = self.pixel_mean.is_cuda and self.amp_dtype in (torch.bfloat16, torch.float16) and batched_inputs[0]["image"].dtype == torch.uint8
        pose_in = None
        if fused_prep:  # [HIP] uint8 frames -> normalised, channel-padded NHWC bf16 in one pass (csrc/prep.hip)
            from .. import _C
            mean, std = self._mean01, self._std01   # (host constants: reading the device buffers would sync every step)
            frames = [self._stack(batched_inputs, "image")]
            if self.training and self.with_depth:
                # the pose network's input is made on the pose network's stream, beside the backbone's (two HBM-bound launches, the larger
                # of which only the pose stem waits for)
                pframes = frames + [self._stack(batched_inputs, "image_prev"), self._stack(batched_inputs, "image_next")]
                if side:
                    handover(main, side[2 if len(side) > 2 else 0], pframes)
                with on(2 if (side and len(side) > 2) else 0):
                    pose_in = _C.prep_input(pframes, mean, std, 16, self.amp_dtype)   # channels: image, prev, next (:264)
            # 3 -> 4 channels where the dense-row stem kernel takes the shape (csrc/conv_stem.hip CP = 4), else 3 -> 8
            cp = _C.stem_input_channels(*[frames[0].shape[i] for i in (0, 2, 3)])
            inputs["image"] = _C.prep_input(frames, mean, std, cp, self.amp_dtype)
        else:
            inputs["image"] = self._net_input(batched_inputs, "image")
            if self.training and self.with_depth:
                inputs["image_prev"] = self._net_input(batched_inputs, "image_prev")
                inputs["image_next"] = self._net_input(batched_inputs, "image_next")
                pose_in = torch.cat(list(inputs.values()), 1)  # mg_net.py:264
        pk = 2 if (side and len(side) > 2) else 0   # the stream of the pose network
        features = None
        if pose_in is not None:
            if not fused_prep:
                handover(main, side[pk] if side else None, pose_in)
            pe, bb = self.pose_net.pose_encoder, self.backbone
            if self.training and self.interleaved_trunks():
                # The two ResNets issued block by block beside each other: autograd replays ready nodes in reverse creation order, so
                # their backward passes alternate as well.  With the pose network issued as a whole FIRST (mg_net.py:262-265 order) its
                # backward was replayed LAST, behind the backbone's: the final 5.6 ms of the step were one stream running the pose
                # encoder's small kernels alone (profiles/r05_critical_path.txt); side by side the step is 2.5 ms shorter (28.7 -> 26.1).
                with on(pk):
                    xp = pe.stem(pose_in)
                xb = bb.stem(inputs["image"])
                features = {"stem": xb} if "stem" in bb._out_features else {}
                for name in bb.stage_names:
                    for blk_p, blk_b in zip(getattr(pe, name), getattr(bb, name)):
                        with on(pk):
                            xp = blk_p(xp)
                        xb = blk_b(xb)
                    if name in bb._out_features:
                        features[name] = xb
                with on(pk):
                    outputs["poses"] = self.pose_net.head(xp)
            else:
                with on(pk):
                    outputs["poses"] = self.pose_net(pose_in)

        if self.msc_flip_eval and not self.training:   # mg_net.py:267-268
            norm = (self._stack(batched_inputs, "image", 255.0) - self.pixel_mean) / self.pixel_std
            return self._inference(batched_inputs, self.forward_multi_scale_flip(norm))
        if features is None:
            features = self.backbone(inputs["image"])
        features["global_context"] = self.global_context(features[self.bb_features[-1]])
        if not self.training:
            if self.with_panoptic:
                outputs["sem_seg"] = self.sem_seg_head(features)
                outputs["center"], outputs["offset"] = self.ins_embed_head(features)
            if self.with_depth:
                outputs["depth"] = self.depth_head(features)
            return self._inference(batched_inputs, outputs)

        if self.with_panoptic:
            targets.update({
                "sem_seg": self._stack(batched_inputs, "sem_seg"),
                "sem_seg_weights": self._stack(batched_inputs, "sem_seg_weights"),
                "center": self._stack(batched_inputs, "center").unsqueeze(1),
                "center_weights": self._stack(batched_inputs, "center_weights"),
                "offset": self._stack(batched_inputs, "offset"),
                "offset_weights": self._stack(batched_inputs, "offset_weights"),
            })
        if self.with_depth:
            orig = self._orig_frames_u8(batched_inputs)
            if orig is None:
                orig = {"image_orig": self._stack(batched_inputs, "image_orig", 255.0),  # NOT mean/std normalised (:320-335)
                        "image_prev_orig": self._stack(batched_inputs, "image_prev_orig", 255.0, rgbx=True),
                        "image_next_orig": self._stack(batched_inputs, "image_next_orig", 255.0, rgbx=True)}
            targets.update(orig)
            targets.update({
                "camera_matrix": self._to_device_async(torch.stack([x["camera_matrix"] for x in batched_inputs], 0), "camera_matrix"),
                "reprojection_mask": self._stack(batched_inputs, "reprojection_mask").unsqueeze(1),
            })

        # the three heads and their losses: semantic on the current stream, instance on side stream 0 (after the pose network),
        # depth on side stream 1 (its loss reads the poses of stream 0)
        losses = {}
        f_sem = f_ins = f_dep = features
        if self.with_panoptic and self.with_depth:
            # every feature map feeds all three heads: three aliases whose gradients one kernel sums (ops.fanout3)
            fan = {k: ops.fanout3(v) for k, v in features.items()}
            f_sem, f_ins, f_dep = ({k: t[j] for k, t in fan.items()} for j in range(3))
        if side:
            handover(main, side[0], f_ins, targets)
            handover(main, side[1], f_dep, targets)
        # Issue order (the host issues ~100 launches per head one after the other): the depth head FIRST -- its reprojection kernel is
        # the longest kernel of the forward and VALU-bound, so it should start while the other heads' MFMA-bound convolutions still have
        # work to overlap with, not after them (MGN_HEAD_ORDER=sem_first: the order of rounds 1-2).  The loss dictionary keeps the
        # reference's order (mg_net.py:290-358: sem_seg, center, offset, photometric, smoothness) -- the uncertainty weights are indexed by it.
        l_sem, l_ins, l_depth = {}, {}, {}
        depth_first = self.with_depth and os.environ.get("MGN_HEAD_ORDER", "depth_first") != "sem_first"
        # Uncertainty weighting (mg_net.py:360-372: loss_k <- tau_k * exp(-log_vars[k]) * loss_k + 0.5 * log_vars[k], tau = 1 for
        # loss_sem_seg, else 0.5; same scalar names, no .item() host syncs) PER HEAD, on the head's own stream, right behind its losses:
        # [HIP] one launch forward, one backward per head (ops.uncertainty_weighting).  Weighted in one launch over the whole dictionary
        # at the end of forward (rounds 1-4), every head's backward had to wait for the slowest head's forward -- on replay the instance
        # and semantic branches sat idle for 2.5 - 4 ms per step (profiles/r05_critical_path_interleaved.txt).  Task index = position in
        # the reference's loss dictionary (sem_seg, center, offset, photometric, smoothness).
        k0 = {"sem": 0, "ins": 1 if self.with_panoptic else 0, "depth": 3 if self.with_panoptic else 0}
        storage = get_event_storage() if self.with_uncertainty else None

        def weigh(part, first):
            if not self.with_uncertainty or not part:
                return
            weighted, raw, unc = ops.uncertainty_weighting(part, self.log_vars, k0=first)
            for key in list(part):
                storage.put_scalar(key + "_raw", raw[key])
                storage.put_scalar(key + "_uncertainty", unc[key])
                part[key] = weighted[key]

        def run_depth():
            with on(1):
                outputs["depth"] = self.depth_head(f_dep)
                if side:
                    handover(side[pk], side[1], outputs.get("poses"))
                l_depth.update(self.depth_head.losses(outputs, targets))
                weigh(l_depth, k0["depth"])

        if depth_first:
            run_depth()
        if self.with_panoptic:
            with on(0):
                outputs["center"], outputs["offset"] = self.ins_embed_head(f_ins)
                l_ins.update(self.ins_embed_head.losses(outputs, targets))
                weigh(l_ins, k0["ins"])
            outputs["sem_seg"] = self.sem_seg_head(f_sem)
            l_sem.update(self.sem_seg_head.losses(outputs, targets))
            weigh(l_sem, k0["sem"])
        if self.with_depth and not depth_first:
            run_depth()
        assert len(l_sem) <= 1 and len(l_ins) in (0, 2), "task indices of the uncertainty weights assume one semantic and two instance losses"
        for part in (l_sem, l_ins, l_depth):
            losses.update(part)
        if side:
            for st in side:
                handover(st, main, losses)
        return losses


def _as_net_input(self, x):
    """fp32 NCHW normalised frames -> what the backbone's stem consumes: under bf16 on the GPU the channels are zero-padded
    to 8 (the packed-tap stem kernel's layout, like csrc/prep.hip produces), channels-last."""
    if self.amp_dtype is not None:
        if x.is_cuda and self.amp_dtype in (torch.bfloat16, torch.float16):
            x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 8 - x.shape[1]))
        x = x.to(self.amp_dtype)
    return x.contiguous(memory_format=torch.channels_last) if x.is_cuda else x.contiguous()


def forward_multi_scale_flip(self, norm_images, scales=None, flip=True, _torch_formulation=False):
    """mg_net.py:427-520: average the raw predictions over rescaled (bilinear, align_corners=True) and horizontally
    flipped copies of the normalised frames; softmax probabilities for sem_seg, offsets rescaled by stride / scale and
    their x component negated for the flipped pass.
    CUDA (16-bit and fp32 trunks): [HIP] csrc/mscflip.hip -- one launch builds each pass's network input, one launch per head output folds
    upsample -> softmax | offset scaling | 1 / depth -> un-flip -> running sum (the last pass divides); no full-resolution torch op."""
    scales = [0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0] if scales is None else scales
    n_flip = 2 if flip else 1
    if norm_images.is_cuda and self.amp_dtype in (torch.bfloat16, torch.float16, None) and not _torch_formulation:   # (the flag: tests only)
        # (None: the fp32 trunk of the reference's PseudoLabelGeneration yamls, SOLVER.AMP.ENABLED False -- same kernels, fp32 maps)
        return _msc_flip_hip(self, norm_images.float().contiguous(), scales, n_flip)
    import torch.nn.functional as F
    up = lambda t, stride, scale: F.interpolate(t.float(), scale_factor=stride / scale, mode="bilinear", align_corners=True)
    avg = {"sem_seg": None, "center": None, "offset": None, "depth": None}

    def add(key, v):
        avg[key] = v if avg[key] is None else avg[key] + v
    for scale in scales:
        x = F.interpolate(norm_images, scale_factor=scale, mode="bilinear", align_corners=True)
        for f in range(n_flip):
            if f:
                x = torch.flip(x, dims=(3,))
            features = self.backbone(self._as_net_input(x))
            features["global_context"] = self.global_context(features[self.bb_features[-1]])
            if self.with_panoptic:
                r = torch.softmax(up(self.sem_seg_head.layers(features), self.sem_seg_head.common_stride, scale), 1)
                center, offset = self.ins_embed_head.layers(features)
                c = up(center, self.ins_embed_head.common_stride, scale)
                o = up(offset, self.ins_embed_head.common_stride, scale) * self.ins_embed_head.common_stride / scale
                if f:
                    r, c, o = torch.flip(r, dims=(3,)), torch.flip(c, dims=(3,)), torch.flip(o, dims=(3,))
                    o[:, 1, :, :] *= -1
                add("sem_seg", r)
                add("center", c)
                add("offset", o)
            if self.with_depth:
                d = 1.0 / up(self.depth_head.layers(features)[0], self.depth_head.common_stride, scale).clamp(min=1e-6)
                add("depth", torch.flip(d, dims=(3,)) if f else d)
    n = n_flip * len(scales)
    return {k: (v / n if v is not None else None) for k, v in avg.items()}


def _msc_flip_hip(self, norm, scales, n_flip):
    """the device path of forward_multi_scale_flip (see there)"""
    import math

    from .. import _C
    N, _, H, W = norm.shape
    n = n_flip * len(scales)
    acc = {}

    def fold(key, lr, mode, stride, scale, f, k):
        # output size of F.interpolate(lr, scale_factor=stride / scale) (floor of size * factor, evaluated in double like torch)
        oh, ow = int(math.floor(lr.shape[2] * (stride / scale))), int(math.floor(lr.shape[3] * (stride / scale)))
        if key not in acc:
            acc[key] = torch.empty((N, lr.shape[1], oh, ow), dtype=torch.float32, device=norm.device)
        if tuple(acc[key].shape[2:]) != (oh, ow):   # (the reference's `average + r` would raise the same way)
            raise RuntimeError(f"multi-scale inference: pass at scale {scale} yields {(oh, ow)} for '{key}', the running average is {tuple(acc[key].shape[2:])}")
        _C.msc_accumulate(acc[key], lr, mode, f, k == 0, stride=float(stride), scale=float(scale), divide=float(n) if k == n - 1 else 0.0)

    k = 0
    for scale in scales:
        h, w = int(math.floor(H * scale)), int(math.floor(W * scale))
        for f in range(n_flip):
            x = _C.msc_input(norm, h, w, f, self.amp_dtype or torch.float32)
            features = self.backbone(x)
            features["global_context"] = self.global_context(features[self.bb_features[-1]])
            if self.with_panoptic:
                fold("sem_seg", self.sem_seg_head.layers(features), "softmax", self.sem_seg_head.common_stride, scale, f, k)
                center, offset = self.ins_embed_head.layers(features)
                fold("center", center, "plain", self.ins_embed_head.common_stride, scale, f, k)
                fold("offset", offset, "offset", self.ins_embed_head.common_stride, scale, f, k)
            if self.with_depth:
                fold("depth", self.depth_head.layers(features)[0], "inv2depth", self.depth_head.common_stride, scale, f, k)
            k += 1
    return {key: acc.get(key) for key in ("sem_seg", "center", "offset", "depth")}


MGNet._as_net_input = _as_net_input
MGNet.forward_multi_scale_flip = forward_multi_scale_flip


def sem_seg_postprocess(result, img_size, output_height, output_width):
    """detectron2.modeling.postprocessing.sem_seg_postprocess (recalled): crop the padding away, resize [C,h,w] logits to
    the requested output resolution (bilinear, align_corners=False)."""
    result = result[:, :img_size[0], :img_size[1]]
    if tuple(result.shape[-2:]) == (output_height, output_width):
        return result   # (interpolating to the same size with align_corners=False is the identity)
    return torch.nn.functional.interpolate(result[None].float(), size=(output_height, output_width), mode="bilinear",
                                           align_corners=False)[0]


def _inference(self, batched_inputs, outputs):
    """mg_net.py:375-425: per image (the post-processing is not batched in the reference either)."""
    results = []
    for idx, inp in enumerate(batched_inputs):
        size = tuple(inp["image"].shape[-2:])
        height, width = inp.get("height", size[0]), inp.get("width", size[1])
        if self.with_panoptic:
            r = sem_seg_postprocess(outputs["sem_seg"][idx], size, height, width)
            c = sem_seg_postprocess(outputs["center"][idx], size, height, width)
            o = sem_seg_postprocess(outputs["offset"][idx], size, height, width)
            pan = self.panoptic_post_proc_func(sem_seg=r.argmax(dim=0, keepdim=True), center_heatmap=c, offsets=o)
            results.append({"sem_seg": r, "panoptic_seg": (pan, None)})
            if self.predict_instances:   # mg_net.py:394-402: instance segmentation evaluation, disabled by default
                from ..structures import Instances
                instances = self.instance_post_proc_func(sem_seg=r, center_heatmap=c, panoptic_image=pan)
                if len(instances) > 0:
                    results[-1]["instances"] = Instances.cat(instances)
        if self.with_depth:
            d = sem_seg_postprocess(outputs["depth"][idx], size, height, width)
            first = batched_inputs[0]   # sic: mg_net.py:409-414 read the camera of the FIRST input
            depth, xyz = self.depth_post_proc_func(
                depth_logits=d.unsqueeze(0),
                camera_matrix=first["camera_matrix"].unsqueeze(0) if "camera_matrix" in first else None,
                real_camera_height=first["camera_height"] if "camera_height" in first else None,
                panoptic_seg=results[-1]["panoptic_seg"][0] if self.with_panoptic else None)
            if self.with_panoptic:
                results[-1]["depth"] = (depth, xyz)
            else:
                results.append({"depth": (depth, xyz)})
    return results


MGNet._inference = _inference


@SEM_SEG_HEADS_REGISTRY.register()
class MGNetSemSegHead(MGNetDecoder):  # mg_net.py:523-610
    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, common_stride, arm_channels, refine_channels, ffm_channels,
                 head_channels, init_method, loss_weight, loss_type, loss_top_k, ohem_threshold, ohem_n_min,
                 ignore_value, num_classes):
        super().__init__(input_shape=input_shape, common_stride=common_stride, arm_channels=arm_channels,
                         refine_channels=refine_channels, ffm_channels=ffm_channels, init_method=init_method)
        self.ignore_value, self.loss_weight, self.loss_type = ignore_value, loss_weight, loss_type
        self.decoder_only = num_classes is None
        self.head = MGNetHead(ffm_channels, head_channels, num_classes, init_method)
        if loss_type == "cross_entropy":
            self.loss = DeepLabCE(ignore_label=ignore_value, top_k_percent_pixels=1.0)
            self._plain_ce = True
        elif loss_type == "hard_pixel_mining":
            self.loss = DeepLabCE(ignore_label=ignore_value, top_k_percent_pixels=loss_top_k)
        elif loss_type == "ohem":
            self.loss = OhemCE(ignore_label=ignore_value, ohem_threshold=ohem_threshold, n_min=ohem_n_min)
        else:
            raise ValueError("Unexpected loss type: %s" % loss_type)

    @classmethod
    def from_config(cls, cfg, input_shape):
        n = cfg.MODEL.SEM_SEG_HEAD
        ret = _decoder_kwargs(n, input_shape)
        ret.update(loss_weight=n.LOSS_WEIGHT, loss_type=n.LOSS_TYPE, loss_top_k=n.LOSS_TOP_K,
                   ohem_threshold=n.OHEM_THRESHOLD, ohem_n_min=n.OHEM_N_MIN, ignore_value=n.IGNORE_VALUE,
                   num_classes=n.NUM_CLASSES)
        return ret

    def forward(self, features):
        from .. import _C
        y = self.layers(features, keep_pad=self.training)   # (training: the channel-padded predictor output, read in place)
        if self.training and _C.upce_supported(ops.real_channels(y)):   # the loss kernel interpolates the low-res logits on the fly
            return ops.LazyUpsample(y, self.common_stride)
        return ops.upsample_bilinear(ops.real_channels(y), self.common_stride)

    def layers(self, features, keep_pad=False):
        y, _ = super().forward(features)
        return self.head(y, keep_pad=keep_pad)

    def losses(self, predictions, targets):
        if self.loss_type == "cross_entropy":  # nn.CrossEntropyLoss(mean, ignore_index): mean over non-ignored pixels
            ce = torch.nn.functional.cross_entropy(ops.materialize(predictions["sem_seg"]).float(), targets["sem_seg"],
                                                   ignore_index=self.ignore_value, reduction="mean")
            return {"loss_sem_seg": ce * self.loss_weight}
        loss = self.loss(predictions["sem_seg"], targets["sem_seg"], targets["sem_seg_weights"])
        return {"loss_sem_seg": loss * self.loss_weight}


@INS_EMBED_HEADS_REGISTRY.register()
class MGNetInsEmbedHead(MGNetDecoder):  # mg_net.py:621-715
    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, common_stride, arm_channels, refine_channels, ffm_channels,
                 head_channels, init_method, center_loss_weight, offset_loss_weight):
        super().__init__(input_shape=input_shape, common_stride=common_stride, arm_channels=arm_channels,
                         refine_channels=refine_channels, ffm_channels=ffm_channels, init_method=init_method)
        self.center_loss_weight, self.offset_loss_weight = center_loss_weight, offset_loss_weight
        self.center_head = MGNetHead(ffm_channels, head_channels, 1, init_method)
        self.offset_head = MGNetHead(ffm_channels, head_channels, 2, init_method)

    @classmethod
    def from_config(cls, cfg, input_shape):
        n = cfg.MODEL.INS_EMBED_HEAD
        ret = _decoder_kwargs(n, input_shape)
        ret.update(center_loss_weight=n.CENTER_LOSS_WEIGHT, offset_loss_weight=n.OFFSET_LOSS_WEIGHT)
        return ret

    def forward(self, features):
        center, offset = self.layers(features, keep_pad=self.training)
        lc = ops.LazyUpsample(center, self.common_stride)
        lo = ops.LazyUpsample(offset, self.common_stride, mult=float(self.common_stride))  # pixel offsets (:682-694)
        if self.training and ops.ins_losses_supported(lc, lo):
            return lc, lo
        return lc.materialize(), lo.materialize()

    def layers(self, features, keep_pad=False):
        y, _ = super().forward(features)
        c, y = self.center_head(y, keep_pad=keep_pad, with_skip=True)   # (y: alias whose gradient joins center_head's data gradient)
        center = ops.head_activation(c, "sigmoid")  # mg_net.py:694 (sigmoid_ before the upsample)
        return center, self.offset_head(y, keep_pad=keep_pad)

    def losses(self, predictions, targets):
        """[torch-staging] weighted MSE / L1 (mg_net.py:697-715) without the two `.sum() > 0` host syncs:
        sum/max(wsum, tiny) * (wsum > 0) is identical in value and gradient."""
        if isinstance(predictions["center"], ops.LazyUpsample):  # [HIP] fused upsampling + weighted MSE / L1
            l2 = ops.upsampled_ins_losses(predictions["center"], predictions["offset"], targets)
            return {"loss_center": l2[0] * self.center_loss_weight, "loss_offset": l2[1] * self.offset_loss_weight}
        cw, ow = targets["center_weights"], targets["offset_weights"]
        lc = ((predictions["center"].float() - targets["center"]) ** 2 * cw).sum()
        cws = cw.sum()
        lc = torch.where(cws > 0, lc / cws.clamp_min(1e-30), lc * 0)
        lo = ((predictions["offset"].float() - targets["offset"]).abs() * ow).sum()
        ows = ow.sum()
        lo = torch.where(ows > 0, lo / ows.clamp_min(1e-30), lo * 0)
        return {"loss_center": lc * self.center_loss_weight, "loss_offset": lo * self.offset_loss_weight}


@DEPTH_HEADS_REGISTRY.register()
class MGNetSelfSupervisedDepthHead(MGNetDecoder):  # mg_net.py:726-829
    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, common_stride, arm_channels, refine_channels, ffm_channels,
                 head_channels, init_method, msc_loss, loss):
        super().__init__(input_shape=input_shape, common_stride=common_stride, arm_channels=arm_channels,
                         refine_channels=refine_channels, ffm_channels=ffm_channels, init_method=init_method)
        self.n, self.msc_loss, self.loss = None, msc_loss, loss
        in_ch = [ffm_channels, arm_channels[1], arm_channels[0]] if self.training and msc_loss else [ffm_channels]
        self.heads = nn.ModuleList([MGNetHead(c, head_channels, 1, init_method) for c in in_ch])

    @classmethod
    def from_config(cls, cfg, input_shape):
        n = cfg.MODEL.DEPTH_HEAD
        loss = MultiViewPhotometricLoss(ssim_loss_weight=n.SSIM_LOSS_WEIGHT, photometric_loss_weight=n.PHOTOMETRIC_LOSS_WEIGHT,
                                        smoothing_loss_weight=n.SMOOTHING_LOSS_WEIGHT, automask_loss=n.AUTOMASK_LOSS,
                                        photometric_reduce_op=n.PHOTOMETRIC_REDUCE_OP, padding_mode=n.PADDING_MODE)
        ret = _decoder_kwargs(n, input_shape, feature_node=cfg.MODEL.INS_EMBED_HEAD)  # sic: mg_net.py:783-785
        ret.update(msc_loss=n.MSC_LOSS, loss=loss)
        return ret

    def forward(self, features):
        y = self.layers(features)
        s = self.common_stride
        strides = [s, 2 * s, 4 * s] if self.training and self.msc_loss else [s]
        inv_depths = [ops.upsample_bilinear(x, st) for x, st in zip(y, strides)]
        if not self.training:
            return 1.0 / inv_depths[0].clamp(min=1e-6)  # inv2depth, depth.py:15
        return inv_depths

    def layers(self, features):
        y, msc = super().forward(features)
        feats = [y, msc[1], msc[0]] if self.training and self.msc_loss else [y]
        # sigmoid / 0.5 -> inverse depth in (0, 2) (mg_net.py:819-823)
        return [ops.head_activation(head(f, keep_pad=self.training), "sigmoid2") for head, f in zip(self.heads, feats)]

    def losses(self, predictions, targets):  # fp32 by contract (custom_fwd(cast_inputs=float32), mg_net.py:827)
        return self.loss(predictions, targets)
