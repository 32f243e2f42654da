"""Host-side mirror of mgnet/modeling/loss.py for the hot path: same class names, constructor arguments,
`forward(predictions, targets)` contract and error behaviour; the arithmetic runs in libmgnet_hip.so."""
import torch
import torch.nn as nn

from .. import _C

__all__ = ["DeepLabCE", "OhemCE", "MultiViewPhotometricLoss"]


def _pixel_ce(logits, labels, weights, ignore_label):
    """[torch-staging] per-pixel cross entropy (reduction none, ignore_index) times the per-pixel weights."""
    ce = torch.nn.functional.cross_entropy(logits.float(), labels, ignore_index=ignore_label, reduction="none")
    if weights is not None:
        ce = ce * weights
    return ce.contiguous().view(-1)


class DeepLabCE(nn.Module):
    """Hard pixel mining CE: mean of the top-k percent pixel losses (mirror of loss.py:9-42)."""

    def __init__(self, ignore_label=-1, top_k_percent_pixels=1.0, weight=None):
        super().__init__()
        assert weight is None, "class weights are not used by any MGNet config"
        self.top_k_percent_pixels = top_k_percent_pixels
        self.ignore_label = ignore_label

    def forward(self, logits, labels, weights=None):
        from . import ops
        if isinstance(logits, ops.LazyUpsample):  # [HIP] fused upsampling + CE (+ top-k)
            if self.top_k_percent_pixels == 1.0:
                return ops.upsampled_ce(logits, labels, weights, self.ignore_label, "mean")
            return ops.upsampled_ce(logits, labels, weights, self.ignore_label, "topk",
                                    n_sel=int(self.top_k_percent_pixels * labels.numel()))
        pixel_losses = _pixel_ce(logits, labels, weights, self.ignore_label)
        if self.top_k_percent_pixels == 1.0:
            return pixel_losses.mean()
        top_k_pixels = int(self.top_k_percent_pixels * pixel_losses.numel())
        return torch.topk(pixel_losses, top_k_pixels)[0].mean()


class OhemCE(nn.Module):
    """Online hard example mining CE (mirror of loss.py:45-81).  The reference sorts ALL pixel losses; the selection
    rule only needs (a) how many losses exceed the threshold and (b), when fewer than n_min+1 do, the n_min largest:
        sorted[n_min] > thr  <=>  count(loss > thr) > n_min      -> mean of {loss > thr}
        otherwise                                                 -> mean of the n_min largest
    which is result-identical without a full sort.  Like the reference it raises IndexError if n_min >= #pixels."""

    def __init__(self, ignore_label=-1, ohem_threshold=0.7, n_min=100000, weight=None):
        super().__init__()
        assert weight is None, "class weights are not used by any MGNet config"
        self.ohem_threshold = float(-torch.log(torch.tensor(ohem_threshold, dtype=torch.float)))
        self.n_min = n_min
        self.ignore_label = ignore_label

    def forward(self, logits, labels, weights=None):
        from . import ops
        if isinstance(logits, ops.LazyUpsample):  # [HIP] fused upsampling + CE + OHEM selection
            return ops.upsampled_ce(logits, labels, weights, self.ignore_label, "ohem", self.ohem_threshold, self.n_min)
        pixel_losses = _pixel_ce(logits, labels, weights, self.ignore_label)
        if self.n_min >= pixel_losses.numel():
            raise IndexError(f"index {self.n_min} is out of bounds for dimension 0 with size {pixel_losses.numel()}")
        hard = pixel_losses > self.ohem_threshold
        n_hard = hard.sum()
        if int(n_hard) > self.n_min:  # one host sync, as in the reference (loss.py:76)
            return (pixel_losses * hard).sum() / n_hard
        return torch.topk(pixel_losses, self.n_min)[0].mean()


class _ReprojLossFn(torch.autograd.Function):
    """losses[2] = f(inv_depth_0..n-1, poses | images, mask, K).  Gradients flow to the inverse depths and the
    poses only (reference: SURVEY 3.3 -- images, intrinsics and mask carry no gradient)."""

    @staticmethod
    def forward(ctx, cfg, img, prev, nxt, mask, cam, poses, *inv):
        want_grad = any(t.requires_grad for t in inv) or poses.requires_grad
        inv = [t.contiguous() for t in inv]
        poses = poses.contiguous()
        fwd = _C.reproj_loss_fwd(cfg, inv, img, prev, nxt, mask, cam, poses, want_grad=want_grad)
        # (the output tensor must not be reachable from ctx: output -> grad_fn -> ctx -> output would be a reference cycle that
        #  only the cyclic collector frees -- at an arbitrary later allocation, e.g. in the middle of a hipGraph capture)
        ctx.cfg, ctx.inv, ctx.img, ctx.mask = cfg, inv, img, mask
        ctx.fwd = {k: v for k, v in fwd.items() if k != "losses"}
        ctx.want_grad = want_grad
        ctx.used = False
        return fwd["losses"]

    @staticmethod
    def backward(ctx, grad_losses):
        if not ctx.want_grad:
            return (None,) * (7 + len(ctx.inv))
        if ctx.used:
            raise RuntimeError("MultiViewPhotometricLoss: backward called twice on the same forward; the fused "
                               "gradient buffers are consumed in place (re-run the forward instead of retain_graph)")
        ctx.used = True
        d_inv, d_pose = _C.reproj_loss_bwd(ctx.cfg, ctx.inv, ctx.img, ctx.mask, grad_losses.contiguous().float(), ctx.fwd)
        return (None, None, None, None, None, None, d_pose) + tuple(d_inv)


class MultiViewPhotometricLoss(nn.Module):
    """Drop-in for mgnet.modeling.loss.MultiViewPhotometricLoss (loss.py:84-154).

    predictions = {"depth": [inv_depth_i [B,1,H,W]], "poses": [B,2,6]}
    targets     = {"image_orig", "image_prev_orig", "image_next_orig": [B,3,H,W] in [0,1],
                   "camera_matrix": [B,4,4] (or [B,3,3]), optional "reprojection_mask": [B,1,H,W] bool}
                  The three frames may also arrive as the BYTES they were before mg_net.py:320-335 divided them by 255: uint8
                  [B,4,H,W] channels_last tensors (RGBX pixels, `_C.u8_frames_to_rgbx`); the kernels then convert in registers with
                  the exactly rounded byte / 255 -- same losses and gradients, a third of the gather instructions and of the bytes.
    returns     {"loss_photometric", "loss_smoothness"}  (already multiplied by their weights, loss.py:151-154)
    """

    def __init__(self, ssim_loss_weight, photometric_loss_weight, smoothing_loss_weight, automask_loss,
                 photometric_reduce_op, padding_mode):
        super().__init__()
        self.n = None
        self.ssim_loss_weight = ssim_loss_weight
        self.photometric_loss_weight = photometric_loss_weight
        self.smoothing_loss_weight = smoothing_loss_weight
        self.automask_loss = automask_loss
        self.photometric_reduce_op = photometric_reduce_op
        self.padding_mode = padding_mode
        self.prof_events = None  # optional (hipEvent_t begin, hipEvent_t end) for the next forward (bench.py roofline leg)
        if self.automask_loss:  # loss.py:105-109
            assert (
                self.photometric_reduce_op == "min"
            ), "For automasking only the min photometric_reduce_op is supported."

    def forward(self, predictions, targets):
        inv_depths = predictions["depth"]
        pose_results = predictions["poses"]
        self.n = len(inv_depths)
        assert pose_results.shape[1] == 2, "Context and poses lists must be of same length"  # loss.py:120
        img = targets["image_orig"]
        B, _, H, W = img.shape
        for x in inv_depths:  # match_scales would resize the image (image.py:101-135); the head already upsamples
            if tuple(x.shape[-2:]) != (H, W):
                raise NotImplementedError("inverse depths must already be at image resolution (mg_net.py:804-807)")
        cfg = _C.make_reproj_cfg(B, H, W, self.n, self.ssim_loss_weight, self.photometric_loss_weight,
                                 self.smoothing_loss_weight, self.automask_loss, self.photometric_reduce_op,
                                 self.padding_mode)
        if self.prof_events is not None:
            cfg.prof_begin, cfg.prof_end = self.prof_events
            self.prof_events = None
        mask = targets.get("reprojection_mask", None)
        if mask is not None:
            mask = mask.contiguous()
        if self.ssim_loss_weight == 0 and ((self.photometric_reduce_op == "min") == (mask is None)):
            # loss.py:196-197: the maps are then the 3-channel L1 maps; the reference's boolean indexing (loss.py:236-246) only works for
            # "min" WITH a reprojection mask and for "mean" WITHOUT one, and raises this error for the other two combinations
            raise IndexError("ssim_loss_weight=0: the shape of the mask does not match the shape of the indexed 3-channel L1 map "
                             "(photometric_reduce_op 'min' needs a reprojection_mask, 'mean' must not have one)")
        f32 = lambda t: t.float().contiguous()
        # frames may arrive pixel-interleaved ([B,4,H,W] channels_last, 4th channel unused; MGNet.forward produces them straight from
        # the uint8 frames): all three as uint8 RGBX (one 4-byte gather per bilinear corner), or the context frames as fp32 RGBx (16)
        nhwc4 = lambda t, dt: t.dim() == 4 and t.shape[1] == 4 and t.is_cuda and t.dtype == dt and t.is_contiguous(memory_format=torch.channels_last)
        prev, nxt = targets["image_prev_orig"], targets["image_next_orig"]
        if all(nhwc4(t, torch.uint8) for t in (img, prev, nxt)):
            pass
        else:
            if any(t.dtype == torch.uint8 for t in (img, prev, nxt)):
                raise ValueError("uint8 frames: image_orig, image_prev_orig and image_next_orig must ALL be uint8 [B,4,H,W] channels_last (RGBX)")
            img = f32(img)
            if not (nhwc4(prev, torch.float32) and nhwc4(nxt, torch.float32)):
                prev, nxt = f32(prev[:, :3]), f32(nxt[:, :3])
        losses = _ReprojLossFn.apply(cfg, img, prev, nxt,
                                     mask, f32(targets["camera_matrix"]), pose_results.float(),
                                     *[x.float() for x in inv_depths])
        return {"loss_photometric": losses[0], "loss_smoothness": losses[1]}
