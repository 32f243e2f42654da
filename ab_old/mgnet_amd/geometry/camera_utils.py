"""mgnet/geometry/camera_utils.py:10-55 -- intrinsics helpers and view synthesis."""
import torch

__all__ = ["construct_K", "scale_intrinsics", "view_synthesis"]


def construct_K(fx, fy, cx, cy, dtype=torch.float, device=None):
    return torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=dtype, device=device)


def scale_intrinsics(K, x_scale, y_scale):
    """in place, pixel-centre convention: f *= s, c = (c + 0.5) * s - 0.5 (camera_utils.py:15-21)"""
    K[..., 0, 0] *= x_scale
    K[..., 1, 1] *= y_scale
    K[..., 0, 2] = (K[..., 0, 2] + 0.5) * x_scale - 0.5
    K[..., 1, 2] = (K[..., 1, 2] + 0.5) * y_scale - 0.5
    return K


class _ViewSynthesisFn(torch.autograd.Function):
    """[HIP] mgn_view_synthesis_fwd/_bwd: lift with `depth`, move by the affine map (A, t), sample `ref`."""

    @staticmethod
    def forward(ctx, ref, depth, A, t, padding_mode):
        from .. import _C
        ref, depth, A, t = ref.contiguous(), depth.contiguous(), A.contiguous(), t.contiguous()
        ctx.save_for_backward(ref, depth, A, t)
        ctx.padding_mode = padding_mode
        return _C.view_synthesis_fwd(ref, depth, A, t, padding_mode)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        ref, depth, A, t = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("view_synthesis: no gradient with respect to the sampled image (the images are data on "
                                      "the training path, loss.py:111-154)")
        d_depth, dA, dt = _C.view_synthesis_bwd(ref, depth, A, t, g.contiguous(), ctx.padding_mode)
        return None, d_depth, dA.reshape(A.shape), dt, None


def view_synthesis(ref_image, depth, ref_cam, cam, mode="bilinear", padding_mode="zeros"):
    """Warp `ref_image` [B,C,H,W] into `cam`'s view through `depth` [B,1,H,W] of `cam` (camera_utils.py:24-55):
    cam.reconstruct(depth, 'w') -> ref_cam.project(., 'w') -> grid_sample(bilinear, align_corners=True), as ONE kernel.
    Differentiable with respect to depth and both cameras' intrinsics / poses."""
    assert depth.size(1) == 1
    if mode != "bilinear":
        raise NotImplementedError(f"view_synthesis: interpolation mode {mode!r} has no kernel (the reference only uses bilinear)")
    if not (ref_image.is_cuda and ref_image.dtype == torch.float32):
        raise RuntimeError("view_synthesis runs on the HIP kernels only: expected float32 CUDA tensors")
    # world = Twc_cam . (depth * Kinv_cam . [u,v,1]);  pixel_ref ~ K_ref . Tcw_ref . world
    T = ref_cam.Tcw.item().bmm(cam.Twc.item())                       # [B,4,4]
    KR = ref_cam.K[:, :3, :3].bmm(T[:, :3, :3])
    A = KR.bmm(cam.Kinv[:, :3, :3])
    t = ref_cam.K[:, :3, :3].bmm(T[:, :3, 3:]).squeeze(-1)
    return _ViewSynthesisFn.apply(ref_image, depth, A.float(), t.float(), padding_mode)
