"""mgnet/geometry/camera.py:11-182 -- differentiable pinhole camera."""
import torch
import torch.nn as nn

from .camera_utils import scale_intrinsics
from .pose import Pose

__all__ = ["Camera"]


class _ReconstructFn(torch.autograd.Function):
    """[HIP] mgn_reconstruct_fwd/_bwd: points = depth * (A . [u,v,1]) + t"""

    @staticmethod
    def forward(ctx, depth, A, t):
        from .. import _C
        depth, A, t = depth.contiguous(), A.contiguous(), t.contiguous()
        ctx.save_for_backward(depth, A, t)
        return _C.reconstruct_fwd(depth, A, t)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        depth, A, t = ctx.saved_tensors
        d_depth, dA, dt = _C.reconstruct_bwd(depth, A, t, g.contiguous())
        return d_depth, dA, dt


class _ProjectFn(torch.autograd.Function):
    """[HIP] mgn_project_fwd/_bwd: normalised image coordinates of A . P + t"""

    @staticmethod
    def forward(ctx, points, A, t):
        from .. import _C
        points, A, t = points.contiguous(), A.contiguous(), t.contiguous()
        ctx.save_for_backward(points, A, t)
        return _C.project_fwd(points, A, t)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        points, A, t = ctx.saved_tensors
        d_points, dA, dt = _C.project_bwd(points, A, t, g.contiguous())
        return d_points, dA, dt


def _hip_only(x, what):
    if not (x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError(f"{what} runs on the HIP kernels only: expected a float32 CUDA tensor, got {x.dtype} on {x.device}")


class Camera(nn.Module):
    """K: [B,3,3] (or [B,4,4]) intrinsics; Tcw: camera -> world Pose (identity by default)."""

    def __init__(self, K, Tcw=None):
        super().__init__()
        self.K = K
        self.Tcw = Pose.identity(len(K), device=K.device, dtype=K.dtype) if Tcw is None else Tcw   # (the reference builds it on the CPU and relies on .to())
        self._Twc = None
        self._Kinv = None

    def __len__(self):
        return len(self.K)

    def to(self, *args, **kwargs):
        self.K = self.K.to(*args, **kwargs)
        self.Tcw = self.Tcw.to(*args, **kwargs)
        self._Twc = self._Kinv = None
        return self

    fx = property(lambda self: self.K[:, 0, 0])
    fy = property(lambda self: self.K[:, 1, 1])
    cx = property(lambda self: self.K[:, 0, 2])
    cy = property(lambda self: self.K[:, 1, 2])

    @property
    def Twc(self):
        """world -> camera (cached like the reference's lru_cache, camera.py:65-69)"""
        if self._Twc is None:
            self._Twc = self.Tcw.inverse()
        return self._Twc

    @property
    def Kinv(self):
        """closed-form inverse intrinsics: a CLONE of K with 1/fx, 1/fy, -cx/fx, -cy/fy written over four entries
        (camera.py:72-81; any skew entry of K therefore survives un-inverted)"""
        if self._Kinv is None:
            Kinv = self.K.clone()
            Kinv[:, 0, 0] = 1.0 / self.fx
            Kinv[:, 1, 1] = 1.0 / self.fy
            Kinv[:, 0, 2] = -1.0 * self.cx / self.fx
            Kinv[:, 1, 2] = -1.0 * self.cy / self.fy
            self._Kinv = Kinv
        return self._Kinv

    def scaled(self, x_scale, y_scale=None):
        if y_scale is None:
            y_scale = x_scale
        if x_scale == 1.0 and y_scale == 1.0:
            return self
        return Camera(scale_intrinsics(self.K.clone(), x_scale, y_scale), Tcw=self.Tcw)

    def reconstruct(self, depth, frame="w"):
        """depth [B,1,H,W] -> 3-D points [B,3,H,W] in the camera ('c') or world ('w') frame (camera.py:107-141)"""
        B, C, H, W = depth.shape
        assert C == 1
        _hip_only(depth, "Camera.reconstruct")
        Kinv = self.Kinv[:, :3, :3].float()
        if frame == "c":
            A, t = Kinv, Kinv.new_zeros(B, 3)
        elif frame == "w":
            T = self.Twc.item().to(depth.device).float()
            A, t = T[:, :3, :3].bmm(Kinv), T[:, :3, 3]
        else:
            raise ValueError("Unknown reference frame {}".format(frame))
        return _ReconstructFn.apply(depth, A, t)

    def project(self, X, frame="w"):
        """3-D points [B,3,H,W] -> grid_sample coordinates [B,H,W,2] in [-1,1] (camera.py:143-182)"""
        B, C, H, W = X.shape
        assert C == 3
        _hip_only(X, "Camera.project")
        K = self.K[:, :3, :3].float()
        if frame == "c":
            A, t = K, K.new_zeros(B, 3)
        elif frame == "w":
            T = self.Tcw.item().to(X.device).float()
            A, t = K.bmm(T[:, :3, :3]), K.bmm(T[:, :3, 3:]).squeeze(-1)
        else:
            raise ValueError("Unknown reference frame {}".format(frame))
        return _ProjectFn.apply(X, A, t)
