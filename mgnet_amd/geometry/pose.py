"""mgnet/geometry/pose.py:9-95 -- a batch of rigid [4,4] transforms."""
import torch

from .pose_utils import invert_pose, pose_vec2mat

__all__ = ["Pose"]


class Pose:
    def __init__(self, mat):
        assert tuple(mat.shape[-2:]) == (4, 4)
        if mat.dim() == 2:
            mat = mat.unsqueeze(0)
        assert mat.dim() == 3
        self.mat = mat

    def __len__(self):
        return len(self.mat)

    @classmethod
    def identity(cls, N=1, device=None, dtype=torch.float):
        return cls(torch.eye(4, device=device, dtype=dtype).repeat([N, 1, 1]))

    @classmethod
    def from_vec(cls, vec, mode):
        """[B,6] pose vector -> Pose (pose.py:40-46)"""
        m34 = pose_vec2mat(vec, mode)
        mat = torch.eye(4, device=vec.device, dtype=vec.dtype).repeat([len(vec), 1, 1])
        mat[:, :3, :3] = m34[:, :3, :3]
        mat[:, :3, -1] = m34[:, :3, -1]
        return cls(mat)

    @property
    def shape(self):
        return self.mat.shape

    def item(self):
        return self.mat

    def repeat(self, *args, **kwargs):
        self.mat = self.mat.repeat(*args, **kwargs)
        return self

    def inverse(self):
        return Pose(invert_pose(self.mat))

    def to(self, *args, **kwargs):
        self.mat = self.mat.to(*args, **kwargs)
        return self

    def transform_pose(self, pose):
        """self * pose"""
        assert tuple(pose.shape[-2:]) == (4, 4)
        return Pose(self.mat.bmm(pose.item()))

    def transform_points(self, points):
        """R . X + t on [B,3,H,W] points (pose.py:77-83).  CUDA fp32: the affine-lift kernel with depth == 1 would need the
        pixel grid, so this is the projection kernel's sibling: one bmm over B x 3 x HW, as in the reference."""
        assert points.shape[1] == 3
        B, _, H, W = points.shape
        out = self.mat[:, :3, :3].bmm(points.view(B, 3, -1)) + self.mat[:, :3, -1].unsqueeze(-1)
        return out.view(B, 3, H, W)

    def __matmul__(self, other):
        if isinstance(other, Pose):
            return self.transform_pose(other)
        if isinstance(other, torch.Tensor):
            if other.shape[1] == 3 and other.dim() > 2:
                assert other.dim() == 3 or other.dim() == 4
                return self.transform_points(other)
            raise ValueError("Unknown tensor dimensions {}".format(other.shape))
        raise NotImplementedError()
