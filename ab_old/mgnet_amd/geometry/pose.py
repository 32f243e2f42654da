"""`Pose`: a batch of rigid transforms, the public name and call surface of mgnet/geometry/pose.py:9-95 (constructor on a
[B,4,4] / [4,4] tensor, `mat`, `identity`, `from_vec`, `shape`, `item`, `repeat`, `inverse`, `to`, `transform_pose`,
`transform_points`, `@`).

Host algebra only: a pose is 12 numbers per image and none of this runs per pixel inside the training step (the reprojection
kernel folds `K R Kinv` and `K t` once per image, csrc/reproj_loss.hip `reproj_prep`).  `transform_points` on a full point map is
the one per-pixel method; it is a single fused `baddbmm` (R X + t) over the flattened map, the product path never calls it
(`view_synthesis` goes through `mgn_view_synthesis_*`, which applies the pose inside the projection)."""
import torch

from .pose_utils import invert_pose, pose_vec2mat

__all__ = ["Pose"]


def _homogeneous(rt):
    """[B,3,4] (R | t) -> [B,4,4] with the constant last row (differentiable in rt; one concatenation)"""
    last = rt.new_tensor([0.0, 0.0, 0.0, 1.0]).expand(rt.shape[0], 1, 4)
    return torch.cat([rt[:, :3, :4], last], dim=1)


class Pose:
    def __init__(self, mat):
        if mat.dim() == 2:
            mat = mat[None]
        assert mat.dim() == 3 and tuple(mat.shape[-2:]) == (4, 4), f"Pose expects [B,4,4] or [4,4], got {tuple(mat.shape)}"
        self.mat = mat

    # ---- constructors ------------------------------------------------------------------------------------------------
    @classmethod
    def identity(cls, N=1, device=None, dtype=torch.float):
        return cls(torch.eye(4, device=device, dtype=dtype).expand(N, 4, 4).clone())

    @classmethod
    def from_vec(cls, vec, mode):
        """[B,6] (tx, ty, tz, rx, ry, rz) -> Pose; `mode` as pose_utils.pose_vec2mat ("euler")"""
        return cls(_homogeneous(pose_vec2mat(vec, mode)))

    # ---- container protocol ------------------------------------------------------------------------------------------
    def __len__(self):
        return self.mat.shape[0]

    @property
    def shape(self):
        return self.mat.shape

    @property
    def rotation(self):
        return self.mat[:, :3, :3]

    @property
    def translation(self):
        return self.mat[:, :3, 3]

    def item(self):
        return self.mat

    def _rebind(self, mat):   # `repeat` / `to` act in place on the object and return it (callers chain them)
        self.mat = mat
        return self

    def repeat(self, *sizes, **kw):
        return self._rebind(self.mat.repeat(*sizes, **kw))

    def to(self, *args, **kw):
        return self._rebind(self.mat.to(*args, **kw))

    # ---- algebra -----------------------------------------------------------------------------------------------------
    def inverse(self):
        return Pose(invert_pose(self.mat))

    def transform_pose(self, pose):
        """the compound transform self . pose"""
        other = pose.item() if isinstance(pose, Pose) else pose
        assert tuple(other.shape[-2:]) == (4, 4)
        return Pose(torch.bmm(self.mat, other))

    def transform_points(self, points):
        """R X + t for a point map [B,3,H,W] (or a point list [B,3,N])"""
        assert points.shape[1] == 3
        flat = points.flatten(2)
        moved = torch.baddbmm(self.translation.unsqueeze(-1), self.rotation, flat)
        return moved.view(points.shape)

    def __matmul__(self, other):
        if isinstance(other, Pose):
            return self.transform_pose(other)
        if not isinstance(other, torch.Tensor):
            raise NotImplementedError()
        if other.dim() in (3, 4) and other.shape[1] == 3:
            return self.transform_points(other)
        raise ValueError("Unknown tensor dimensions {}".format(other.shape))
