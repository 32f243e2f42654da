"""Drop-in for `mgnet.geometry` (mgnet/geometry/__init__.py:1-16): the same public names and signatures.

The per-pixel stages (Camera.reconstruct, Camera.project, view_synthesis) are differentiable wrappers over the C-ABI
entry points of csrc/geometry.hip; the B x 12 numbers of intrinsics/pose algebra and the one-line tensor helpers
(inv2depth, gradient_x/y, ...) are torch expressions on whatever device the caller's tensors live on.  Inside the
training step none of these run: MultiViewPhotometricLoss fuses all of them into one kernel (csrc/reproj_loss.hip)."""
from .camera import Camera
from .camera_utils import construct_K, scale_intrinsics, view_synthesis
from .depth import calc_smoothness, inv2depth
from .image import gradient_x, gradient_y, image_grid, interpolate_image, match_scales, meshgrid, same_shape
from .pose import Pose
from .pose_utils import euler2mat, invert_pose, pose_vec2mat

__all__ = [k for k in globals().keys() if not k.startswith("_")]
