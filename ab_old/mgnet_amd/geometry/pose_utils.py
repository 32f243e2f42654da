"""mgnet/geometry/pose_utils.py:9-59 -- Euler pose vectors <-> matrices (B x 12 numbers: torch expressions)."""
import torch

__all__ = ["euler2mat", "pose_vec2mat", "invert_pose"]


def euler2mat(angle):
    """[B,3] (rx, ry, rz) -> [B,3,3] rotation R = Rx(rx) . Ry(ry) . Rz(rz)   (pose_utils.py:9-38)"""
    rx, ry, rz = angle.unbind(1)
    zero = rz.detach() * 0
    one = zero + 1

    def mat(rows):
        return torch.stack(rows, dim=1).view(-1, 3, 3)

    cz, sz = torch.cos(rz), torch.sin(rz)
    cy, sy = torch.cos(ry), torch.sin(ry)
    cx, sx = torch.cos(rx), torch.sin(rx)
    Rz = mat([cz, -sz, zero, sz, cz, zero, zero, zero, one])
    Ry = mat([cy, zero, sy, zero, one, zero, -sy, zero, cy])
    Rx = mat([one, zero, zero, zero, cx, -sx, zero, sx, cx])
    return Rx.bmm(Ry).bmm(Rz)


def pose_vec2mat(vec, mode="euler"):
    """[B,6] (tx,ty,tz,rx,ry,rz) -> [B,3,4] = [R | t]; mode None returns the input (pose_utils.py:41-51)"""
    if mode is None:
        return vec
    if mode != "euler":
        raise ValueError("Rotation mode not supported {}".format(mode))
    return torch.cat([euler2mat(vec[:, 3:]), vec[:, :3].unsqueeze(-1)], dim=2)


def invert_pose(T):
    """Rigid inverse of [B,4,4]: [R^T | -R^T t]   (pose_utils.py:54-59)"""
    Rt = T[:, :3, :3].transpose(-2, -1)
    out = torch.eye(4, device=T.device, dtype=T.dtype).repeat([len(T), 1, 1])
    out[:, :3, :3] = Rt
    out[:, :3, -1] = torch.bmm(-1.0 * Rt, T[:, :3, -1].unsqueeze(-1)).squeeze(-1)
    return out
