"""Pose -> camera-to-world matrices (host-side, tiny).  Same names / argument meaning as the reference:
dataset/CameraTrajDataset.py:643-779 (3x4) and utils/geometry.py:5-89 (4x4).  R = Rz*Ry*Rx, angles in degrees;
relative=True expresses every frame in the first frame's coordinates."""
import torch

UNITY_TO_OPENCV = (1, -1, 1, -1, 1, -1)  # utils/constant.py:3


def _rot(xyz_euler):
    rx, ry, rz = (xyz_euler[:, i:i + 1] * torch.pi / 180 for i in (3, 4, 5))
    zero, one = torch.zeros_like(rx), torch.ones_like(rx)
    n = xyz_euler.shape[0]
    Rx = torch.cat([one, zero, zero, zero, torch.cos(rx), -torch.sin(rx), zero, torch.sin(rx), torch.cos(rx)], 1).view(n, 3, 3)
    Ry = torch.cat([torch.cos(ry), zero, torch.sin(ry), zero, one, zero, -torch.sin(ry), zero, torch.cos(ry)], 1).view(n, 3, 3)
    Rz = torch.cat([torch.cos(rz), -torch.sin(rz), zero, torch.sin(rz), torch.cos(rz), zero, zero, zero, one], 1).view(n, 3, 3)
    return torch.bmm(Rz, torch.bmm(Ry, Rx))


def xyz_euler_to_three_by_four_matrix_batch(xyz_euler, relative=False, flatten=False, debug=False, euler_as_rotation=False):
    if euler_as_rotation:
        raise NotImplementedError("euler_as_rotation is not used on the inference path")
    n = xyz_euler.shape[0]
    R = _rot(xyz_euler)
    t = xyz_euler[:, :3].reshape(n, 3, 1)
    F = torch.cat([R, t], dim=2)
    if relative:
        R0inv = F[0:1, :, :3].transpose(1, 2).expand(n, -1, -1)
        F = torch.cat([torch.bmm(R0inv, F[:, :, :3]), torch.bmm(R0inv, F[:, :, 3:] - F[0:1, :, 3:])], dim=2)
    if debug:
        F = F[0].repeat(n, 1, 1)
    return F.reshape(n, 12) if flatten else F


def xyz_euler_to_four_by_four_matrix_batch(xyz_euler, relative=False, flatten=False, debug=False, euler_as_rotation=False):
    F = xyz_euler_to_three_by_four_matrix_batch(xyz_euler, relative=relative, debug=debug, euler_as_rotation=euler_as_rotation)
    n = F.shape[0]
    bottom = torch.tensor([0, 0, 0, 1], dtype=F.dtype, device=F.device).view(1, 1, 4).expand(n, -1, -1)
    F = torch.cat([F, bottom], dim=1)
    return F.reshape(n, 16) if flatten else F
