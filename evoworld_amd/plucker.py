"""Plücker camera embedding.  Same call surface as utils/plucker_embedding.py:56 (equirectangular_to_ray, host
numpy, computed once per process) and :221 (ray_c2w_to_plucker, here one HIP kernel: ew_plucker_embed)."""
import numpy as np
import torch

from . import ops


def equirectangular_to_ray(target_H=576, target_W=1024):
    """(H, W, 3) unit ray directions, RDF, image centre = +Z  (utils/plucker_embedding.py:56-116)."""
    ys = np.arange(target_H, dtype=np.float32)
    xs = np.arange(target_W, dtype=np.float32)
    phi = (xs / target_W - 0.5) * 2.0 * np.pi
    theta = (ys / target_H - 0.5) * np.pi
    Phi, Theta = np.meshgrid(phi, theta)
    cosT = np.cos(Theta)
    return np.stack([cosT * np.sin(Phi), np.sin(Theta), cosT * np.cos(Phi)], axis=-1)


def ray_c2w_to_plucker(ray, c2w):
    """ray (H,W,3), c2w (N,3,4) -> (N,6,H,W) fp32 = [R d | t x (R d)] on the GPU (utils/plucker_embedding.py:221-255)."""
    ray = torch.as_tensor(ray)
    dev = c2w.device if c2w.is_cuda else (ray.device if ray.is_cuda else torch.device("cuda"))
    ray = ray.to(device=dev, dtype=torch.float32).contiguous()
    c2w = c2w.to(device=dev, dtype=torch.float32)[:, :3, :4].contiguous()
    return ops.plucker_embed(ray, c2w)
