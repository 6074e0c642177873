#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden vectors for BASELINE.json configs[0] (runs ONLY in the build container).

configs[0] = run_single_segment.sh on example/case_000 with 8 frames and 2 denoise steps: the single-segment path takes the episode's LAST
num_frames poses (dataset/CameraTrajDataset.py `reprojection` mode: rows 119..126 of camera_poses.txt, 1-based), flips Unity -> RDF
(utils/constant.py:3), scales the positions by pos_scale 0.1 (CameraTrajDataset.py:348), makes them relative to the window's first pose
(xyz_euler_to_three_by_four_matrix_batch, :643) and turns them into the Pluecker embedding at the 72x128 latent size
(utils/plucker_embedding.py:56,221).  This script imports exactly those reference functions (read-only, same stubs as make_goldens.py) and writes
inputs + expected outputs to tests/golden/config0_plucker.npz.  Data only: nothing from /root/reference is copied.

Usage:  python oracle/make_goldens_config0.py   (from the repo root)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, REF, _import_reference  # noqa: E402


def main():
    out = os.path.abspath(OUT)
    _import_reference()
    from dataset.CameraTrajDataset import xyz_euler_to_three_by_four_matrix_batch
    from utils.plucker_embedding import equirectangular_to_ray, ray_c2w_to_plucker
    rows = open(os.path.join(REF, "example/case_000/camera_poses.txt")).read().strip().split("\n")[1:]
    poses_unity = np.array([[float(v) for v in r.split(",")[1:]] for r in rows], dtype=np.float64)   # [126, 6]
    flip = np.array([1, -1, 1, -1, 1, -1], dtype=np.float64)
    p = poses_unity[-8:] * flip
    p[:, :3] *= 0.1
    cam = torch.tensor(p, dtype=torch.float32)
    c2w = xyz_euler_to_three_by_four_matrix_batch(cam, relative=True)
    pl = ray_c2w_to_plucker(torch.tensor(equirectangular_to_ray(72, 128)).float(), c2w)             # [8, 6, 72, 128]
    np.savez_compressed(os.path.join(out, "config0_plucker.npz"), poses_unity_last8=poses_unity[-8:], rows_1based=np.arange(119, 127),
                        cam=cam.numpy(), c2w=c2w.numpy(), plucker_f0_3_7=pl[[0, 3, 7]].numpy(),
                        plucker_rowsum=pl.double().sum(dim=(2, 3)).numpy(), plucker_abs_sum=np.array([pl.double().abs().sum().item()]))
    print("wrote config0_plucker.npz:", tuple(pl.shape), float(pl.abs().max()))


if __name__ == "__main__":
    main()
