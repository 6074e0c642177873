#!/usr/bin/env python3
"""SURVEY.md §8(d) measurement leg (i): BASELINE.json configs[0] -- the single-segment path with 8 frames and 2 denoise steps -- run COMPLETELY on the CPU
oracle (the fp32 restatement of the reference's diffusers pipeline: full-size SVD-Xtend U-Net, random init, B = 2 CFG rows, 72x128 latents), timed on
the host cores.  Conditioning latents / embedding are seeded random tensors (SURVEY's protocol for this config; VAE and CLIP are outside the denoise
loop), the Pluecker embedding comes from example/case_000 rows 119..126 (tests/golden/config0_plucker.npz holds the reference's own output for them).
Prints ONE JSON line.  A reported baseline, not a target.   Usage: python tools/cpu_config0.py [--threads 32]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    from evoworld_amd.geometry import xyz_euler_to_three_by_four_matrix_batch
    from evoworld_amd.plucker import equirectangular_to_ray
    from oracle.pipeline_ref import oracle_loop
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef
    cores = min(os.cpu_count() or 1, a.threads)
    torch.set_num_threads(cores)
    T, h, w = a.frames, 72, 128
    with torch.device("meta"):
        m = UNetSpatioTemporalConditionModelRef(num_frames=T)
    m = m.to_empty(device="cpu").eval()
    with torch.no_grad():
        gw = torch.Generator().manual_seed(1)
        for n, p in m.named_parameters():
            if p.ndim > 1:
                p.uniform_(-0.02, 0.02, generator=gw)
            else:
                p.fill_(1.0 if n.endswith("weight") else 0.0)
    g0 = np.load(os.path.join(ROOT, "tests", "golden", "config0_plucker.npz"))
    cam = torch.tensor(g0["cam"])[-T:]
    c2w = xyz_euler_to_three_by_four_matrix_batch(cam, relative=True)
    rays = torch.tensor(equirectangular_to_ray(h, w)).float()
    # CPU restatement of utils/plucker_embedding.py:221 (the product's version is a HIP kernel): d = R ray, m = t x d
    d = torch.einsum("nij,hwj->nhwi", c2w[:, :, :3], rays)
    t = c2w[:, :, 3][:, None, None, :].expand_as(d)
    pl = torch.cat([d, torch.cross(t, d, dim=-1)], dim=-1).permute(0, 3, 1, 2)[None].contiguous()
    g = torch.Generator().manual_seed(0)
    il = torch.randn(1, T + 1, 4, h, w, generator=g)
    ehs = torch.randn(1, 1, 1024, generator=g)
    lat0 = torch.randn(1, T, 4, h, w, generator=g)
    t0 = time.time()
    with torch.no_grad():
        out = oracle_loop(m, lat0, il, ehs, pl, T, a.steps, mask_mem=False)
    dt = time.time() - t0
    print(json.dumps({"config": f"configs[0]: single segment, {T} frames, {a.steps} denoise steps, 576x1024 (72x128 latents), full-size U-Net, CFG batch 2",
                      "backend": "fp32 CPU oracle (oracle/pipeline_ref.py + oracle/unet_ref.py)", "seconds": round(dt, 1), "frames_per_s": round(T / dt, 5),
                      "cores": cores, "host_cores": os.cpu_count(), "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
