#!/usr/bin/env python
"""End-to-end timing of BASELINE.json configs[2]: the 3-clip loop of unified_loop_consistency.py with evolving 3D memory at
full size (576x1024, 25 frames per clip, 25 Euler steps), one MI355X: denoise loop on the HIP U-Net, VAE encode / decode and
CLIP on their HIP implementations (random-init full architectures), the reprojection stage (pano->pers, depth lift of 49
frames x 392x518, filter, splat into 24 x 6 x 512^2, cube->equirect 1000x2000, Pillow-exact resize) on its HIP kernels; the
depth network (VGGT-1B, SURVEY.md N4, out of scope) is the synthetic stand-in.  Prints one JSON line.
Usage: python bench_episode.py [--num_segments 3] [--steps 25]"""
import argparse
import json
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_segments", type=int, default=3)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    a = ap.parse_args()
    from evoworld_amd.inference import UnifiedLoopConsistencyPipeline
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from evoworld_amd.stages import HipStages
    from evoworld_amd.unet import UNetSpatioTemporalConditionModel
    import unified_loop_consistency as cli
    dev = "cuda"
    unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device=dev)
    pipe = StableVideoDiffusionPipeline(unet=unet)
    cam = cli.synthetic_episode(24 * a.num_segments + 8)
    st = HipStages(device=dev, camera_params=cam, depth_hw=(392, 518))
    marks = {}

    def timed(name, fn):
        def w(*x, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = fn(*x, **k)
            torch.cuda.synchronize(); marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
            return r
        return w
    # the pipeline owns VAE + CLIP (reference flow: aug-noise draw, then latents, from one generator per window)
    pipe.set_components(vae=st.vae, image_encoder=st.image_encoder)
    loop = UnifiedLoopConsistencyPipeline(pipe, timed("depth_standin", st.depth_model),
                                          height=a.height, width=a.width, num_frames=25, num_segments=a.num_segments,
                                          num_inference_steps=a.steps)
    pipe.denoise = timed("denoise", pipe.denoise)
    pipe.decode_latents = timed("vae_decode", pipe.decode_latents)
    st.vae.encode = timed("vae_encode", st.vae.encode)
    pipe._encode_image = timed("clip", pipe._encode_image)
    g = torch.Generator().manual_seed(0)
    start = (torch.rand(3, a.height, a.width, generator=g) * 2 - 1).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = loop.process_episode(start, cam)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    other = dt - sum(marks.values())
    print(json.dumps({"workload": f"configs[2]: {a.num_segments}-clip loop, {a.height}x{a.width}x25f, {a.steps} steps, evolving 3D memory",
                      "frames": int(frames.shape[0]), "seconds": round(dt, 3), "frames_per_s": round(frames.shape[0] / dt, 3),
                      "breakdown_s": {k: round(v, 3) for k, v in marks.items()} | {"reprojection+glue": round(other, 3)},
                      "finite": bool(torch.isfinite(frames).all())}))


if __name__ == "__main__":
    main()
