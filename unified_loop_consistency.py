#!/usr/bin/env python
"""Drop-in entry point for the reference's `unified_loop_consistency.py` (CLI flags of :542-571, `main` :574-581) on the
MI355X-native hot path: U-Net denoising loop + reprojection on the device, one process per GPU, episodes sharded over ranks.

Same flags as the reference.  Additions (all optional):
  --stages pkg.mod:factory   provider of the out-of-scope networks (VAE, CLIP image encoder, VGGT); default = the synthetic,
                             weight-free stand-ins of `evoworld_amd.stages` so the script runs without checkpoints
  --random_init              random U-Net weights of the full architecture instead of loading --unet_path
  --num_inference_steps N    (reference fixes 25)
  --height/--width           (reference fixes 576x1024)

Launch on N GPUs:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 unified_loop_consistency.py ...
"""
import argparse
import json
import os
import time

import numpy as np
import torch


def parse_arguments(argv=None):
    p = argparse.ArgumentParser(description="Unified Loop Consistency Pipeline (MI355X-native hot path)")
    p.add_argument("--unet_path", type=str, required=True, help="Path to UNet model")
    p.add_argument("--svd_path", type=str, default="stabilityai/stable-video-diffusion-img2vid-xt-1-1", help="Path to SVD model")
    p.add_argument("--base_folder", type=str, default="data/Curve_Loop/test", help="Base folder containing episodes")
    p.add_argument("--save_dir", type=str, default="unified_output", help="Output directory")
    p.add_argument("--dataset_name", type=str, default="CameraTrajDataset", help="Dataset name")
    p.add_argument("--num_data", type=int, default=1, help="Number of episodes to process")
    p.add_argument("--start_idx", type=int, default=0, help="Start index")
    p.add_argument("--num_segments", type=int, default=3, help="Number of segments to process")
    p.add_argument("--num_frames", type=int, default=25, help="Frames per segment")
    p.add_argument("--save_frames", action="store_true", help="Save intermediate frames")
    p.add_argument("--curve_path", action="store_true", help="Use curve path navigation")
    p.add_argument("--seed", type=int, default=42, help="Random seed")
    p.add_argument("--single_segment", action="store_true", help="Use single segment fast path")
    # additions
    p.add_argument("--stages", type=str, default=None)
    p.add_argument("--random_init", action="store_true")
    p.add_argument("--num_inference_steps", type=int, default=25)
    p.add_argument("--height", type=int, default=576)
    p.add_argument("--width", type=int, default=1024)
    return p.parse_args(argv)


def load_camera_poses(episode_path):
    """camera_poses.txt: 'Frame,PosX,PosY,PosZ,RotX,RotY,RotZ' rows -> [P,6] in the OpenCV convention
    (unified_loop_consistency.py:370-395)."""
    from evoworld_amd.geometry import UNITY_TO_OPENCV
    f = os.path.join(episode_path, "camera_poses.txt")
    if not os.path.isfile(f):
        raise FileNotFoundError(f"camera_poses.txt not found under {episode_path}")
    rows = []
    for line in open(f):
        parts = [s.strip() for s in line.strip().split(",")]
        if len(parts) >= 7 and "rame" not in parts[0]:
            rows.append([float(x) for x in parts[1:7]])
    if not rows:
        raise ValueError(f"No valid camera pose rows parsed from {f}")
    return np.asarray(rows, dtype=float) * np.asarray(UNITY_TO_OPENCV, dtype=float)


def list_episodes(base_folder):
    """A folder holding camera_poses.txt is one episode; otherwise each sub-folder that holds one is
    (unified_loop_consistency.py `determine_data_config`)."""
    if os.path.isfile(os.path.join(base_folder, "camera_poses.txt")):
        return [base_folder]
    if not os.path.isdir(base_folder):
        return []
    return sorted(os.path.join(base_folder, d) for d in os.listdir(base_folder)
                  if os.path.isfile(os.path.join(base_folder, d, "camera_poses.txt")))


def load_start_image(episode_path, height, width, device):
    """panorama/001.png -> float [3,H,W] in [-1,1] (Resize + ToTensor + CustomRescale of the reference dataset)."""
    f = os.path.join(episode_path, "panorama", "001.png")
    if os.path.isfile(f):
        from PIL import Image
        from evoworld_amd import reprojection as RP
        u8 = torch.tensor(np.array(Image.open(f).convert("RGB"))).to(device)
        return RP.memory_to_pixel_values(u8[None], height, width)[0]
    g = torch.Generator().manual_seed(0)
    return (torch.rand(3, height, width, generator=g) * 2 - 1).to(device)


def synthetic_episode(n_poses=80):
    i = np.arange(n_poses, dtype=np.float64)
    return np.stack([0.04 * i * np.sin(i / 9), 0 * i, 0.04 * i * np.cos(i / 9), 0 * i, 95 + 3.6 * i, 0 * i], 1)


def main(argv=None):
    args = parse_arguments(argv)
    from evoworld_amd import distributed as D
    from evoworld_amd.inference import UnifiedLoopConsistencyPipeline
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from evoworld_amd.stages import load_stages
    from evoworld_amd.unet import UNetSpatioTemporalConditionModel

    rank, world, local = D.init()
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    os.makedirs(args.save_dir, exist_ok=True)
    if args.random_init:
        unet = UNetSpatioTemporalConditionModel.from_random(seed=args.seed, device=dev, num_frames=args.num_frames)
    else:
        sub = "unet" if os.path.isdir(os.path.join(args.unet_path, "unet")) else None
        unet = UNetSpatioTemporalConditionModel.from_pretrained(args.unet_path, subfolder=sub, device=dev)
    pipe = StableVideoDiffusionPipeline(unet=unet)

    episodes = list_episodes(args.base_folder)[args.start_idx: args.start_idx + args.num_data]
    synthetic = not episodes
    if synthetic:
        episodes = [f"synthetic_{i:03d}" for i in range(args.start_idx, args.start_idx + args.num_data)]
    mine = D.shard_clips(len(episodes), rank, world)
    report = []
    for idx in mine:
        ep = episodes[idx]
        cam = synthetic_episode(24 * args.num_segments + 8) if synthetic else load_camera_poses(ep)
        stages = load_stages(args.stages, args, cross_attention_dim=unet._cfg["cross_attention_dim"], camera_params=cam)
        start = load_start_image(ep, args.height, args.width, dev)
        n_seg = 1 if args.single_segment else args.num_segments
        loop = UnifiedLoopConsistencyPipeline(pipe, stages.depth_model, stages.frames_from_latents, height=args.height,
                                              width=args.width, num_frames=args.num_frames, num_segments=n_seg,
                                              num_inference_steps=args.num_inference_steps)
        out_dir = os.path.join(args.save_dir, os.path.basename(ep.rstrip("/")))
        os.makedirs(out_dir, exist_ok=True)
        torch.cuda.synchronize()
        t0 = time.time()
        frames = loop.process_episode(start, cam, stages.image_latents_fn, save_dir=out_dir if args.save_frames else None)
        torch.cuda.synchronize()
        dt = time.time() - t0
        if args.save_frames:
            from PIL import Image
            d = os.path.join(out_dir, "predictions")
            os.makedirs(d, exist_ok=True)
            u8 = ((frames / 2 + 0.5).clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
            for i, f in enumerate(u8):
                Image.fromarray(f).save(os.path.join(d, f"{i + 1:03}.png"))
        report.append({"episode": os.path.basename(ep), "frames": int(frames.shape[0]), "seconds": round(dt, 3), "rank": rank})
        print(json.dumps(report[-1]), flush=True)
    D.barrier()
    D.shutdown()
    return report


if __name__ == "__main__":
    main()
