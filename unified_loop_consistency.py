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
    """camera_poses.txt -> [P,6] UNSCALED poses in the OpenCV convention (unified_loop_consistency.py:370-395)."""
    from evoworld_amd.dataset import load_camera_poses as _load
    return _load(episode_path)


def list_episodes(base_folder):
    """A folder holding camera_poses.txt is one episode; otherwise each sub-folder that holds one is
    (unified_loop_consistency.py `determine_data_config`)."""
    if os.path.isfile(os.path.join(base_folder, "camera_poses.txt")):
        return [base_folder]
    if not os.path.isdir(base_folder):
        return []
    return sorted(os.path.join(base_folder, d) for d in os.listdir(base_folder)
                  if os.path.isfile(os.path.join(base_folder, d, "camera_poses.txt")))


def synthetic_episode(n_poses=80):
    i = np.arange(n_poses, dtype=np.float64)
    return np.stack([0.04 * i * np.sin(i / 9), 0 * i, 0.04 * i * np.cos(i / 9), 0 * i, 95 + 3.6 * i, 0 * i], 1)


def _save_frames_u8(u8_hwc, d, start=0):
    from PIL import Image
    os.makedirs(d, exist_ok=True)
    for i, f in enumerate(u8_hwc.cpu().numpy()):
        Image.fromarray(f).save(os.path.join(d, f"{i + start + 1:03}.png"))


def run_single_segment(args, ep, pipe, unet, dev, out_dir, synthetic):
    """The reference's single-segment fast path (`run_single_segment`, unified_loop_consistency.py:513-535): the dataset in
    'reprojection' mode hands out the episode's LAST 25 frames / poses (positions x pos_scale) with the PRE-RENDERED memory
    panoramas [panorama/001.png] + rendered_panorama_vggt_open3d/*.png, and the batch goes through forward_evoworld.process_batch
    with mask_mem=False.  Counterparts: evoworld_amd.dataset.load_single_segment_batch -> evoworld_amd.inference.process_batch."""
    from evoworld_amd import ops
    from evoworld_amd.dataset import load_single_segment_batch
    from evoworld_amd.inference import process_batch
    from evoworld_amd.plucker import equirectangular_to_ray
    from evoworld_amd.stages import load_stages
    if synthetic:
        cam = synthetic_episode(max(args.num_frames, 25))
        traj = torch.tensor(cam[-args.num_frames:], dtype=torch.float32)
        traj[:, :3] *= 0.1
        g = torch.Generator().manual_seed(0)
        pix = (torch.rand(args.num_frames, 3, args.height, args.width, generator=g) * 2 - 1).to(dev)
        batch = {"pixel_values": pix[None], "cam_traj": traj[None], "memorized_pixel_values": torch.zeros_like(pix)[None]}
    else:
        cam = load_camera_poses(ep)
        batch = load_single_segment_batch(ep, args.height, args.width, dev, sequence_length=args.num_frames)
    stages = load_stages(args.stages, args, cross_attention_dim=unet._cfg["cross_attention_dim"], camera_params=cam)
    rays = torch.tensor(equirectangular_to_ray(args.height // 8, args.width // 8)).float().to(dev)
    args.mask_mem = False                                                                   # :532
    pipe.set_components(vae=stages.vae, image_encoder=stages.image_encoder, feature_extractor=stages.feature_extractor)
    torch.cuda.synchronize()
    t0 = time.time()
    # the pipeline encodes [first frame | memory] itself (aug-noise draw, VAE mode, CLIP) as the reference's does
    latents = process_batch(batch, args, pipe, rays, torch.float32, None, os.path.basename(ep), output_type="latent",
                            num_inference_steps=args.num_inference_steps)
    dec = pipe.decode_latents(latents, args.num_frames, 8)[0].permute(1, 0, 2, 3)                  # [T,3,H,W] in [-1,1]
    frames_u8 = ops.f32_chw_to_u8_hwc(dec.float().contiguous())
    torch.cuda.synchronize()
    dt = time.time() - t0
    if args.save_frames:
        _save_frames_u8(frames_u8, os.path.join(out_dir, "predictions"))                  # forward_evoworld.save_frames
        _save_frames_u8(ops.f32_chw_to_u8_hwc(batch["pixel_values"][0].float().contiguous()), os.path.join(out_dir, "predictions_gt"))
    return {"episode": os.path.basename(ep), "frames": int(frames_u8.shape[0]), "seconds": round(dt, 3), "mode": "single_segment"}


def run_episode(args, ep, pipe, unet, dev, out_dir, synthetic):
    """N segments with evolving 3D memory (`process_episode`, unified_loop_consistency.py:398-492)."""
    from evoworld_amd.dataset import load_complete_episode_batch
    from evoworld_amd.inference import UnifiedLoopConsistencyPipeline
    from evoworld_amd.stages import load_stages
    cam = synthetic_episode(24 * args.num_segments + 8) if synthetic else load_camera_poses(ep)     # UNSCALED
    stages = load_stages(args.stages, args, cross_attention_dim=unet._cfg["cross_attention_dim"], camera_params=cam)
    start = None if synthetic else load_complete_episode_batch(ep, args.height, args.width, dev, cam=cam)["first_frame"]
    if start is None:
        g = torch.Generator().manual_seed(0)
        start = (torch.rand(3, args.height, args.width, generator=g) * 2 - 1).to(dev)
    pipe.set_components(vae=stages.vae, image_encoder=stages.image_encoder, feature_extractor=stages.feature_extractor)
    loop = UnifiedLoopConsistencyPipeline(pipe, stages.depth_model, height=args.height,
                                          width=args.width, num_frames=args.num_frames, num_segments=args.num_segments,
                                          num_inference_steps=args.num_inference_steps)
    torch.cuda.synchronize()
    t0 = time.time()
    # the unscaled poses go in; process_episode derives the pos-scaled Navigator / Plücker path itself (pos_scale = 0.1);
    # --save_frames also writes the reference's per-segment dumps predictions_{seg}/ and perspective_look_at_center_{seg}/
    frames = loop.process_episode(start, cam, save_dir=out_dir if args.save_frames else None,
                                  save_segment_frames=args.save_frames)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if args.save_frames:
        _save_frames_u8(loop.last_frames_u8, os.path.join(out_dir, "predictions"))
    return {"episode": os.path.basename(ep), "frames": int(frames.shape[0]), "seconds": round(dt, 3)}


def main(argv=None):
    args = parse_arguments(argv)
    from evoworld_amd import distributed as D
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from evoworld_amd.unet import UNetSpatioTemporalConditionModel

    rank, world, local = D.init()
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    os.makedirs(args.save_dir, exist_ok=True)
    # rank 0 reads (or draws) the weights; the other ranks receive the packed set once over RCCL / xGMI
    sub = "unet" if os.path.isdir(os.path.join(args.unet_path, "unet")) else None
    if rank == 0 or world == 1:
        if args.random_init:
            unet = UNetSpatioTemporalConditionModel.from_random(seed=args.seed, device=dev, num_frames=args.num_frames)
        else:
            unet = UNetSpatioTemporalConditionModel.from_pretrained(args.unet_path, subfolder=sub, device=dev)
    else:
        cfg = {"num_frames": args.num_frames}
        cj = os.path.join(args.unet_path, sub or "", "config.json")
        if not args.random_init and os.path.exists(cj):
            from evoworld_amd.unet import DEFAULT_CONFIG
            cfg = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.load(open(cj)).items() if k in DEFAULT_CONFIG}
        unet = UNetSpatioTemporalConditionModel.from_zeros(device=dev, **cfg)
    if world > 1:
        unet.broadcast_weights(src=0)
    pipe = StableVideoDiffusionPipeline(unet=unet)

    episodes = list_episodes(args.base_folder)[args.start_idx: args.start_idx + args.num_data]
    synthetic = not episodes
    if synthetic:
        episodes = [f"synthetic_{i:03d}" for i in range(args.start_idx, args.start_idx + args.num_data)]
    mine = D.shard_clips(len(episodes), rank, world)
    report = []
    for idx in mine:
        ep = episodes[idx]
        out_dir = os.path.join(args.save_dir, os.path.basename(ep.rstrip("/")))
        os.makedirs(out_dir, exist_ok=True)
        if args.single_segment:
            rec = run_single_segment(args, ep, pipe, unet, dev, out_dir, synthetic)
        else:
            rec = run_episode(args, ep, pipe, unet, dev, out_dir, synthetic)
        rec["rank"] = rank
        report.append(rec)
        print(json.dumps(rec), flush=True)
    D.barrier()
    D.shutdown()
    return report


if __name__ == "__main__":
    main()
