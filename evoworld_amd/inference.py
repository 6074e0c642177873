"""Inference drivers of the hot path -- the build's counterparts of the reference callers:

  prepare_batch_data / process_batch     <- evoworld/inference/forward_evoworld.py:119-211          (C1, single clip)
  Navigator.move_forward / navigate_curve_path / split_curve_into_segments / extend_segment
                                         <- evoworld/inference/navigator_evoworld.py:146-154,173-231,303-318,394-448  (C2)
  UnifiedLoopConsistencyPipeline.process_episode / convert_pano_to_pers
                                         <- unified_loop_consistency.py:299-334,398-492               (C3, N-segment loop)

Same tensor preparation and pipeline kwargs (decode_chunk_size=8, motion_bucket_id=127, fps=7, noise_aug_strength=0.02,
mask_mem, per-window generator re-seeded with torch.manual_seed(-1)).  Differences, all device-side: frames stay tensors
(no PIL / PNG / numpy round-trips between the stages, SURVEY.md §3.2 'a native design keeps all of this on-device'); the
depth network (VGGT-1B, out of scope) is a duck-typed callable `depth_model(perspective_frames_u8) -> predictions dict`.
File output (PNG dumps) is optional and off the hot path.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import reprojection as RP
from .geometry import UNITY_TO_OPENCV, xyz_euler_to_four_by_four_matrix_batch, xyz_euler_to_three_by_four_matrix_batch
from .plucker import equirectangular_to_ray, ray_c2w_to_plucker


# ------------------------------------------------------------------ C1: single clip
def prepare_batch_data(batch, args, rays, weight_dtype=torch.float32):
    """batch: pixel_values [B,T,3,H,W] in [-1,1], cam_traj [B,T,6] (RDF, pos-scaled), memorized_pixel_values [B,T,3,H,W]
    -> (first_frame, camera_traj [B,T,3,4], plucker_embedding [B,T,6,H/8,W/8], memorized_pixel_values, images)."""
    images = batch["pixel_values"]
    first_frame = images[:, 0].cuda()
    raw = batch["cam_traj"].cuda()
    B = raw.shape[0]
    camera_traj = torch.zeros(B, args.num_frames, 3, 4, dtype=weight_dtype, device="cuda")
    plucker = torch.zeros(B, args.num_frames, 6, args.height // 8, args.width // 8, dtype=weight_dtype, device="cuda")
    for i in range(B):
        camera_traj[i] = xyz_euler_to_three_by_four_matrix_batch(raw[i], relative=True)
        plucker[i] = ray_c2w_to_plucker(rays, camera_traj[i])
    return first_frame, camera_traj, plucker, batch["memorized_pixel_values"].cuda(), images


def process_batch(batch, args, pipeline, rays, weight_dtype=torch.float32, output_path=None, episode="episode", **pipe_kw):
    """One clip through the pipeline with the reference's fixed kwargs; returns the pipeline's frames (and saves PNGs when
    output_path is given and the frames are PIL images)."""
    first_frame, _traj, plucker, memory, images = prepare_batch_data(batch, args, rays, weight_dtype)
    frames = pipeline(first_frame, height=args.height, width=args.width, num_frames=args.num_frames, decode_chunk_size=8,
                      motion_bucket_id=127, fps=7, noise_aug_strength=0.02, plucker_embedding=plucker,
                      memorized_pixel_values=memory, mask_mem=args.mask_mem, **pipe_kw).frames
    if output_path and isinstance(frames, list):
        d = os.path.join(output_path, episode, "predictions")
        os.makedirs(d, exist_ok=True)
        for i, f in enumerate(frames[0]):
            f.save(os.path.join(d, f"{i + 1:03}.png"))
    return frames


# ------------------------------------------------------------------ C2: windowed navigation
class Navigator:
    def __init__(self, pipe, height=576, width=1024, num_frames=25, fps=7, step_size=0.4, position_scale=0.1):
        self.pipe, self.model_height, self.model_width, self.num_frames, self.fps = pipe, height, width, num_frames, fps
        self.step_size, self.position_scale = step_size, position_scale            # navigator_evoworld.py:49-52
        self.rays = torch.tensor(equirectangular_to_ray(height // 8, width // 8)).float().cuda()
        self.memorized_images = None
        self.generations = []

    split_curve_into_segments = staticmethod(RP.split_curve_into_segments)

    def extend_segment(self, segment, n):
        """Extrapolate a short window to n poses (navigator_evoworld.py:132-172): a 1-pose window steps forward by
        step_size*position_scale along its yaw; a longer one continues with the last xyz increment, rotation held (the
        reference asserts the last two rotations agree)."""
        if len(segment) == 0:
            return segment
        seg = segment if isinstance(segment, torch.Tensor) else torch.stack(list(segment), dim=0)
        if len(seg) == 1:
            roty = seg[0][4]
            dz = self.step_size * torch.cos(torch.deg2rad(roty)) * self.position_scale
            dx = self.step_size * torch.sin(torch.deg2rad(roty)) * self.position_scale
            step = torch.stack([dx, torch.zeros_like(dx), dz, torch.zeros_like(dx), torch.zeros_like(dx), torch.zeros_like(dx)]).to(seg)
            return torch.cat([seg] + [seg[-1:] + step[None] * (i + 1) for i in range(n - 1)], dim=0)
        if len(seg) < n:
            last, prev = seg[-1], seg[-2]
            if not torch.allclose(last[3:], prev[3:]):
                raise AssertionError("The rotation of the last two steps are not the same.")
            delta = last - prev
            extra = torch.stack([last + delta * (i + 1) for i in range(n - len(seg))], dim=0)
            return torch.cat([seg, extra], dim=0)
        return seg

    def move_forward(self, image, segment, num_model_frames=25, num_inference_steps=25, noise_aug_strength=0.02,
                     use_memory=False, **pipe_kw):
        """One 25-pose window (navigator_evoworld.py:173-231): relative c2w -> Plücker -> pipeline with a CPU generator
        re-seeded per window (every segment starts from identical noise), mask_mem = not use_memory."""
        n = len(segment)
        if n < num_model_frames:
            segment = self.extend_segment(segment, num_model_frames)
        raw = segment if isinstance(segment, torch.Tensor) else torch.stack(list(segment), dim=0)
        raw = raw.cuda().float()
        c2w = xyz_euler_to_three_by_four_matrix_batch(raw, relative=True)
        pl = ray_c2w_to_plucker(self.rays, c2w)
        generator = torch.manual_seed(-1)                                   # navigator_evoworld.py:198
        frames = self.pipe(image.unsqueeze(0), num_frames=self.num_frames, width=self.model_width, height=self.model_height,
                           decode_chunk_size=8, generator=generator, motion_bucket_id=127, fps=self.fps,
                           num_inference_steps=num_inference_steps, noise_aug_strength=noise_aug_strength,
                           plucker_embedding=pl[:25].unsqueeze(0), memorized_pixel_values=self.memorized_images.clone(),
                           mask_mem=not use_memory, **pipe_kw).frames
        return frames, n

    @staticmethod
    def _last_frame_tensor(frames, n):
        """the window's last generated frame as the next window's start image (navigator :436-437: movement[-1] through
        the ToTensor + rescale transform)"""
        if isinstance(frames, list):                      # PIL clips [[img]*T]
            arr = np.asarray(frames[0][n - 1].convert("RGB"))
            return (torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255.0 * 2 - 1).cuda()
        raise NotImplementedError("chaining windows needs decoded frames (output_type='pil'); with output_type='latent' "
                                  "run one window per call (infer_segment=True), as process_episode does")

    def navigate_curve_path(self, path, start_image, num_inference_steps=25, memorized_images=None, infer_segment=False,
                            segment_id=None, **pipe_kw):
        """Windows [0:25],[24:49],... ; with infer_segment only window `segment_id` is generated (navigator :394-448);
        otherwise every window starts from the previous window's last frame."""
        self.memorized_images = memorized_images.clone()
        segments = self.split_curve_into_segments(path)
        generations, current = [], 0
        image = start_image
        for segment in segments:
            if segment_id is not None and current < segment_id and infer_segment:
                current += 1
                continue
            if len(segment) != 0:
                frames, n = self.move_forward(image, segment, num_inference_steps=num_inference_steps,
                                              use_memory=(segment_id != 0), **pipe_kw)
                generations.append((frames, n))
                last_window = (infer_segment and current + 1 > segment_id) or segment is segments[-1]
                if not last_window:
                    image = self._last_frame_tensor(frames, n)
            current += 1
            if infer_segment and current > segment_id:
                break
        self.generations = generations
        return generations


# ------------------------------------------------------------------ C3: N-segment loop with evolving 3D memory
class UnifiedLoopConsistencyPipeline:
    """process_episode of unified_loop_consistency.py:398-492 with every stage on the device.
    `frames_from_latents(latents[1,T,4,h,w]) -> float [T,3,H,W] in [-1,1]` stands for the VAE decode (row N1);
    `depth_model(persp_u8 [F,384,512,3]) -> dict(depth, depth_conf, images, extrinsic, intrinsic)` stands for VGGT (row N4)."""

    def __init__(self, pipeline, depth_model, frames_from_latents=None, height=576, width=1024, num_frames=25, num_segments=3,
                 num_inference_steps=25, pano_size=(1000, 2000), face_res=512):
        self.nav = Navigator(pipeline, height, width, num_frames)
        self.depth_model, self.frames_from_latents = depth_model, frames_from_latents
        self.height, self.width, self.num_frames, self.num_segments = height, width, num_frames, num_segments
        self.steps = num_inference_steps
        self.equi2pers = RP.Equi2Pers(height=384, width=512, fov_x=90.0, mode="bilinear")      # :178-183
        self.pano_size, self.renderer = pano_size, RP.CubemapRenderer(face_res=face_res)

    def convert_pano_to_pers(self, frames_u8, camera_params, segment_id):
        """frames uint8 [F,H,W,3] (device; the 8-bit frames the reference holds as PIL images) -> uint8 [F,384,512,3] +
        target yaws in degrees (:299-334).  camera_params: UNSCALED poses (camera_poses.txt)."""
        yaws = RP.calculate_target_yaws(camera_params, frames_u8.shape[0], segment_id)
        pers = self.equi2pers.batch(frames_u8, [{"pitch": 0, "roll": 0, "yaw": float(y)} for y in yaws])
        return pers, yaws / np.pi * 180.0

    def process_episode(self, start_image, camera_params, image_latents_fn=None, save_dir=None, pos_scale=0.1,
                        save_segment_frames=False, **pipe_kw):
        """start_image float [3,H,W] in [-1,1]; camera_params [P,6] numpy: the UNSCALED RDF poses of camera_poses.txt
        (unified_loop_consistency.py:370-395), used as they are for the target yaws and the reprojection alignment; the
        Navigator / Plücker path gets the copy with xyz * pos_scale that the dataset hands out as batch['cam_traj']
        (dataset/CameraTrajDataset.py:223,348).

        Default (image_latents_fn=None, the reference's flow): the PIPELINE owns `vae` and `image_encoder` and every window
        is one `pipe(image, generator=torch.manual_seed(-1), memorized_pixel_values=...)` call, so the generator's first draw
        is the [1+T,3,H,W] augmentation noise and its second the latents (pipeline_evoworld.py:596-600, then :663-673 ->
        :401-435; navigator_evoworld.py:198) -- same seed, same noise as the reference.  Frames are decoded by the
        pipeline's `decode_latents` (chunks of 8).
        With `image_latents_fn(first_frame, memory [T,3,H,W]) -> dict(image_latents=[1,1+T,4,h,w], image_embeddings=[1,1,X])`
        (stand-ins for VAE-encode + CLIP; decode through `frames_from_latents`) the conditioning is injected; the pipeline
        still consumes draw #1 from the generator so that the latents stay the generator's second draw.
        Every generated frame is carried as the 8-bit image the reference's PIL frames hold (:418-419, navigator :214-226).
        With save_dir and save_segment_frames, the per-segment dumps of :432-453 are written: predictions_{seg}/NNN.png
        (the segment's new frames, NNN continuing at seg*(T-1)+1) and perspective_look_at_center_{seg}/NNN.png (the
        pano->pers views fed to the depth network).  Returns all generated frames float [N,3,H,W] in [-1,1] (25 -> 49 -> 73 ...) on the 8-bit grid."""
        from . import ops
        dev = start_image.device
        camera_params = np.asarray(camera_params, dtype=np.float64)
        cam_t = torch.tensor(camera_params, dtype=torch.float32, device=dev)
        cam_t[:, :3] *= pos_scale
        all_u8 = None
        memory = torch.zeros(self.num_frames, 3, self.height, self.width, device=dev)       # 'empty_with_traj' memory
        for seg in range(self.num_segments):
            start_idx, end_idx, _ = RP.calculate_segment_indices(seg)
            first = start_image if seg == 0 else ops.u8_hwc_to_f32_chw(all_u8[-1:])[0]    # pil_to_tensor(tensor_to_pil(.)) (:418-419)
            cond = image_latents_fn(first, memory) if image_latents_fn is not None else {}
            gens = self.nav.navigate_curve_path(cam_t, first, num_inference_steps=self.steps, memorized_images=memory[None],
                                                infer_segment=True, segment_id=seg, output_type="latent", **cond, **pipe_kw)
            latents, _n = gens[-1]
            if self.frames_from_latents is None:                                          # the pipeline's own VAE decodes (chunks of 8)
                pipe = self.nav.pipe
                dec = pipe.decode_latents(latents, self.num_frames, 8)[0].permute(1, 0, 2, 3)      # [T,3,H,W] in [-1,1]
            else:
                dec = self.frames_from_latents(latents)
            frames_u8 = ops.f32_chw_to_u8_hwc(dec.float().contiguous())
            if all_u8 is not None:
                frames_u8 = frames_u8[1:]                                               # drop the duplicated first frame (:427-429)
            if save_dir and save_segment_frames:                                        # :432-435, file index continues across segments
                _save_u8_frames(frames_u8, os.path.join(save_dir, f"predictions_{seg}"), seg * (self.num_frames - 1))
            all_u8 = frames_u8 if all_u8 is None else torch.cat([all_u8, frames_u8], dim=0)
            if seg < self.num_segments - 1:
                pers, target_yaws = self.convert_pano_to_pers(all_u8, camera_params, seg)
                if save_dir and save_segment_frames:
                    _save_u8_frames(pers, os.path.join(save_dir, f"perspective_look_at_center_{seg}"))   # :449-453
                temp_cam = camera_params.copy()
                s = max(0, end_idx - len(target_yaws))
                temp_cam[s:end_idx, 4] = target_yaws[: end_idx - s]                        # :456-459
                preds = self.depth_model(pers)
                poses = xyz_euler_to_four_by_four_matrix_batch(torch.tensor(temp_cam, dtype=torch.float32), relative=True)
                outdir = os.path.join(save_dir or "", f"rendered_panorama_vggt_open3d_{seg}")
                panos = RP.predictions_to_target_view(preds, poses.numpy(), conf_thres=50.0, prediction_mode="depth_unproject",
                                                      num_target_view=24, outdir=outdir, cubemap_renderer=_Sized(self.renderer, self.pano_size),
                                                      return_device_tensor=True, save_png=bool(save_dir))
                mem24 = RP.memory_to_pixel_values(panos, self.height, self.width)           # [24,3,H,W]
                memory = torch.cat([start_image[None], mem24], dim=0)                      # [episode frame 1] + 24 reprojected (:277-279)
        self.last_frames_u8 = all_u8
        return ops.u8_hwc_to_f32_chw(all_u8)


def _save_u8_frames(u8_hwc, d, start=0):
    """NNN.png (1-based, offset by `start`) dumps of uint8 [F,H,W,3] frames (unified_loop_consistency.py:104-108,432-453)"""
    from PIL import Image
    os.makedirs(d, exist_ok=True)
    for i, f in enumerate(u8_hwc.cpu().numpy()):
        Image.fromarray(f).save(os.path.join(d, f"{i + start + 1:03}.png"))


class _Sized:
    def __init__(self, cr, pano_size):
        self.cr, self.pano_size = cr, pano_size

    def render_cubemaps_to_panoramas(self, v, c, target, n, outdir):
        return self.cr.render_cubemaps_to_panoramas(v, c, target, n, outdir, width=self.pano_size[1], height=self.pano_size[0])
