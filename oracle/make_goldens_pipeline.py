#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden-vector generator for the PIPELINE GLUE (runs ONLY in the build container).

Runs the reference's OWN `StableVideoDiffusionPipeline.__call__` (/root/reference/evoworld/pipeline/pipeline_evoworld.py:456-741,
imported read-only, nothing copied) and the reference's OWN `Navigator.move_forward`
(/root/reference/evoworld/inference/navigator_evoworld.py:173-231) on seeded inputs and commits inputs + captured tensors as
data fixtures.  The pieces of the diffusers package the pipeline class leans on are absent from this image; they are
replaced by the thinnest duck types that let the reference's code run unchanged:

  DiffusionPipeline.register_modules / _execution_device / progress_bar / maybe_free_model_hooks   (plumbing, no arithmetic)
  VideoProcessor.preprocess      tensor branch of diffusers 0.31 VaeImageProcessor.preprocess: (nearest resize if the size
                                 differs,) x*2-1 for inputs in [0,1]
  randn_tensor                   diffusers 0.31 utils/torch_utils.py: a CPU generator draws on the host, tensor moved after
  unet                           oracle/unet_ref.py (tiny config, fp32, UN-rounded random weights -- SURVEY §8d's protocol)
  scheduler                      evoworld_amd.scheduler.EulerDiscreteScheduler (host class; pinned by its own KATs)
  vae / image_encoder            oracle/standins.py (closed-form, weight-free)
  feature_extractor              the installed transformers.CLIPImageProcessor (the real third-party component)

What is pinned by this: everything between the call arguments and the U-Net / scheduler calls -- image/memory concat and
rescale (:570-579), CLIP preprocessing (:264-285), aug-noise draw and VAE encode (:594-612), CFG duplication with zeroed
negatives (:297-303, :320-326), mask_mem zeroing (:626-628), Plücker duplicated on both rows (:632), conditioning concat
(:639-640), added_time_ids with fps-1 (:589, :643-652), latents draw + init sigma (:660-671), the guidance ramp (:675-680),
per-step scale_model_input + concat (:691-697), CFG combine (:709-711), scheduler step (:714) -- and the RNG draw ORDER.

Usage:  python oracle/make_goldens_pipeline.py   (from the repo root)  ->  tests/golden/pipeline_glue.npz, navigator_glue.npz
"""
import contextlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

UNET_SEED = 3
T, H, W = 4, 128, 256
STEPS = 25


def case_inputs(seed, B=1):
    """Seeded call arguments of one glue case (the test rebuilds nothing: these are stored in the fixture)."""
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    memory = torch.rand(B, T, 3, H, W, generator=g) * 2 - 1
    plucker = torch.randn(B, T, 6, H // 8, W // 8, generator=g)
    return image.half().float(), memory.half().float(), plucker     # pixels on the fp16 grid: stored as fp16 in the fixture


def _patch_reference():
    from oracle.make_goldens import _import_reference
    cwd = os.getcwd()
    _import_reference()
    os.chdir(cwd)
    import evoworld.pipeline.pipeline_evoworld as P

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        device = torch.device(device or "cpu")
        rand_device = device
        if generator is not None and not isinstance(generator, list):
            if generator.device.type != device.type and generator.device.type == "cpu":
                rand_device = torch.device("cpu")
        return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)

    class VideoProcessor:
        def __init__(self, do_resize=True, vae_scale_factor=8):
            self.vae_scale_factor = vae_scale_factor

        def preprocess(self, image, height=None, width=None):
            assert isinstance(image, torch.Tensor) and image.ndim == 4
            if height is not None and tuple(image.shape[-2:]) != (height, width):
                image = torch.nn.functional.interpolate(image, size=(height, width))
            if image.min() < 0:          # diffusers: inputs already in [-1,1] are passed through with a warning
                return image
            return 2.0 * image - 1.0

        def postprocess_video(self, video, output_type="np"):
            raise AssertionError("the glue goldens are captured with output_type='latent'")

    P.randn_tensor = randn_tensor
    P.VideoProcessor = VideoProcessor
    C = P.StableVideoDiffusionPipeline

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        yield SimpleNamespace(update=lambda *a: None)

    C.register_modules = register_modules
    C._execution_device = property(lambda self: torch.device("cpu"))
    C.progress_bar = progress_bar
    C.maybe_free_model_hooks = lambda self: None
    return P


class UNetShim:
    """call surface the reference loop uses (:700-706) over the fp32 oracle; records what the loop hands to the model."""

    def __init__(self, ref, cfg):
        self.ref = ref
        self.config = SimpleNamespace(in_channels=cfg["in_channels"], addition_time_embed_dim=cfg["addition_time_embed_dim"],
                                      sample_size=96, num_frames=cfg["num_frames"])
        self.add_embedding = ref.add_embedding
        self.calls = []

    def __call__(self, sample, timestep, encoder_hidden_states=None, added_time_ids=None, return_dict=False):
        out = self.ref(sample, timestep, encoder_hidden_states, added_time_ids)
        self.calls.append(dict(sample=sample.clone(), t=float(timestep), ehs=encoder_hidden_states.clone(),
                               ids=added_time_ids.clone(), out=out.clone()))
        return (out,)


def build_oracle_unet():
    from evoworld_amd.unet import DEFAULT_CONFIG, random_state_dict
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef, tiny_config
    cfg = tiny_config()
    sd = random_state_dict({**DEFAULT_CONFIG, **cfg}, UNET_SEED)          # fp32, NOT rounded to fp16
    ref = UNetSpatioTemporalConditionModelRef(**cfg).eval()
    ref.load_state_dict(sd)
    return cfg, ref


def pipeline_goldens(P):
    from transformers import CLIPImageProcessor
    from evoworld_amd.scheduler import EulerDiscreteScheduler
    from oracle.standins import StandInCLIP, StandInVAE
    cfg, ref = build_oracle_unet()
    fe = CLIPImageProcessor()          # default image_mean / image_std = the OpenAI CLIP statistics
    gold = {"T": np.int64(T), "H": np.int64(H), "W": np.int64(W), "steps": np.int64(STEPS), "unet_seed": np.int64(UNET_SEED),
            "image_mean": np.asarray(fe.image_mean, np.float32), "image_std": np.asarray(fe.image_std, np.float32)}
    cases = [("mem", 21, False, dict(), 1),
             ("mask", 22, True, dict(min_guidance_scale=1.5, max_guidance_scale=4.0, fps=9, motion_bucket_id=63,
                                     noise_aug_strength=0.05), 1),
             ("b2", 23, False, dict(), 2)]          # batch of two clips (:573-578): one noise draw / one latents draw for the batch
    for tag, seed, mask_mem, extra, B in cases:
        image, memory, plucker = case_inputs(seed, B)
        unet = UNetShim(ref, cfg)
        pipe = P.StableVideoDiffusionPipeline(vae=StandInVAE(), image_encoder=StandInCLIP(cfg["cross_attention_dim"]), unet=unet,
                                              scheduler=EulerDiscreteScheduler(), feature_extractor=fe)
        trace = []
        gen = torch.manual_seed(-1)      # what navigator_evoworld.py:198 hands over: the re-seeded DEFAULT cpu generator
        out = pipe(image, height=H, width=W, num_frames=T, num_inference_steps=STEPS, generator=gen, decode_chunk_size=8,
                   output_type="latent", plucker_embedding=plucker, memorized_pixel_values=memory, mask_mem=mask_mem,
                   callback_on_step_end=lambda p, i, t, kw: trace.append(kw["latents"].clone()) or {}, **extra)
        c0 = unet.calls[0]
        assert len(unet.calls) == STEPS and len(trace) == STEPS
        if B > 1:                        # batch case: inputs, step-0 model input ([2B,...]: B uncond rows, then B cond rows) and the result
            gold.update({f"{tag}_image": image.half().numpy(), f"{tag}_memory": memory.half().numpy(), f"{tag}_plucker": plucker.numpy(),
                         f"{tag}_step0_latent_model_input": c0["sample"].numpy(), f"{tag}_final_latents": out.frames.numpy(),
                         f"{tag}_rng_state_after": gen.get_state().numpy()[:64].copy()})
            print(tag, "final latents", tuple(out.frames.shape))
            continue
        gold.update({
            f"{tag}_image": image.half().numpy(), f"{tag}_memory": memory.half().numpy(), f"{tag}_plucker": plucker.numpy(),
            f"{tag}_mask_mem": np.bool_(mask_mem),
            f"{tag}_kwargs": np.array([extra.get("min_guidance_scale", 1.0), extra.get("max_guidance_scale", 3.0), extra.get("fps", 7),
                                       extra.get("motion_bucket_id", 127), extra.get("noise_aug_strength", 0.02)], np.float64),
            f"{tag}_step0_latent_model_input": c0["sample"].numpy(),            # [2,T,18,h,w]  (:691-697)
            f"{tag}_step0_timestep": np.float64(c0["t"]),
            f"{tag}_image_embeddings": c0["ehs"].numpy(),                        # [2,1,X]: zeros | CLIP(image)
            f"{tag}_added_time_ids": c0["ids"].numpy(),                          # [2,3]
            f"{tag}_guidance_scale": pipe.guidance_scale.numpy(),                # [1,T,1,1,1]
            f"{tag}_step1_latent_model_input": unet.calls[1]["sample"].numpy(),  # after one CFG combine + Euler step
            f"{tag}_latents_after_step": torch.stack([trace[i] for i in (0, 1, 2, STEPS - 1)]).numpy(),
            f"{tag}_final_latents": out.frames.numpy(),
            f"{tag}_rng_state_after": gen.get_state().numpy()[:64].copy(),       # the generator was advanced by exactly two draws
        })
        print(tag, "final latents", tuple(out.frames.shape), "|x|", float(out.frames.norm()))
    np.savez_compressed(os.path.join(OUT, "pipeline_glue.npz"), **gold)
    print("pipeline_glue.npz", os.path.getsize(os.path.join(OUT, "pipeline_glue.npz")))


def navigator_goldens():
    """Navigator.move_forward (navigator_evoworld.py:173-231) with a recording pipe: what the window hands to the pipeline
    (Plücker from the extended segment, generator state, mask_mem, memory clone) and the bookkeeping it leaves behind.
    The method hard-codes `.to('cuda:0')`; in the build container those calls are redirected to the CPU."""
    import evoworld.inference.navigator_evoworld as NV
    from PIL import Image
    from utils.plucker_embedding import equirectangular_to_ray
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        return real_to(self, *a, **k)

    gold = {}
    torch.Tensor.to = to_cpu
    try:
        for tag, n_seg, use_memory in (("full", 25, True), ("short", 11, False)):
            nav = NV.Navigator.__new__(NV.Navigator)
            nav.logger = SimpleNamespace(info=lambda *a, **k: None)
            nav.position_scale, nav.generations = 0.1, []
            nav.previous_images = torch.zeros(0, 3, 64, 128)
            nav.previous_trajectoies = torch.tensor([])
            nav.rays = torch.tensor(equirectangular_to_ray(target_H=8, target_W=16)).to(torch.float32)
            nav.transform = lambda im: torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float() / 255.0 * 2 - 1
            nav.model_width, nav.model_height, nav.num_frames, nav.fps = 128, 64, 25, 7
            g = torch.Generator().manual_seed(31)
            nav.memorized_images = torch.rand(1, 25, 3, 64, 128, generator=g) * 2 - 1
            rec = {}

            def pipe(image, **kw):
                rec.update(kw, image=image, rng=kw["generator"].get_state().clone(),
                           default_rng_is_generator=kw["generator"] is torch.default_generator)
                frames = [Image.fromarray(np.full((64, 128, 3), 10 * i, np.uint8)) for i in range(25)]
                return SimpleNamespace(frames=[frames])
            nav.pipe = pipe
            # a straight segment with constant rotation (what split_path_into_segments produces)
            start = torch.tensor([0.3, 0.0, -0.2, 0.0, 30.0, 0.0])
            step = torch.tensor([0.04, 0.0, 0.03, 0.0, 0.0, 0.0])
            segment = [start + step * i for i in range(n_seg)]
            image = torch.rand(3, 64, 128, generator=g) * 2 - 1
            frames = nav.move_forward(image=image, segment=segment, num_model_frames=25, width=128, height=64,
                                      num_inference_steps=7, noise_aug_strength=0.03, use_memory=use_memory)
            ref_state = torch.manual_seed(-1).get_state()
            gold.update({
                f"{tag}_segment": torch.stack(segment).numpy(), f"{tag}_use_memory": np.bool_(use_memory),
                f"{tag}_plucker_embedding": rec["plucker_embedding"].numpy(),                 # [1,25,6,8,16]
                f"{tag}_mask_mem": np.bool_(rec["mask_mem"]),
                f"{tag}_generator_is_default_reseeded": np.bool_(rec["default_rng_is_generator"] and torch.equal(rec["rng"], ref_state)),
                f"{tag}_pipe_scalars": np.array([rec["num_frames"], rec["width"], rec["height"], rec["decode_chunk_size"],
                                                 rec["motion_bucket_id"], rec["fps"], rec["num_inference_steps"]], np.float64),
                f"{tag}_noise_aug_strength": np.float64(rec["noise_aug_strength"]),
                f"{tag}_image_passed": rec["image"].numpy(), f"{tag}_image": image.numpy(),
                f"{tag}_memory_passed_equal": np.bool_(torch.equal(rec["memorized_pixel_values"], nav.memorized_images)),
                f"{tag}_n_frames_returned": np.int64(len(frames)), f"{tag}_n_generation": np.int64(len(nav.generations[-1])),
                f"{tag}_current_pose": nav.current_pose.numpy(),
                f"{tag}_previous_trajectories": nav.previous_trajectoies.numpy(),
                f"{tag}_previous_images_shape": np.array(nav.previous_images.shape, np.int64),
                f"{tag}_first_returned_pixel": np.int64(np.asarray(frames[0])[0, 0, 0]),
                f"{tag}_last_returned_pixel": np.int64(np.asarray(frames[-1])[0, 0, 0]),
            })
            print(tag, "plucker", tuple(rec["plucker_embedding"].shape), "frames", len(frames), "pose", nav.current_pose.tolist())
    finally:
        torch.Tensor.to = real_to
    np.savez_compressed(os.path.join(OUT, "navigator_glue.npz"), **gold)
    print("navigator_glue.npz", os.path.getsize(os.path.join(OUT, "navigator_glue.npz")))


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    P = _patch_reference()
    pipeline_goldens(P)
    navigator_goldens()


if __name__ == "__main__":
    main()
