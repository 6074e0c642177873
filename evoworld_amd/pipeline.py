"""StableVideoDiffusionPipeline with EvoWorld conditioning -- the reference's call surface
(evoworld/pipeline/pipeline_evoworld.py:197-741) over the HIP U-Net and the fused denoise-step kernel.

    pipe = StableVideoDiffusionPipeline(unet=unet, scheduler=EulerDiscreteScheduler(), vae=vae, image_encoder=clip)
    frames = pipe(image, height=576, width=1024, num_frames=25, decode_chunk_size=8, motion_bucket_id=127, fps=7,
                  noise_aug_strength=0.02, plucker_embedding=..., memorized_pixel_values=..., mask_mem=False).frames[0]

Same 23 keyword parameters, same conditioning assembly (first-frame latent repeated, memory latents, Plücker
duplicated -- NOT zeroed -- on the unconditional CFG row, :635-643), same RNG draw order (aug noise, then latents).
Hot loop (:689-725) on the GPU: per step ONE U-Net forward on the persistent fp16 channels-last input buffer
[2*T*h*w, 64] plus ONE fused kernel (CFG combine + Euler v-prediction step + scale_model_input + rewrite of the 4 noisy
channels of the next input); the 14 conditioning channels are written once per clip.

Scope: the temporal VAE (N1) and the CLIP image encoder (N2) are third-party models outside the hot
path; they are duck-typed components here (`vae.encode(x).latent_dist.mode()`, `vae.decode(z, num_frames=k).sample`,
`image_encoder(x).image_embeds`).  When they are not supplied, pass `image_latents=` ([1, 1+T, 4, h, w], unscaled VAE
mode, before CFG duplication) and `image_embeddings=` ([1,1,1024]) and use output_type="latent".  `image_noise=` injects
the augmentation noise draw (SURVEY.md §7 RNG parity) -- all three extra kwargs default to None = reference behaviour.
"""
from types import SimpleNamespace

import torch

from . import ops
from .scheduler import EulerDiscreteScheduler
from .unet import CPAD_IN


_RANDN_CACHE = {}
_RANDN_CACHE_BYTES = 512 << 20          # device bytes the memoised draws may hold (two full-size windows' worth)


def _randn_like_reference(shape, generator, device, dtype=torch.float32):
    """diffusers `randn_tensor(shape, generator, device)`: a CPU generator draws on the host and the tensor is moved.  The
    reference re-seeds ONE CPU generator identically for every window (`torch.manual_seed(-1)`, navigator_evoworld.py:198), so
    the 46 M-value augmentation noise and the latents are the same tensors window after window: draws are memoised on
    (generator state, shape, dtype) -- a hit restores the generator to the state the real draw would have left it in and returns
    the device-resident tensor (0.35 s of host RNG + 184 MB of PCIe per window at 576x1024x25 otherwise).  Bit-identical to
    drawing again by construction; a clone is returned (a 184 MB device copy is ~50 us)."""
    if not isinstance(generator, torch.Generator):
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)
    if generator.device.type != "cpu":
        return torch.randn(shape, generator=generator, device=generator.device, dtype=dtype).to(device)
    key = (bytes(generator.get_state().numpy()), tuple(shape), dtype, str(device))
    hit = _RANDN_CACHE.get(key)
    if hit is not None:
        generator.set_state(hit[1])
        return hit[0].clone()              # callers may modify what they get; the memoised tensor stays pristine
    t = torch.randn(shape, generator=generator, device="cpu", dtype=dtype).to(device)
    _RANDN_CACHE[key] = (t, generator.get_state().clone())
    while len(_RANDN_CACHE) > 1 and sum(v[0].numel() * v[0].element_size() for v in _RANDN_CACHE.values()) > _RANDN_CACHE_BYTES:
        _RANDN_CACHE.pop(next(iter(_RANDN_CACHE)))            # oldest first
    return t.clone()


def clear_randn_cache():
    """Drop the memoised reference-noise tensors (device memory: 184 MB for the augmentation noise of a 576x1024x25 window)."""
    _RANDN_CACHE.clear()


def _append_dims(x, target_dims):
    """pipeline_evoworld.py:128-133"""
    d = target_dims - x.ndim
    if d < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * d]


class StableVideoDiffusionPipelineOutput(SimpleNamespace):
    pass


class StableVideoDiffusionPipeline:
    def __init__(self, unet, scheduler=None, vae=None, image_encoder=None, feature_extractor=None):
        self.unet, self.vae, self.image_encoder, self.feature_extractor = unet, vae, image_encoder, feature_extractor
        self.scheduler = scheduler or EulerDiscreteScheduler()
        self.vae_scale_factor = 8 if vae is None else 2 ** (len(vae.config.block_out_channels) - 1)
        self._device = unet.device or torch.device("cuda")
        self._progress = {}
        self.cfg_group = None       # evoworld_amd.distributed.CfgGroup: split the two CFG rows of every step over a rank pair

    def set_components(self, vae=None, image_encoder=None, feature_extractor=None):
        """Plug in (or replace) the conditioning / decoding networks after construction: a stage provider of
        `evoworld_amd.stages` hands its `vae` / `image_encoder` to the pipeline so that the pipeline itself runs the
        reference's conditioning assembly (pipeline_evoworld.py:570-623), RNG draws included."""
        if vae is not None:
            self.vae = vae
            self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        if image_encoder is not None:
            self.image_encoder = image_encoder
        if feature_extractor is not None:
            self.feature_extractor = feature_extractor
        return self

    @classmethod
    def from_pretrained(cls, path=None, unet=None, **kw):
        """The reference builds the pipeline from a diffusers folder and injects its own unet
        (unified_loop_consistency.py:193-197: StableVideoDiffusionPipeline.from_pretrained(svd_path, unet=unet, ...)).
        The folder's `vae/` (AutoencoderKLTemporalDecoder), `image_encoder/` (CLIPVisionModelWithProjection) and
        `feature_extractor/preprocessor_config.json` (image_mean / image_std) are loaded onto the HIP implementations when
        present; components passed as keywords win."""
        import json
        import os
        if unet is None:
            raise ValueError("pass unet= (evoworld_amd.unet.UNetSpatioTemporalConditionModel)")
        dev = unet.device or "cuda"
        vae, enc, fe = kw.get("vae"), kw.get("image_encoder"), kw.get("feature_extractor")
        if path and os.path.isdir(path):
            if vae is None and os.path.isdir(os.path.join(path, "vae")):
                from .vae import AutoencoderKLTemporalDecoder
                vae = AutoencoderKLTemporalDecoder.from_pretrained(path, subfolder="vae", device=dev)
            if enc is None and os.path.isdir(os.path.join(path, "image_encoder")):
                from .clip import CLIPVisionModelWithProjection
                enc = CLIPVisionModelWithProjection.from_pretrained(path, subfolder="image_encoder", device=dev)
            pj = os.path.join(path, "feature_extractor", "preprocessor_config.json")
            if fe is None and os.path.exists(pj):
                raw = json.load(open(pj))
                fe = SimpleNamespace(image_mean=raw.get("image_mean"), image_std=raw.get("image_std"))
        return cls(unet=unet, scheduler=kw.get("scheduler"), vae=vae, image_encoder=enc, feature_extractor=fe)

    def _encode_image(self, image01):
        """pipeline_evoworld.py:255-305: [0,1] image -> x*2-1 -> antialiased resize to 224x224 -> (x+1)/2 -> CLIP mean / std
        normalisation (feature_extractor with do_resize / do_center_crop / do_rescale off) -> image_encoder -> [N,1,X]."""
        from .clip import CLIP_MEAN, CLIP_STD, encode_image_preprocess
        fe = self.feature_extractor
        mean = tuple(getattr(fe, "image_mean", None) or CLIP_MEAN)
        std = tuple(getattr(fe, "image_std", None) or CLIP_STD)
        pixel_values = encode_image_preprocess(image01.to(self._device), mean, std)
        return self.image_encoder(pixel_values).image_embeds.unsqueeze(1)

    def to(self, device=None, dtype=None):
        return self

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    @property
    def _execution_device(self):
        return self._device

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        g = self._guidance_scale
        return g > 1 if isinstance(g, (int, float)) else bool(g.max() > 1)

    @property
    def num_timesteps(self):
        return self._num_timesteps

    def check_inputs(self, image, height, width):
        import PIL.Image
        if not isinstance(image, (torch.Tensor, PIL.Image.Image, list)):                     # :387-396
            raise ValueError("`image` has to be of type `torch.Tensor` or `PIL.Image.Image` or `List[PIL.Image.Image]` but is"
                             f" {type(image)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def _image_to_tensor(self, image, height, width):
        """PIL image / list of PIL images -> float [B,3,H,W] in [-1,1] (what the reference callers pass as a tensor: ToTensor +
        rescale, navigator_evoworld.py:436-437); resized to (height, width) with PIL's bilinear filter when the size differs
        (VideoProcessor.preprocess semantics).  Tensors pass through."""
        if isinstance(image, torch.Tensor):
            return image
        import numpy as np
        import PIL.Image
        imgs = image if isinstance(image, list) else [image]
        out = []
        for im in imgs:
            if not isinstance(im, PIL.Image.Image):
                raise ValueError(f"`image` list entries have to be PIL images but found {type(im)}")
            im = im.convert("RGB")
            if im.size != (width, height):
                im = im.resize((width, height), PIL.Image.BILINEAR)
            out.append(torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float() / 255.0 * 2 - 1)
        return torch.stack(out).to(self._device)

    def _get_add_time_ids(self, fps, motion_bucket_id, noise_aug_strength, dtype, batch_size, num_videos_per_prompt, cfg):
        add_time_ids = [fps, motion_bucket_id, noise_aug_strength]
        passed = self.unet.config.addition_time_embed_dim * len(add_time_ids)
        expected = self.unet.add_embedding.linear_1.in_features
        if expected != passed:
            raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} "
                             "was created. The model has an incorrect config.")
        ids = torch.tensor([add_time_ids], dtype=dtype).repeat(batch_size * num_videos_per_prompt, 1)
        return torch.cat([ids, ids]) if cfg else ids

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_frames, 4, height // self.vae_scale_factor, width // self.vae_scale_factor)  # :413-420
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                             f"batch size of {batch_size}.")
        if latents is None:
            latents = _randn_like_reference(shape, generator, device, dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------ the hot loop
    def denoise(self, latents, conditional_latents, image_embeddings, added_time_ids, guidance, num_inference_steps,
                callback=None, cfg_group=None):
        """latents fp32 [1,T,4,h,w] (already scaled by init_noise_sigma); conditional_latents fp32 [2,T,14,h,w];
        image_embeddings [2,1,X]; added_time_ids [2,3]; guidance fp32 [T].  Returns final latents fp32 [1,T,4,h,w]."""
        dev = self._device
        _, T, _, h, w = latents.shape
        ncond = conditional_latents.shape[2]
        if 4 + ncond != self.unet.config.in_channels:
            raise ValueError(f"4 + {ncond} conditioning channels != unet in_channels {self.unet.config.in_channels}")
        sig = self.scheduler.sigmas
        ts = self.scheduler.timesteps
        lat = latents[0].to(device=dev, dtype=torch.float32).contiguous().clone()
        x_in = torch.zeros(2 * T * h * w, CPAD_IN, dtype=torch.float16, device=dev)
        # the model-input row as the U-Net wants it: plain fp16 channels, or (default) the split operand [x_hi | x_lo | x_hi 2^-10] of its conv_in
        split = getattr(self.unet, "in_split", None)
        ops.nchw_f32_to_nhwc_f16(conditional_latents.to(device=dev, dtype=torch.float32).reshape(2 * T, ncond, h, w).contiguous(),
                                 x_in, CPAD_IN, c_off=4, split=split)
        s0 = float(sig[0])
        for half in range(2):  # latents duplicated on both CFG rows, scale_model_input for step 0 (:691-692)
            ops.nchw_f32_to_nhwc_f16(lat, x_in[half * T * h * w:], CPAD_IN, c_off=0, scale=1.0 / (s0 * s0 + 1.0) ** 0.5, split=split)
        guidance = guidance.to(device=dev, dtype=torch.float32).contiguous()
        ehs = image_embeddings.to(device=dev, dtype=torch.float16)        # once per clip: the forward takes fp16 embeddings / fp32 ids as they are
        ids = added_time_ids.to(device=dev, dtype=torch.float32)
        grp = cfg_group if cfg_group is not None else self.cfg_group
        if grp is None:
            for i in range(num_inference_steps):
                eps = self.unet.forward_nhwc(x_in, ts[i], ehs, ids, 2, T, h, w)
                ops.euler_cfg_step(eps, eps.shape[-1], lat, guidance, float(sig[i]), float(sig[i + 1]), x_in, CPAD_IN, T, h, w, split=split)
                if callback is not None:
                    callback(i, ts[i], lat)
        else:
            # CFG-pair split (north_star "denoising-step batch"; the CFG batch of pipeline_evoworld.py:691-711): this rank runs
            # the U-Net on its row(s) only (B = 1), ONE all_gather of the eps rows per step, combine + Euler step replicated --
            # every member keeps the full latents and both rows of the next model input, so eps is all that crosses xGMI.
            rows = T * h * w
            eps_all = None
            for i in range(num_inference_steps):
                for r in grp.rows():
                    eps_r = self.unet.forward_nhwc(x_in[r * rows:(r + 1) * rows], ts[i], ehs[r:r + 1], ids[r:r + 1], 1, T, h, w)
                    if eps_all is None:
                        eps_all = torch.zeros(2, rows, eps_r.shape[-1], dtype=eps_r.dtype, device=dev)
                    eps_all[r].copy_(eps_r)
                grp.all_gather_rows(eps_all, eps_r)       # eps_r: the forward's own output buffer (not a view of eps_all)
                ops.euler_cfg_step(eps_all, eps_all.shape[-1], lat, guidance, float(sig[i]), float(sig[i + 1]), x_in, CPAD_IN, T, h, w, split=split)
                if callback is not None:
                    callback(i, ts[i], lat)
        ops.streamk_check()      # synchronises; raises if any stream-K hand-over of the loop timed out (wrong tile)
        return lat[None]

    @torch.no_grad()
    def __call__(self, image, height=576, width=1024, num_frames=None, num_inference_steps=25, sigmas=None,
                 min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=127, noise_aug_strength=0.02,
                 decode_chunk_size=None, num_videos_per_prompt=1, generator=None, latents=None, output_type="pil",
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=["latents"], return_dict=True,
                 plucker_embedding=None, memorized_plucker_embedding=None, memorized_pixel_values=None, mask_mem=False,
                 image_latents=None, image_embeddings=None, image_noise=None):
        dev = self._device
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        num_frames = num_frames if num_frames is not None else self.unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        if num_videos_per_prompt != 1:
            # the reference repeats image latents / embeddings num_videos_per_prompt times (:292-294, :316) but NOT plucker_embedding,
            # so its conditioning concat (:639-640) fails with a batch mismatch for any value but 1
            raise ValueError("num_videos_per_prompt must be 1: plucker_embedding / memorized_pixel_values are per clip "
                             "(the reference's own conditioning concat, pipeline_evoworld.py:639-640, needs matching batch sizes)")
        self.check_inputs(image, height, width)
        image = self._image_to_tensor(image, height, width)
        if plucker_embedding is None:
            raise ValueError("plucker_embedding [B,T,6,h,w] is required (evoworld conditioning)")
        batch_size = image.shape[0]
        if batch_size != 1:
            return self._call_batched(image, dict(
                height=height, width=width, num_frames=num_frames, num_inference_steps=num_inference_steps, sigmas=sigmas,
                min_guidance_scale=min_guidance_scale, max_guidance_scale=max_guidance_scale, fps=fps, motion_bucket_id=motion_bucket_id,
                noise_aug_strength=noise_aug_strength, decode_chunk_size=decode_chunk_size, output_type=output_type,
                callback_on_step_end=callback_on_step_end, callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs,
                mask_mem=mask_mem), generator, latents, plucker_embedding, memorized_plucker_embedding, memorized_pixel_values,
                image_latents, image_embeddings, image_noise, return_dict)
        self._guidance_scale = max_guidance_scale
        cfg = self.do_classifier_free_guidance

        # --- 3/4. image embedding + VAE latents of [first frame | memory frames]  (:570-623)
        if image_latents is None or image_embeddings is None:
            if self.vae is None or self.image_encoder is None:
                raise ValueError("no vae / image_encoder component: pass image_latents= and image_embeddings= "
                                 "(VAE and CLIP are 'next' rows, SURVEY.md §8f N1/N2)")
            img = torch.cat([image.unsqueeze(1), memorized_pixel_values], dim=1).to(dev)   # :570
            img = img / 2.0 + 0.5                                                           # :579
            image_embeddings = self._encode_image(img[:, 0])
            flat = img.flatten(0, 1) * 2.0 - 1.0                                            # VideoProcessor.preprocess
            noise = image_noise if image_noise is not None else _randn_like_reference(flat.shape, generator, dev, flat.dtype)
            flat = flat + noise_aug_strength * noise.to(dev)                                 # :599-600
            image_latents = self.vae.encode(flat).latent_dist.mode().reshape(1, -1, 4, height // 8, width // 8)
        elif image_noise is None and isinstance(generator, torch.Generator):
            # injected conditioning: keep the reference's RNG order anyway -- the [n_cond,3,height,width] augmentation-noise draw
            # (shape after VideoProcessor.preprocess, :594-599) comes first, so the latents below are the generator's SECOND draw
            _randn_like_reference((image_latents.shape[1], 3, height, width), generator, dev)
        image_latents = image_latents.to(device=dev, dtype=torch.float32).clone()
        image_embeddings = image_embeddings.to(device=dev, dtype=torch.float32)
        if image_latents.shape[1] != num_frames + 1:
            raise ValueError(f"memory frames ({image_latents.shape[1] - 1}) must equal num_frames ({num_frames})")  # :643
        if cfg:                                                                              # :297-303, :320-326
            image_embeddings = torch.cat([torch.zeros_like(image_embeddings), image_embeddings])
            image_latents = torch.cat([torch.zeros_like(image_latents), image_latents])
        else:
            image_embeddings = torch.cat([image_embeddings, image_embeddings])
            image_latents = torch.cat([image_latents, image_latents])
        if mask_mem:
            image_latents[:, 1:] = 0                                                         # :629-631
        plucker = plucker_embedding.to(device=dev, dtype=torch.float32)
        plucker = torch.cat([plucker, plucker], dim=0)                                       # :635 (dup, not zeroed)
        first = image_latents[:, 0:1].repeat(1, num_frames, 1, 1, 1)
        conditional_latents = torch.cat([first, image_latents[:, 1:], plucker], dim=2)       # :642-643

        # --- 5-8. ids, timesteps, latents, guidance  (:646-682)
        added_time_ids = self._get_add_time_ids(fps - 1, motion_bucket_id, noise_aug_strength, torch.float32, 1, 1, True)
        if sigmas is not None:                                                               # retrieve_timesteps (:138-194)
            self.scheduler.set_timesteps(sigmas=sigmas, device="cpu")
            num_inference_steps = self.scheduler.num_inference_steps
        else:
            self.scheduler.set_timesteps(num_inference_steps, device="cpu")
        latents = self.prepare_latents(1, num_frames, self.unet.config.in_channels, height, width, torch.float32, dev,
                                       generator, latents)
        if cfg:
            guidance = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames)
        else:
            guidance = torch.ones(num_frames)      # both rows carry the conditional inputs -> e == cond prediction
        self._guidance_scale = _append_dims(guidance.unsqueeze(0), latents.ndim)
        self._num_timesteps = num_inference_steps

        cb = None
        if callback_on_step_end is not None:
            def cb(i, t, lat):
                out = callback_on_step_end(self, i, t, {"latents": lat[None]})
                if out and "latents" in out and out["latents"] is not lat[None]:
                    raise NotImplementedError("callbacks that replace latents are not supported on the fused path")
        latents = self.denoise(latents, conditional_latents, image_embeddings, added_time_ids, guidance, num_inference_steps, cb)

        if output_type != "latent":
            if self.vae is None:
                raise ValueError("output_type != 'latent' needs a vae component")
            frames = self.decode_latents(latents, num_frames, decode_chunk_size)
            frames = self._postprocess(frames, output_type)
        else:
            frames = latents
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)

    def _call_batched(self, image, kw, generator, latents, plucker, mem_plucker, memory, image_latents, image_embeddings, image_noise,
                      return_dict):
        """Batch B > 1 (pipeline_evoworld.py:573-578: batch_size = image.shape[0]): the clips of a batch are independent, so they
        run one after the other through the single-clip path -- each with its own [uncond, cond] U-Net batch -- after the RANDOM
        DRAWS have been made for the whole batch exactly as the reference makes them: one [(B*(1+T)),3,H,W] augmentation-noise
        tensor (:596-599), then one [B,T,4,h,w] latents tensor (:660-671), sliced per clip."""
        dev = self._device
        B, T = image.shape[0], kw["num_frames"]
        H, W = kw["height"], kw["width"]
        if isinstance(generator, list):
            if len(generator) != B:
                raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                                 f"batch size of {B}.")
            raise ValueError("a list of generators with batch > 1 is not supported (the reference's randn_tensor call for the "
                             "augmentation noise indexes the list by (batch x frame) and fails too); pass one generator")
        for name, t in (("plucker_embedding", plucker), ("memorized_pixel_values", memory), ("latents", latents),
                        ("image_latents", image_latents), ("image_embeddings", image_embeddings)):
            if t is not None and t.shape[0] != B:
                raise ValueError(f"{name} has batch {t.shape[0]} but image has batch {B}")
        n_cond = (1 + memory.shape[1]) if memory is not None else (image_latents.shape[1] if image_latents is not None else 1 + T)
        if image_noise is None and (image_latents is None or image_embeddings is None or isinstance(generator, torch.Generator)):
            image_noise = _randn_like_reference((B * n_cond, 3, H, W), generator, dev)                     # draw #1, whole batch
        if latents is None:
            latents = _randn_like_reference((B, T, 4, H // self.vae_scale_factor, W // self.vae_scale_factor), generator, dev)   # draw #2
        outs = []
        for b in range(B):
            sl = slice(b, b + 1)
            outs.append(self(image[sl], **kw, generator=None, latents=latents[sl], return_dict=False,
                             plucker_embedding=plucker[sl], memorized_plucker_embedding=None if mem_plucker is None else mem_plucker[sl],
                             memorized_pixel_values=None if memory is None else memory[sl],
                             image_latents=None if image_latents is None else image_latents[sl],
                             image_embeddings=None if image_embeddings is None else image_embeddings[sl],
                             image_noise=None if image_noise is None else image_noise[b * n_cond:(b + 1) * n_cond]))
        if isinstance(outs[0], torch.Tensor):
            frames = torch.cat(outs, dim=0)
        elif isinstance(outs[0], list):
            frames = [clip for o in outs for clip in o]
        else:
            import numpy as np
            frames = np.concatenate(outs, axis=0)
        return StableVideoDiffusionPipelineOutput(frames=frames) if return_dict else frames

    def decode_latents(self, latents, num_frames, decode_chunk_size=14):
        latents = latents.flatten(0, 1) / self.vae.config.scaling_factor                     # :360-362
        frames = [self.vae.decode(latents[i:i + decode_chunk_size], num_frames=latents[i:i + decode_chunk_size].shape[0]).sample
                  for i in range(0, latents.shape[0], decode_chunk_size)]
        frames = torch.cat(frames, dim=0)
        return frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4).float()

    @staticmethod
    def _postprocess(frames, output_type):
        vid = (frames / 2 + 0.5).clamp(0, 1)                                                 # [B,C,T,H,W]
        if output_type == "pt":
            return vid.permute(0, 2, 1, 3, 4)
        arr = (vid.permute(0, 2, 3, 4, 1).cpu().numpy() * 255).round().astype("uint8")
        if output_type == "np":
            return arr
        from PIL import Image
        return [[Image.fromarray(f) for f in clip] for clip in arr]
