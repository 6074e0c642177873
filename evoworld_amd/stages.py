"""Stage providers for the networks that sit either side of the hot path and are out of its scope (rows N1 VAE,
N2 CLIP image encoder, N4 VGGT-1B).  The N-segment loop (`inference.UnifiedLoopConsistencyPipeline`) takes them as three
callables; a deployment plugs the real networks in with `--stages package.module:factory`, where `factory(args)` returns an
object with the same three methods as `SyntheticStages`.

`SyntheticStages` is NOT a model: it is a deterministic, weight-free stand-in with the right shapes and value ranges so that
the entry points (`unified_loop_consistency.py`, `run_single_segment.sh`, `run_unified_pipeline.sh`) run end to end on a box
that has no checkpoints (there is no network in the build environment).  All of it runs on the device.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from .geometry import xyz_euler_to_four_by_four_matrix_batch


class _SyntheticVae:
    """Duck type of the two VAE calls the pipeline makes (pipeline_evoworld.py:307-328 `vae.encode(x).latent_dist.mode()`,
    :358-385 `vae.decode(z, num_frames=k).sample`) over the weight-free colour mixes of `SyntheticStages`."""

    def __init__(self, stages):
        self._s = stages
        self.config = SimpleNamespace(scaling_factor=1.0, block_out_channels=(1, 1, 1, 1))

    def encode(self, x):
        lat = torch.einsum("oc,nchw->nohw", self._s.enc.to(x), F.avg_pool2d(x, 8))
        return SimpleNamespace(latent_dist=SimpleNamespace(mode=lambda: lat))

    def decode(self, z, num_frames=1):
        return SimpleNamespace(sample=self._s.frames_from_latents(z[None]))


class SyntheticStages:
    def __init__(self, cross_attention_dim=1024, camera_params=None, depth_hw=(48, 64), vae_scale=0.18215):
        self.xdim, self.cam, self.depth_hw, self.vae_scale = cross_attention_dim, camera_params, depth_hw, vae_scale
        # fixed 3->4 / 4->3 colour mixes standing for the VAE's channel change
        self.enc = torch.tensor([[0.6, 0.3, 0.1], [-0.2, 0.7, -0.5], [0.5, -0.5, 0.0], [0.3, 0.3, 0.3]])
        self.dec = torch.linalg.pinv(self.enc)
        # the pipeline's own components (`StableVideoDiffusionPipeline(unet, vae=stages.vae, image_encoder=stages.image_encoder)`):
        # with them the pipeline performs the reference's conditioning assembly itself, RNG draws included
        self.vae = _SyntheticVae(self)
        self.image_encoder = self._embed
        self.feature_extractor = None

    def _embed(self, pixel_values):
        """CLIP stand-in on the pipeline's 224x224 normalised input -> .image_embeds [N, xdim]"""
        emb = F.adaptive_avg_pool2d(pixel_values.float(), (16, self.xdim // 16 // 3 + 1)).flatten(1)[:, : self.xdim]
        if emb.shape[1] < self.xdim:
            emb = F.pad(emb, (0, self.xdim - emb.shape[1]))
        return SimpleNamespace(image_embeds=emb)

    # VAE encode of [first frame + 25 memory frames] and CLIP embedding of the first frame
    # (pipeline_evoworld.py:214-263 `_encode_vae_image`, :186-212 `_encode_image`)
    def image_latents_fn(self, first, memory):
        x = torch.cat([first[None], memory], 0)
        lat = torch.einsum("oc,nchw->nohw", self.enc.to(x), F.avg_pool2d(x, 8))
        emb = F.adaptive_avg_pool2d(first[None], (16, self.xdim // 16 // 3 + 1)).flatten(1)[:, : self.xdim]
        if emb.shape[1] < self.xdim:
            emb = F.pad(emb, (0, self.xdim - emb.shape[1]))
        return dict(image_latents=lat[None], image_embeddings=emb[:, None])

    # VAE decode (pipeline_evoworld.py:331-364 `decode_latents`)
    def frames_from_latents(self, latents):
        x = torch.einsum("oc,nchw->nohw", self.dec.to(latents), latents[0].float() / 1.0)
        return torch.tanh(F.interpolate(x, scale_factor=8.0, mode="bilinear", align_corners=False))

    # VGGT-1B (unified_loop_consistency.py:336-368): depth, confidence, poses.  Stand-in: a smooth room-like depth field,
    # confidence falling off with depth, ground-truth relative poses, 90-degree pinhole intrinsics.
    def depth_model(self, pers_u8):
        n, hp, wp, _ = pers_u8.shape
        h, w = self.depth_hw
        dev = pers_u8.device
        v, u = torch.meshgrid(torch.linspace(-1, 1, h, device=dev), torch.linspace(-1, 1, w, device=dev), indexing="ij")
        depth = (3.0 / torch.maximum(u.abs(), v.abs() * 1.5).clamp_min(0.25)).expand(n, h, w)
        img = F.interpolate(pers_u8.permute(0, 3, 1, 2).float() / 255, size=(h, w), mode="bilinear", align_corners=False)
        cam = np.zeros((n, 6)) if self.cam is None else np.asarray(self.cam[:n], dtype=np.float64)
        poses = xyz_euler_to_four_by_four_matrix_batch(torch.tensor(cam, dtype=torch.float32), relative=True).double().numpy()
        f = w / 2.0
        K = np.repeat(np.array([[[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]]], np.float32), n, 0)
        return {"depth": depth[..., None].cpu().numpy(), "depth_conf": (1.0 / depth).cpu().numpy(), "images": img.cpu().numpy(),
                "extrinsic": np.linalg.inv(poses)[:, :3, :4].astype(np.float32), "intrinsic": K}


class HipStages(SyntheticStages):
    """VAE (row N1) and CLIP (row N2) on the HIP implementations (`evoworld_amd.vae`, `evoworld_amd.clip`); only the depth
    network (VGGT-1B, row N4) stays the synthetic stand-in.  Weights: `svd_path/vae`, `svd_path/image_encoder` when those
    diffusers folders exist, otherwise random-init of the full architectures (timing / plumbing runs)."""

    def __init__(self, svd_path=None, device="cuda", seed=0, noise_aug_strength=0.02, vae_config=None, clip_config=None, **kw):
        super().__init__(**kw)
        import os
        from .clip import CLIPVisionModelWithProjection
        from .vae import AutoencoderKLTemporalDecoder
        have = lambda sub: bool(svd_path) and os.path.isdir(os.path.join(svd_path, sub))
        # vae_config / clip_config: overrides of the random-init architectures (tests use reduced widths)
        self.vae = (AutoencoderKLTemporalDecoder.from_pretrained(svd_path, subfolder="vae", device=device) if have("vae")
                    else AutoencoderKLTemporalDecoder.from_random(seed=seed, device=device, **(vae_config or {})))
        self.clip = (CLIPVisionModelWithProjection.from_pretrained(svd_path, subfolder="image_encoder", device=device)
                     if have("image_encoder") else CLIPVisionModelWithProjection.from_random(seed=seed, device=device, **(clip_config or {})))
        self.image_encoder = self.clip
        self.noise_aug_strength = noise_aug_strength
        self.decode_chunk_size = 8

    def image_latents_fn(self, first, memory, image_noise=None):
        """pipeline_evoworld.py:570-623: CLIP embedding of the first frame; VAE mode() of [first | memory] + 0.02 * noise.
        Stand-alone form for timing the stage; the episode / CLI paths hand `vae` and `image_encoder` to the pipeline instead,
        which draws the augmentation noise from the window's generator BEFORE the latents, as the reference does
        (pipeline_evoworld.py:596-600 then :663-673)."""
        from .clip import encode_image_preprocess
        x = torch.cat([first[None], memory], 0)                                         # [-1,1]
        emb = self.clip(encode_image_preprocess(x[:1] / 2 + 0.5)).image_embeds[:, None]
        if image_noise is None:
            image_noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(-1))   # private: the global RNG is not touched
        lat = self.vae.encode(x + self.noise_aug_strength * image_noise.to(x.device)).latent_dist.mode()
        return dict(image_latents=lat[None], image_embeddings=emb)

    def frames_from_latents(self, latents):
        """pipeline_evoworld.py:358-385: decode in chunks of decode_chunk_size after dividing by the scaling factor"""
        z = latents[0].float() / self.vae.config.scaling_factor
        c = self.decode_chunk_size
        return torch.cat([self.vae.decode(z[i:i + c], num_frames=min(c, z.shape[0] - i)).sample for i in range(0, z.shape[0], c)])


def load_stages(spec, args, **kw):
    """spec 'pkg.mod:factory' -> factory(args); 'hip' -> HipStages (VAE + CLIP on the HIP kernels); None -> SyntheticStages."""
    if spec == "hip":
        return HipStages(svd_path=getattr(args, "svd_path", None), cross_attention_dim=kw.get("cross_attention_dim", 1024),
                         camera_params=kw.get("camera_params"), depth_hw=(392, 518))
    if not spec:
        return SyntheticStages(**kw)
    import importlib
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn or "make_stages")(args)
