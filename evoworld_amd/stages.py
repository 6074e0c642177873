"""Stage providers for the networks that sit either side of the hot path and are out of its scope (rows N1 VAE,
N2 CLIP image encoder, N4 VGGT-1B).  The N-segment loop (`inference.UnifiedLoopConsistencyPipeline`) takes them as three
callables; a deployment plugs the real networks in with `--stages package.module:factory`, where `factory(args)` returns an
object with the same three methods as `SyntheticStages`.

`SyntheticStages` is NOT a model: it is a deterministic, weight-free stand-in with the right shapes and value ranges so that
the entry points (`unified_loop_consistency.py`, `run_single_segment.sh`, `run_unified_pipeline.sh`) run end to end on a box
that has no checkpoints (there is no network in the build environment).  All of it runs on the device.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .geometry import xyz_euler_to_four_by_four_matrix_batch


class SyntheticStages:
    def __init__(self, cross_attention_dim=1024, camera_params=None, depth_hw=(48, 64), vae_scale=0.18215):
        self.xdim, self.cam, self.depth_hw, self.vae_scale = cross_attention_dim, camera_params, depth_hw, vae_scale
        # fixed 3->4 / 4->3 colour mixes standing for the VAE's channel change
        self.enc = torch.tensor([[0.6, 0.3, 0.1], [-0.2, 0.7, -0.5], [0.5, -0.5, 0.0], [0.3, 0.3, 0.3]])
        self.dec = torch.linalg.pinv(self.enc)

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


def load_stages(spec, args, **kw):
    """spec 'pkg.mod:factory' -> factory(args); None -> SyntheticStages."""
    if not spec:
        return SyntheticStages(**kw)
    import importlib
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn or "make_stages")(args)
