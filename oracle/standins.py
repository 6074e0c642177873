"""TEST INFRASTRUCTURE -- deterministic, weight-free stand-ins with the duck types the reference pipeline expects of its
`vae` and `image_encoder` components (evoworld/pipeline/pipeline_evoworld.py:255-330, 355-383).  They exist so that the
reference's own `StableVideoDiffusionPipeline.__call__` can be RUN in the build container (oracle/make_goldens_pipeline.py)
and so that the HIP pipeline can be driven with the very same components on the GPU box (tests/test_gpu_pipeline_glue.py).
Closed-form arithmetic only (no RNG, no weights): identical on CPU and GPU up to fp32 rounding.  Not a product path."""
from types import SimpleNamespace

import torch


def _mix(rows, cols, device, dtype=torch.float32):
    """fixed [rows, cols] mixing matrix, entries in [-1, 1], closed form"""
    r = torch.arange(rows, device=device, dtype=torch.float32)[:, None]
    c = torch.arange(cols, device=device, dtype=torch.float32)[None, :]
    return torch.sin(1.0 + 1.7 * r + 0.9 * c + 0.31 * r * c).to(dtype)


class StandInVAE:
    """encode: 8x8 average pool -> 3->4 channel mix -> tanh;  decode: 4->3 channel mix -> nearest x8."""
    config = SimpleNamespace(block_out_channels=(1, 1, 1, 1), scaling_factor=0.18215, force_upcast=False)
    dtype = torch.float32

    def to(self, *a, **k):
        return self

    def encode(self, x):
        lat = torch.nn.functional.avg_pool2d(x.float(), 8)                            # [N,3,h,w]
        lat = torch.tanh(torch.einsum("oc,nchw->nohw", _mix(4, 3, x.device), lat))    # [N,4,h,w]
        return SimpleNamespace(latent_dist=SimpleNamespace(mode=lambda: lat))

    def forward(self, z, num_frames=None):      # signature probed by decode_latents (:365-366)
        return self.decode(z, num_frames=num_frames)

    def decode(self, z, num_frames=None):
        rgb = torch.einsum("oc,nchw->nohw", _mix(3, 4, z.device), z.float())
        return SimpleNamespace(sample=torch.nn.functional.interpolate(rgb, scale_factor=8.0, mode="nearest"))


class StandInCLIP:
    """image_encoder(pixel_values).image_embeds: 224x224 -> 7x7 average pool -> fixed projection to `dim`."""

    def __init__(self, dim):
        self.dim = dim
        self._p = torch.nn.Parameter(torch.zeros(1), requires_grad=False)           # `next(self.image_encoder.parameters()).dtype`

    def parameters(self):
        yield self._p

    def __call__(self, pixel_values):
        f = torch.nn.functional.adaptive_avg_pool2d(pixel_values.float(), 7).flatten(1)   # [N, 3*49]
        v = f @ _mix(f.shape[1], self.dim, pixel_values.device) / 12.0
        return SimpleNamespace(image_embeds=v)
