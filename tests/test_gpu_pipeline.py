"""-m gpu: the pipeline call surface (evoworld/pipeline/pipeline_evoworld.py:456-741 semantics) on the HIP U-Net against
a CPU restatement of the same loop built from the oracle U-Net + oracle Euler/CFG step."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


class _FakeVAE:
    """Deterministic stand-in with the VAE duck type (encode(x).latent_dist.mode(), decode(z,num_frames).sample)."""
    class config:
        block_out_channels = (1, 1, 1, 1)
        scaling_factor = 0.18215
        force_upcast = False
    dtype = torch.float32

    def encode(self, x):
        lat = torch.nn.functional.avg_pool2d(x, 8)                       # [N,3,h,w]
        lat = torch.cat([lat, lat[:, :1] * 0.5], dim=1)                   # 4 channels
        return type("E", (), {"latent_dist": type("D", (), {"mode": staticmethod(lambda: lat)})})

    def decode(self, z, num_frames=None):
        up = torch.nn.functional.interpolate(z[:, :3], scale_factor=8.0, mode="nearest")
        return type("O", (), {"sample": up})


class _FakeCLIP:
    def __init__(self, dim):
        self.dim = dim

    def __call__(self, img):
        v = img.mean(dim=(1, 2, 3), keepdim=False)[:, None] * torch.linspace(-1, 1, self.dim, device=img.device)[None]
        return type("O", (), {"image_embeds": v})


def _oracle_loop(ref, lat0, il, ehs, pl, T, steps, mask_mem=False):
    from evoworld_amd.scheduler import EulerDiscreteScheduler
    from oracle.reproject_ref import euler_cfg_step_ref
    s = EulerDiscreteScheduler()
    s.set_timesteps(steps)
    lat = lat0 * s.init_noise_sigma
    il2 = torch.cat([torch.zeros_like(il), il])
    if mask_mem:
        il2[:, 1:] = 0
    cond = torch.cat([il2[:, 0:1].repeat(1, T, 1, 1, 1), il2[:, 1:], torch.cat([pl, pl])], dim=2)
    e2 = torch.cat([torch.zeros_like(ehs), ehs])
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    guid = torch.linspace(1.0, 3.0, T)
    for i in range(steps):
        sig, sign = float(s.sigmas[i]), float(s.sigmas[i + 1])
        x = torch.cat([torch.cat([lat, lat]) / (sig ** 2 + 1) ** 0.5, cond], dim=2)
        eps = ref(x, s.timesteps[i], e2, ids)
        lat = euler_cfg_step_ref(eps[0:1], eps[1:2], lat, guid, sig, sign)
    return lat


@pytest.fixture(scope="module")
def models():
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from evoworld_amd.unet import DEFAULT_CONFIG, UNetSpatioTemporalConditionModel, random_state_dict
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef, tiny_config
    cfg = tiny_config()
    sd = {k: v.half().float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}
    ref = UNetSpatioTemporalConditionModelRef(**cfg).eval()
    ref.load_state_dict(sd)
    unet = UNetSpatioTemporalConditionModel(**cfg).load_state_dict(sd, device="cuda")
    return cfg, ref, unet, StableVideoDiffusionPipeline


@pytest.mark.parametrize("mask_mem", [False, True])
def test_denoise_loop_vs_oracle(models, mask_mem):
    cfg, ref, unet, Pipe = models
    T, h, w, steps = 4, 16, 32, 3
    g = torch.Generator().manual_seed(5)
    lat0, il = torch.randn(1, T, 4, h, w, generator=g), torch.randn(1, T + 1, 4, h, w, generator=g)
    ehs, pl = torch.randn(1, 1, cfg["cross_attention_dim"], generator=g), torch.randn(1, T, 6, h, w, generator=g)
    pipe = Pipe(unet=unet)
    out = pipe(torch.zeros(1, 3, h * 8, w * 8), height=h * 8, width=w * 8, num_frames=T, num_inference_steps=steps, latents=lat0,
               output_type="latent", plucker_embedding=pl, image_latents=il, image_embeddings=ehs, mask_mem=mask_mem).frames
    want = _oracle_loop(ref, lat0, il, ehs, pl, T, steps, mask_mem)
    e = rel_l2(out.cpu(), want)
    print(f"denoise loop ({steps} steps, mask_mem={mask_mem}) rel-L2 {e:.3e}")
    assert e < 5e-3


def test_full_call_surface_with_component_duck_types(models):
    """image -> CLIP / VAE duck types -> conditioning assembly (RNG order: aug noise, then latents; CPU generator as in
    navigator_evoworld.py:198) -> loop -> decode -> PIL frames."""
    cfg, ref, unet, Pipe = models
    T, H, W, steps = 4, 128, 256, 2
    g = torch.Generator().manual_seed(9)
    image = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    memory = torch.rand(1, T, 3, H, W, generator=g) * 2 - 1
    pl = torch.randn(1, T, 6, H // 8, W // 8, generator=g)
    vae, clip = _FakeVAE(), _FakeCLIP(cfg["cross_attention_dim"])
    pipe = Pipe(unet=unet, vae=vae, image_encoder=clip)
    gen = torch.Generator().manual_seed(123)
    out = pipe(image.cuda(), height=H, width=W, num_frames=T, num_inference_steps=steps, generator=gen, decode_chunk_size=3,
               plucker_embedding=pl, memorized_pixel_values=memory.cuda(), mask_mem=False, noise_aug_strength=0.02)
    frames = out.frames[0]
    assert len(frames) == T and frames[0].size == (W, H)
    # CPU restatement of the assembly with the same generator
    gen = torch.Generator().manual_seed(123)
    img = torch.cat([image.unsqueeze(1), memory], dim=1) / 2.0 + 0.5
    ehs = clip(img[:, 0]).image_embeds.unsqueeze(1)
    flat = img.flatten(0, 1) * 2.0 - 1.0
    flat = flat + 0.02 * torch.randn(flat.shape, generator=gen)
    il = vae.encode(flat).latent_dist.mode().reshape(1, T + 1, 4, H // 8, W // 8)
    lat0 = torch.randn(1, T, 4, H // 8, W // 8, generator=gen)
    want = _oracle_loop(ref, lat0, il, ehs, pl, T, steps)
    got = pipe(image.cuda(), height=H, width=W, num_frames=T, num_inference_steps=steps, generator=torch.Generator().manual_seed(123),
               plucker_embedding=pl, memorized_pixel_values=memory.cuda(), output_type="latent").frames
    assert rel_l2(got.cpu(), want) < 5e-3


def test_input_validation_matches_reference(models):
    cfg, ref, unet, Pipe = models
    pipe = Pipe(unet=unet)
    pl = torch.zeros(1, 4, 6, 8, 16)  # shapes only matter for validation
    with pytest.raises(ValueError):
        pipe(torch.zeros(1, 3, 60, 128), height=60, width=128, num_frames=4, plucker_embedding=pl,
             image_latents=torch.zeros(1, 5, 4, 8, 16), image_embeddings=torch.zeros(1, 1, 64))      # H % 8 != 0
    with pytest.raises(ValueError):
        pipe(torch.zeros(1, 3, 64, 128), height=64, width=128, num_frames=4, plucker_embedding=pl,
             image_latents=torch.zeros(1, 4, 4, 8, 16), image_embeddings=torch.zeros(1, 1, 64))      # memory frames != T
