"""Full-size timing of the stages either side of the denoise loop (rows N1, N2): temporal VAE encode of the 26 conditioning
frames and decode of 25 frames at 576x1024 (chunks of 8, as the reference's decode_chunk_size), CLIP ViT-H/14 on one frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evoworld_amd.clip import CLIPVisionModelWithProjection, encode_image_preprocess
from evoworld_amd.vae import AutoencoderKLTemporalDecoder

dev = "cuda"
vae = AutoencoderKLTemporalDecoder.from_random(seed=0, device=dev)
clip = CLIPVisionModelWithProjection.from_random(seed=0, device=dev)
g = torch.Generator().manual_seed(0)
frames = (torch.rand(26, 3, 576, 1024, generator=g) * 2 - 1).to(dev)
lat = torch.randn(25, 4, 72, 128, generator=g).to(dev)


def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms_e, z = timed(lambda: vae.encode(frames).latent_dist.mode())
def dec():
    return torch.cat([vae.decode(lat[i:i + 8], num_frames=min(8, 25 - i)).sample for i in range(0, 25, 8)])
ms_d, img = timed(dec)
ms_c, emb = timed(lambda: clip(encode_image_preprocess(frames[:1] / 2 + 0.5)).image_embeds)
print(f"VAE encode 26 x 576x1024: {ms_e:.1f} ms  -> {tuple(z.shape)} finite={bool(torch.isfinite(z).all())}")
print(f"VAE decode 25 x 576x1024 (chunks of 8): {ms_d:.1f} ms -> {tuple(img.shape)} finite={bool(torch.isfinite(img).all())}")
print(f"CLIP ViT-H/14 preprocess + encode, 1 frame: {ms_c:.2f} ms -> {tuple(emb.shape)}")
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
