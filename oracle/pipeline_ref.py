"""TEST INFRASTRUCTURE -- fp32 CPU restatement of the reference pipeline's glue (evoworld/pipeline/pipeline_evoworld.py:456-741)
around an oracle U-Net: conditioning assembly and the denoise loop.  PINNED: tests/test_cpu_pipeline_glue.py checks this file
against tests/golden/pipeline_glue.npz, which was captured from a run of the reference's own `__call__`
(oracle/make_goldens_pipeline.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
import torch

from .reproject_ref import euler_cfg_step_ref


def assemble_conditioning_ref(image, memory, vae, image_encoder, generator, noise_aug_strength=0.02,
                              image_mean=None, image_std=None):
    """image [1,3,H,W], memory [1,T,3,H,W] in [-1,1] -> (image_embeddings [1,1,X], image_latents [1,1+T,4,h,w]) before CFG
    duplication.  Draw #1 of `generator` is the [1+T,3,H,W] augmentation noise (:596-600).  (:570-612, :264-285)"""
    from .clip_ref import resize_with_antialiasing_ref
    if image_mean is None:
        from evoworld_amd.clip import CLIP_MEAN as image_mean, CLIP_STD as image_std
    img = torch.cat([image.unsqueeze(1), memory], dim=1) / 2.0 + 0.5                                     # :570, :579
    pv = (resize_with_antialiasing_ref(img[:, 0] * 2.0 - 1.0, (224, 224)) + 1.0) / 2.0                   # :275-277
    pv = (pv - torch.tensor(image_mean)[None, :, None, None]) / torch.tensor(image_std)[None, :, None, None]
    ehs = image_encoder(pv).image_embeds.unsqueeze(1)
    flat = img.flatten(0, 1) * 2.0 - 1.0                                                                 # VideoProcessor.preprocess
    flat = flat + noise_aug_strength * torch.randn(flat.shape, generator=generator)
    il = vae.encode(flat).latent_dist.mode()
    return ehs, il.reshape(1, -1, *il.shape[1:])


def oracle_loop(ref, lat0, il, ehs, pl, T, steps, mask_mem=False, trace=None, start=0, lat_start=None, stop_after=None,
                on_step=None, ids=(6.0, 127.0, 0.02), guidance=(1.0, 3.0), inputs=None):
    """fp32 CPU oracle of the denoise loop (:625-714).  lat0 [1,T,4,h,w] unit noise; il [1,1+T,4,h,w]; ehs [1,1,X]; pl [1,T,6,h,w].
    ids = (fps-1, motion_bucket_id, noise_aug_strength); guidance = (min, max).  start / lat_start resume it from the latents
    after step `start` (the full-size 25-step run is longer than one gpurun call), stop_after ends it early, on_step(i, lat) is
    called after every step, `inputs` (a list) receives every step's U-Net input."""
    from evoworld_amd.scheduler import EulerDiscreteScheduler
    s = EulerDiscreteScheduler()
    s.set_timesteps(steps)
    lat = lat0 * s.init_noise_sigma if lat_start is None else lat_start
    il2 = torch.cat([torch.zeros_like(il), il])
    if mask_mem:
        il2[:, 1:] = 0
    cond = torch.cat([il2[:, 0:1].repeat(1, T, 1, 1, 1), il2[:, 1:], torch.cat([pl, pl])], dim=2)
    e2 = torch.cat([torch.zeros_like(ehs), ehs])
    idt = torch.tensor([list(ids)] * 2, dtype=torch.float32)
    guid = torch.linspace(guidance[0], guidance[1], T)
    for i in range(start, steps if stop_after is None else min(steps, stop_after)):
        sig, sign = float(s.sigmas[i]), float(s.sigmas[i + 1])
        x = torch.cat([torch.cat([lat, lat]) / (sig ** 2 + 1) ** 0.5, cond], dim=2)
        if inputs is not None:
            inputs.append(x)
        eps = ref(x, s.timesteps[i], e2, idt)
        lat = euler_cfg_step_ref(eps[0:1], eps[1:2], lat, guid, sig, sign)
        if trace is not None:
            trace.append(lat.clone())
        if on_step is not None:
            on_step(i, lat)
    return lat
