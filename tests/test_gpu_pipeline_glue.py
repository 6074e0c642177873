"""-m gpu: evoworld_amd.pipeline / evoworld_amd.inference.Navigator against goldens captured from RUNS OF THE REFERENCE'S OWN
CODE in the build container (oracle/make_goldens_pipeline.py): StableVideoDiffusionPipeline.__call__
(evoworld/pipeline/pipeline_evoworld.py:456-741) and Navigator.move_forward (evoworld/inference/navigator_evoworld.py:173-231).

Weight protocol of SURVEY §8d: the reference run used the UN-rounded fp32 random weights; the HIP model packs the same
state dict to fp16.  Tolerances: tensors the glue only moves / scales are compared at the fp16 storage grid of the U-Net input
buffer (2^-11 relative per element -> 5e-4 rel-L2 bound, measured ~2.9e-4); the 25-step clip at the north_star's 1e-3."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "pipeline_glue.npz"))


@pytest.fixture(scope="module")
def hip_unet(gold):
    from evoworld_amd.unet import DEFAULT_CONFIG, UNetSpatioTemporalConditionModel, random_state_dict
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    sd = random_state_dict({**DEFAULT_CONFIG, **cfg}, int(gold["unet_seed"]))          # fp32; packed to fp16 by the loader
    return cfg, UNetSpatioTemporalConditionModel(**cfg).load_state_dict(sd, device="cuda")


@pytest.mark.parametrize("tag", ["mem", "mask"])
def test_pipeline_call_vs_reference_run(gold, hip_unet, tag):
    from types import SimpleNamespace
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from oracle.standins import StandInCLIP, StandInVAE
    cfg, unet = hip_unet
    T, H, W, steps = int(gold["T"]), int(gold["H"]), int(gold["W"]), int(gold["steps"])
    h, w = H // 8, W // 8
    image = torch.from_numpy(gold[f"{tag}_image"]).float().cuda()
    memory = torch.from_numpy(gold[f"{tag}_memory"]).float().cuda()
    pl = torch.from_numpy(gold[f"{tag}_plucker"])
    gmin, gmax, fps, mb, aug = [float(v) for v in gold[f"{tag}_kwargs"]]
    fe = SimpleNamespace(image_mean=gold["image_mean"].tolist(), image_std=gold["image_std"].tolist())
    pipe = StableVideoDiffusionPipeline(unet=unet, vae=StandInVAE(), image_encoder=StandInCLIP(cfg["cross_attention_dim"]),
                                        feature_extractor=fe)
    seen = []
    real = unet.forward_nhwc

    def spy(x_in, t, ehs, ids, *a, **k):
        seen.append(dict(x=x_in.clone(), t=float(t), ehs=ehs.clone(), ids=ids.clone()))
        return real(x_in, t, ehs, ids, *a, **k)
    unet.forward_nhwc = spy
    trace = []
    try:
        gen = torch.manual_seed(-1)                                    # navigator_evoworld.py:198
        out = pipe(image, height=H, width=W, num_frames=T, num_inference_steps=steps, generator=gen, decode_chunk_size=8,
                   output_type="latent", plucker_embedding=pl, memorized_pixel_values=memory, mask_mem=bool(gold[f"{tag}_mask_mem"]),
                   min_guidance_scale=gmin, max_guidance_scale=gmax, fps=int(fps), motion_bucket_id=int(mb), noise_aug_strength=aug,
                   callback_on_step_end=lambda p, i, t, kw: trace.append(kw["latents"].detach().cpu().clone()) or {}).frames
    finally:
        unet.forward_nhwc = real
    assert len(seen) == steps
    # RNG: the generator was advanced by exactly the reference's two draws, in the reference's order
    assert np.array_equal(gen.get_state().numpy()[:64], gold[f"{tag}_rng_state_after"])

    def unpack(x):   # [2*T*h*w, 64] fp16 channels-last -> [2,T,18,h,w] fp32
        return x[:, :18].float().reshape(2, T, h, w, 18).permute(0, 1, 4, 2, 3).cpu()
    x0, g0 = unpack(seen[0]["x"]), torch.from_numpy(gold[f"{tag}_step0_latent_model_input"])
    e_noisy, e_img, e_pl = rel_l2(x0[:, :, :4], g0[:, :, :4]), rel_l2(x0[1, :, 4:12], g0[1, :, 4:12]), rel_l2(x0[:, :, 12:], g0[:, :, 12:])
    print(f"[{tag}] step-0 model input vs reference run: noisy {e_noisy:.2e}  image/memory latents {e_img:.2e}  plucker {e_pl:.2e}")
    assert e_noisy < 5e-4 and e_img < 5e-4 and e_pl < 5e-4
    assert float(x0[0, :, 4:12].abs().max()) == 0.0                   # negative image latents: zeros (:320-326)
    assert torch.equal(x0[0, :, 12:], x0[1, :, 12:])                  # Plücker duplicated, not zeroed (:632)
    if bool(gold[f"{tag}_mask_mem"]):
        assert float(x0[:, :, 8:12].abs().max()) == 0.0               # :626-628
    assert abs(seen[0]["t"] - float(gold[f"{tag}_step0_timestep"])) < 1e-5
    e2 = torch.from_numpy(gold[f"{tag}_image_embeddings"])
    assert float(seen[0]["ehs"][0].abs().max()) == 0.0
    # (round 6: the pipeline hands the forward the embeddings already on the fp16 grid the forward always converted them to: compare on that grid)
    e_clip = rel_l2(seen[0]["ehs"][1].float().cpu(), e2[1].half().float())
    print(f"[{tag}] image embeddings (HIP antialias resize + normalisation -> stand-in CLIP) {e_clip:.2e}")
    assert e_clip < 1e-4
    assert np.array_equal(seen[0]["ids"].float().cpu().numpy(), gold[f"{tag}_added_time_ids"])
    assert np.allclose(pipe.guidance_scale.cpu().numpy(), gold[f"{tag}_guidance_scale"], atol=1e-6)
    e1 = rel_l2(unpack(seen[1]["x"]), torch.from_numpy(gold[f"{tag}_step1_latent_model_input"]))
    curve = [rel_l2(trace[i], torch.from_numpy(gold[f"{tag}_latents_after_step"][k])) for k, i in enumerate((0, 1, 2, steps - 1))]
    e = rel_l2(out.cpu(), torch.from_numpy(gold[f"{tag}_final_latents"]))
    print(f"[{tag}] step-1 input {e1:.2e}; latents after steps 1,2,3,{steps}: " + " ".join(f"{c:.2e}" for c in curve)
          + f"; FINAL (fp32 reference weights vs fp16-packed) rel-L2 {e:.3e}")
    assert e1 < 6e-4
    assert e < 1e-3


@pytest.mark.parametrize("tag", ["full", "short"])
def test_navigator_window_vs_reference_run(tag):
    """What one window hands to the pipeline: Plücker of the (extended) segment, the re-seeded default generator, mask_mem,
    the fixed scalars, a clone of the memory (navigator_evoworld.py:173-214)."""
    from types import SimpleNamespace
    from evoworld_amd.inference import Navigator
    g = np.load(os.path.join(GOLDEN, "navigator_glue.npz"))
    rec = {}

    def pipe(image, **kw):
        rec.update(kw, image=image, rng=kw["generator"].get_state().clone(), is_default=kw["generator"] is torch.default_generator)
        return SimpleNamespace(frames="frames")
    nav = Navigator(pipe, height=64, width=128, num_frames=25, fps=7)
    gen = torch.Generator().manual_seed(31)
    nav.memorized_images = (torch.rand(1, 25, 3, 64, 128, generator=gen) * 2 - 1).cuda()
    image = torch.from_numpy(g[f"{tag}_image"]).cuda()
    segment = [torch.from_numpy(r) for r in g[f"{tag}_segment"]]
    frames, n = nav.move_forward(image, segment, num_model_frames=25, num_inference_steps=7, noise_aug_strength=0.03,
                                 use_memory=bool(g[f"{tag}_use_memory"]))
    assert frames == "frames" and n == int(g[f"{tag}_n_frames_returned"])
    e = rel_l2(rec["plucker_embedding"].cpu(), torch.from_numpy(g[f"{tag}_plucker_embedding"]))
    print(f"navigator [{tag}] plucker of the window vs reference run: {e:.2e}")
    assert rec["plucker_embedding"].shape == (1, 25, 6, 8, 16) and e < 5e-6
    assert rec["mask_mem"] == bool(g[f"{tag}_mask_mem"])
    assert bool(g[f"{tag}_generator_is_default_reseeded"]) and rec["is_default"] and torch.equal(rec["rng"], torch.manual_seed(-1).get_state())
    got = [rec["num_frames"], rec["width"], rec["height"], rec["decode_chunk_size"], rec["motion_bucket_id"], rec["fps"], rec["num_inference_steps"]]
    assert np.array_equal(np.array(got, np.float64), g[f"{tag}_pipe_scalars"])
    assert rec["noise_aug_strength"] == float(g[f"{tag}_noise_aug_strength"])
    assert torch.equal(rec["image"].cpu(), torch.from_numpy(g[f"{tag}_image_passed"]))
    assert torch.equal(rec["memorized_pixel_values"], nav.memorized_images) and rec["memorized_pixel_values"] is not nav.memorized_images


def test_pipeline_batch_of_two_vs_reference_run(gold, hip_unet):
    """Batch B = 2 (pipeline_evoworld.py:573-578): the reference draws ONE augmentation-noise tensor and ONE latents tensor for the
    whole batch and runs a 2B-row U-Net batch; the build makes the same two draws and runs the clips one after the other."""
    from types import SimpleNamespace
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from oracle.standins import StandInCLIP, StandInVAE
    cfg, unet = hip_unet
    T, H, W, steps = int(gold["T"]), int(gold["H"]), int(gold["W"]), int(gold["steps"])
    h, w = H // 8, W // 8
    image = torch.from_numpy(gold["b2_image"]).float().cuda()
    memory = torch.from_numpy(gold["b2_memory"]).float().cuda()
    pl = torch.from_numpy(gold["b2_plucker"])
    fe = SimpleNamespace(image_mean=gold["image_mean"].tolist(), image_std=gold["image_std"].tolist())
    pipe = StableVideoDiffusionPipeline(unet=unet, vae=StandInVAE(), image_encoder=StandInCLIP(cfg["cross_attention_dim"]),
                                        feature_extractor=fe)
    seen = []
    real = unet.forward_nhwc

    def spy(x_in, *a, **k):
        seen.append(x_in[:, :18].float().reshape(2, T, h, w, 18).permute(0, 1, 4, 2, 3).cpu())
        return real(x_in, *a, **k)
    unet.forward_nhwc = spy
    try:
        gen = torch.manual_seed(-1)
        out = pipe(image, height=H, width=W, num_frames=T, num_inference_steps=steps, generator=gen, decode_chunk_size=8,
                   output_type="latent", plucker_embedding=pl, memorized_pixel_values=memory, mask_mem=False).frames
    finally:
        unet.forward_nhwc = real
    assert out.shape == (2, T, 4, h, w) and len(seen) == 2 * steps
    assert np.array_equal(gen.get_state().numpy()[:64], gold["b2_rng_state_after"])          # two draws, batch-sized, in order
    g0 = torch.from_numpy(gold["b2_step0_latent_model_input"])                               # rows: uncond clip 0, 1, cond clip 0, 1
    for b in range(2):
        mine = seen[b * steps]                                                                # [uncond, cond] of clip b at its step 0
        e = rel_l2(mine, torch.stack([g0[b], g0[2 + b]]))
        print(f"[b2] clip {b} step-0 model input vs reference run: {e:.2e}")
        assert e < 5e-4
    e = rel_l2(out.cpu(), torch.from_numpy(gold["b2_final_latents"]))
    print(f"[b2] final latents of the batch vs reference run: {e:.3e}")
    assert e < 1e-3
