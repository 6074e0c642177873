"""-m "not gpu": pins the CHECKER.  tests/golden/pipeline_glue.npz was captured from a run of the reference's own
StableVideoDiffusionPipeline.__call__ (evoworld/pipeline/pipeline_evoworld.py:456-741; oracle/make_goldens_pipeline.py); the
fp32 restatement every GPU pipeline test compares against (oracle/pipeline_ref.py) must reproduce it: conditioning assembly,
RNG draw order, mask_mem, CFG duplication, added_time_ids, guidance ramp, per-step model input, CFG combine + Euler step."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_glue.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def oracle_unet(gold):
    from evoworld_amd.unet import DEFAULT_CONFIG, random_state_dict
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef, tiny_config
    cfg = tiny_config()
    ref = UNetSpatioTemporalConditionModelRef(**cfg).eval()
    ref.load_state_dict(random_state_dict({**DEFAULT_CONFIG, **cfg}, int(gold["unet_seed"])))
    return cfg, ref


@pytest.mark.parametrize("tag", ["mem", "mask"])
def test_oracle_glue_reproduces_the_reference_run(gold, oracle_unet, tag):
    from oracle.pipeline_ref import assemble_conditioning_ref, oracle_loop
    from oracle.standins import StandInCLIP, StandInVAE
    cfg, ref = oracle_unet
    T, steps = int(gold["T"]), int(gold["steps"])
    image = torch.from_numpy(gold[f"{tag}_image"]).float()
    memory = torch.from_numpy(gold[f"{tag}_memory"]).float()
    pl = torch.from_numpy(gold[f"{tag}_plucker"])
    gmin, gmax, fps, mb, aug = [float(v) for v in gold[f"{tag}_kwargs"]]
    gen = torch.Generator().manual_seed(-1)
    ehs, il = assemble_conditioning_ref(image, memory, StandInVAE(), StandInCLIP(cfg["cross_attention_dim"]), gen, aug,
                                        gold["image_mean"].tolist(), gold["image_std"].tolist())
    lat0 = torch.randn(1, T, 4, image.shape[-2] // 8, image.shape[-1] // 8, generator=gen)            # draw #2
    assert np.array_equal(gen.get_state().numpy()[:64], gold[f"{tag}_rng_state_after"])
    inputs, trace = [], []
    final = oracle_loop(ref, lat0, il, ehs, pl, T, steps, bool(gold[f"{tag}_mask_mem"]), trace=trace, ids=(fps - 1, mb, aug),
                        guidance=(gmin, gmax), inputs=inputs)
    x0 = torch.from_numpy(gold[f"{tag}_step0_latent_model_input"])
    # noisy channels and Plücker: same arithmetic -> fp32 round-off; VAE / CLIP branches go through the antialias restatement
    assert rel_l2(inputs[0][:, :, :4], x0[:, :, :4]) < 1e-6
    assert torch.equal(inputs[0][:, :, 12:], x0[:, :, 12:])                       # Plücker duplicated on BOTH rows, not zeroed
    assert rel_l2(inputs[0][:, :, 4:12], x0[:, :, 4:12]) < 1e-6
    assert float(x0[0, :, 4:12].abs().max()) == 0.0                               # negative image latents are zeros
    if bool(gold[f"{tag}_mask_mem"]):
        assert float(x0[:, :, 8:12].abs().max()) == 0.0                           # memory latents zeroed on both rows
    else:
        assert float(x0[1, :, 8:12].abs().max()) > 0.0
    e2 = torch.from_numpy(gold[f"{tag}_image_embeddings"])
    assert float(e2[0].abs().max()) == 0.0 and rel_l2(ehs[0], e2[1]) < 1e-5
    assert np.array_equal(gold[f"{tag}_added_time_ids"], np.array([[fps - 1, mb, aug]] * 2, np.float32))
    assert np.allclose(gold[f"{tag}_guidance_scale"].reshape(-1), np.linspace(gmin, gmax, T), atol=1e-6)
    assert rel_l2(inputs[1], torch.from_numpy(gold[f"{tag}_step1_latent_model_input"])) < 1e-5
    for k, i in enumerate((0, 1, 2, steps - 1)):
        assert rel_l2(trace[i], torch.from_numpy(gold[f"{tag}_latents_after_step"][k])) < 2e-5
    e = rel_l2(final, torch.from_numpy(gold[f"{tag}_final_latents"]))
    print(f"oracle glue vs reference run [{tag}]: final latents rel-L2 {e:.2e}")
    assert e < 2e-5
