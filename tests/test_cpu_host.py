"""-m 'not gpu': host logic + the C-ABI library loads and exports every symbol include/evoworld_hip.h declares."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from evoworld_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "evoworld_hip.h")).read()
    declared = set(re.findall(r"\b(ew_[a-z0-9_]+)\s*\(", hdr)) - {"ew_gemm_args"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.ew_abi_version() == _lib.ABI_VERSION


def test_gemm_args_struct_matches_header():
    from evoworld_amd._lib import GemmArgs
    hdr = open(os.path.join(ROOT, "include", "evoworld_hip.h")).read()
    body = hdr[hdr.index("typedef struct ew_gemm_args {"):hdr.index("} ew_gemm_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("*", " ").split(",")
        names.append(parts[0].split()[-1])
        names += [p.strip() for p in parts[1:]]
    assert names == [f[0] for f in GemmArgs._fields_]


def test_ops_refuse_cpu_tensors():
    from evoworld_amd import ops
    from evoworld_amd._lib import EvoWorldHipError
    with pytest.raises(EvoWorldHipError):
        ops.linear(torch.zeros(4, 64, dtype=torch.float16), torch.zeros(4, 64, dtype=torch.float16))


def test_param_spec_matches_oracle_state_dict():
    from evoworld_amd.unet import DEFAULT_CONFIG, param_spec
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef, tiny_config
    for cfg in (tiny_config(), {}):
        with torch.device("meta"):
            ref = UNetSpatioTemporalConditionModelRef(**cfg)
        sd = ref.state_dict()
        spec = param_spec({**DEFAULT_CONFIG, **cfg})
        assert list(spec.keys()) == list(sd.keys())
        for k, (shape, _) in spec.items():
            assert tuple(sd[k].shape) == tuple(shape), k
    # the public SVD-XT U-Net has 1,524,623,082 parameters with in_channels=8; EvoWorld widens conv_in to 18
    n = sum(int(np.prod(s)) for s, _ in param_spec(DEFAULT_CONFIG).values())
    assert n == 1_524_623_082 + (18 - 8) * 320 * 9


def test_scheduler_known_answers():
    from evoworld_amd.scheduler import EulerDiscreteScheduler
    s = EulerDiscreteScheduler()
    s.set_timesteps(25)
    sig = s.sigmas.numpy()
    np.testing.assert_allclose(sig[:4], [700.0, 545.72925, 421.56912, 322.45367], rtol=2e-6)   # SURVEY.md §8a S1 KAT
    np.testing.assert_allclose(sig[-4:], [0.02480258, 0.0078825, 0.002, 0.0], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(s.timesteps.numpy()[[0, 1, 2, -1]], [1.63777, 1.5755308, 1.510996, -1.553652], rtol=2e-6)
    assert abs(s.init_noise_sigma - 700.000714) < 1e-5
    # v-prediction step == the c_skip / c_out form of the reference's training loss (train_evoworld.py:698-700)
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(3, 4, generator=g) * 500, torch.randn(3, 4, generator=g)
    out = s.step(v, s.timesteps[0], x)
    sg = s.sigmas[0]
    c_skip, c_out = 1 / (sg ** 2 + 1), -sg / (sg ** 2 + 1) ** 0.5
    assert torch.allclose(out.pred_original_sample, c_skip * x + c_out * v, rtol=1e-6, atol=1e-6)


def test_scheduler_custom_sigmas_and_stage_duck_types():
    """`sigmas=` through retrieve_timesteps (pipeline_evoworld.py:180-190): used as given, timesteps = 0.25 ln sigma; the
    synthetic stage provider exposes the pipeline's vae / image_encoder duck types (host tensors: no kernel involved)."""
    from evoworld_amd.scheduler import EulerDiscreteScheduler
    from evoworld_amd.stages import SyntheticStages
    s = EulerDiscreteScheduler()
    s.set_timesteps(sigmas=[700.0, 20.0, 1.0, 0.0])
    assert s.num_inference_steps == 3 and s.sigmas.tolist() == [700.0, 20.0, 1.0, 0.0]
    np.testing.assert_allclose(s.timesteps.numpy(), 0.25 * np.log([700.0, 20.0, 1.0]), rtol=1e-6, atol=1e-7)
    s2 = EulerDiscreteScheduler()
    s2.set_timesteps(5)
    s.set_timesteps(sigmas=s2.sigmas.tolist())
    assert torch.equal(s.sigmas, s2.sigmas) and torch.equal(s.timesteps, s2.timesteps)
    with pytest.raises(ValueError):
        s.set_timesteps(sigmas=[1.0])
    with pytest.raises(ValueError):
        s.set_timesteps(timesteps=[1, 2])
    st = SyntheticStages(cross_attention_dim=64)
    x = torch.rand(3, 3, 32, 64) * 2 - 1
    z = st.vae.encode(x).latent_dist.mode()
    assert z.shape == (3, 4, 4, 8) and len(st.vae.config.block_out_channels) == 4
    assert st.vae.decode(z, num_frames=3).sample.shape == (3, 3, 32, 64)
    assert st.image_encoder(torch.rand(1, 3, 224, 224)).image_embeds.shape == (1, 64)


def test_window_rng_draws_are_memoised_bit_exactly():
    """The reference re-seeds one CPU generator per window (navigator_evoworld.py:198): the pipeline memoises the draws on the
    generator state; a hit must return the same values AND leave the generator where a real draw would have."""
    from evoworld_amd.pipeline import _RANDN_CACHE, _randn_like_reference
    _RANDN_CACHE.clear()
    g = torch.manual_seed(-1)
    a1 = _randn_like_reference((2, 3, 5), g, "cpu")
    b1 = _randn_like_reference((1, 7), g, "cpu")
    tail1 = torch.randn(4, generator=g)
    g = torch.manual_seed(-1)
    a2 = _randn_like_reference((2, 3, 5), g, "cpu")              # cache hits
    b2 = _randn_like_reference((1, 7), g, "cpu")
    tail2 = torch.randn(4, generator=g)
    ref = torch.manual_seed(-1)
    want_a, want_b, want_t = torch.randn((2, 3, 5), generator=ref), torch.randn((1, 7), generator=ref), torch.randn(4, generator=ref)
    assert torch.equal(a2, a1) and torch.equal(b2, b1) and len(_RANDN_CACHE) == 2
    a2.zero_()                                                  # callers own what they get: the memoised tensor is not aliased
    assert torch.equal(_randn_like_reference((2, 3, 5), torch.manual_seed(-1), "cpu"), want_a)
    assert torch.equal(a1, want_a) and torch.equal(b1, want_b) and torch.equal(tail1, want_t) and torch.equal(tail2, want_t)
    g = torch.manual_seed(7)                                    # another seed: no false hit
    assert not torch.equal(_randn_like_reference((2, 3, 5), g, "cpu"), a1)


def test_geometry_and_rays_golden(golden_dir):
    from evoworld_amd.geometry import xyz_euler_to_four_by_four_matrix_batch, xyz_euler_to_three_by_four_matrix_batch
    from evoworld_amd.plucker import equirectangular_to_ray
    g = np.load(f"{golden_dir}/plucker.npz")
    rp = torch.tensor(g["rand_poses"])
    for rel, k3, k4 in ((False, "rand_c2w_abs", "rand_c2w4_abs"), (True, "rand_c2w_rel", "rand_c2w4_rel")):
        np.testing.assert_allclose(xyz_euler_to_three_by_four_matrix_batch(rp, relative=rel).numpy(), g[k3], atol=1e-6)
        np.testing.assert_allclose(xyz_euler_to_four_by_four_matrix_batch(rp, relative=rel).numpy(), g[k4], atol=1e-6)
    for tag in ("ps01", "ps10"):
        c2w = xyz_euler_to_three_by_four_matrix_batch(torch.tensor(g[f"cam_{tag}"]), relative=True)
        np.testing.assert_allclose(c2w.numpy(), g[f"c2w_{tag}"], atol=1e-6)
    np.testing.assert_allclose(g["c2w_ps01"][24], [[0.691786, 0, 0.722103, 0.721130], [0, 1, 0, 0],
                                                   [-0.722103, 0, 0.691786, 0.692799]], atol=2e-6)
    np.testing.assert_allclose(equirectangular_to_ray(72, 128).astype(np.float32), g["rays_72x128"], atol=1e-7)
    np.testing.assert_allclose(equirectangular_to_ray(8, 16).astype(np.float32), g["rays_8x16"], atol=1e-7)


def test_split_operand_packing_reproduces_fp32_conv():
    """Round 5 host logic (no GPU): conv_in with both operands split inside its K padding.  The fp16 row [x_hi | x_lo | x_hi 2^-10] against the
    fp16-rounded pack of [W_hi | W_hi | W_lo 2^10] -- evaluated in fp64 exactly as the MFMA's exact products + fp32-or-better accumulation would --
    reproduces the fp32 conv to ~1e-6 (x_lo W_lo is the only term left), where single-rounded operands give ~3e-4; and the [W_hi | W_lo] pack of
    conv_out / the level-0 projections removes the weight term.  Also checks the layout contract ew_nchw_f32_to_nhwc_split_f16 implements."""
    import torch.nn.functional as F
    from evoworld_amd.unet import CPAD_IN, SPLIT_DUP_LOG2, hi_lo, split_conv_in_weight, split_in_offsets, split_input_row
    assert split_in_offsets(18) == (20, 40) and split_in_offsets(21) is None and split_in_offsets(4) == (4, 8)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 18, 12, 20, generator=g)
    w = (torch.rand(32, 18, 3, 3, generator=g) * 2 - 1) / (18 * 9) ** 0.5
    sp = split_in_offsets(18)
    row = split_input_row(x, sp)
    assert row.shape == (2, CPAD_IN, 12, 20) and row.dtype == torch.float16
    hi = x.half()
    assert torch.equal(row[:, :18], hi) and torch.equal(row[:, 20:38], (x - hi.float()).half())
    assert torch.equal(row[:, 40:58], (hi.float() * 2.0 ** -SPLIT_DUP_LOG2).half())
    assert (row[:, 18:20] == 0).all() and (row[:, 38:40] == 0).all() and (row[:, 58:] == 0).all()
    wp = split_conv_in_weight(w, sp).half()                                    # what the loader rounds and packs
    want = F.conv2d(x.double(), w.double(), padding=1)
    got = F.conv2d(row.double(), wp.double(), padding=1)
    single = F.conv2d(hi.double(), w.half().double(), padding=1)
    e_split = float((got - want).norm() / want.norm())
    e_single = float((single - want).norm() / want.norm())
    assert e_split < 3e-6 and e_single > 1e-4, (e_split, e_single)
    # second K block over the same A: x_hi (W_hi + W_lo) against x_hi W_hi
    w_hi, w_lo = hi_lo(w)
    assert torch.equal(w_hi + w_lo, w)
    two = F.conv2d(torch.cat([hi, hi], 1).double(), torch.cat([w_hi, w_lo], 1).half().double(), padding=1)
    ref_a = F.conv2d(hi.double(), w.double(), padding=1)                        # activations rounded, weights exact
    assert float((two - ref_a).norm() / ref_a.norm()) < 3e-6 < float((single - ref_a).norm() / ref_a.norm())
