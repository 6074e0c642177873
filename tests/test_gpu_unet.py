"""-m gpu: the HIP U-Net (through the C ABI) against the fp32 CPU oracle on the same seeded weights/inputs.
Stated tolerance (BASELINE.json north_star): 1e-3 rel-L2 on the model OUTPUT.  fp16 MFMA operands, fp32 accumulation, split
(hi + lo8) residual stream: one forward measures 7.8e-4 (tiny config; 8.2e-4 before round 5's split operands), 8.3e-4 (T=25) and 7.5e-4 at the full config-2 size (round 3)
against the fp32 oracle -- of which 7.5e-4 ... 8.0e-4 is the floor of ANY design that feeds fp16 operands to the MFMA
(tests/analysis_fp16_floor.py: operands of every conv / linear AND of the attention matmuls rounded, nothing else).
The OUTPUT assertion is the north_star's 1e-3.  The per-block TAPS are internal tensors, not outputs: the error peaks at the
bottleneck (mid / up0: 1.17e-3 tiny, 1.01e-3 full size) and falls again towards the output; they are asserted at 1.4e-3
(1.2 x the worst tap)."""
TOL_FORWARD = 1.0e-3
TOL_TAP = 1.4e-3
# SURVEY §8d weight protocol (oracle keeps fp32 weights, HIP packs them to fp16): rounding 1.5 G weights to fp16 is one more
# operand-rounding term of the same size as the activations' -- with every operand rounded once the fp16-operand floor of ONE forward
# rises from 7.5e-4 to 1.05e-3 (tests/analysis_fp16_floor.py main(): "(a) fp16 operands only") and rounds 1-4 measured 1.083e-3 here.
# Round 5: the operands of conv_in, conv_out and the level-0 proj_in / proj_out are split (hi + lo; evoworld_amd/unet.py split_operands) --
# ~40 % of the weight term for < 1 % of the flops (tests/analysis_fp16_floor.py --per-group).  One forward of an fp32 checkpoint now measures
# 8.83e-4 at FULL SIZE (asserted at the north_star's 1.0e-3) and 9.44e-4 / 1.029e-3 / 8.54e-4 on three seeds of the tiny config (64-channel
# level 0: fewer, noisier terms).  Round 6: the ACTIVATION operand of conv_out and of the level-0 proj_in split as well (GroupNorm writes
# [x_hi | x_lo] rows, ew_groupnorm_apply_split_f16; +1.8 ms per full-size forward): 9.05e-4 / 9.77e-4 / 8.33e-4 -- every tiny seed inside the
# north_star's 1e-3, which is now the assertion here too (the worst seed has a 2.3 % margin: the tiny model is the noisy end; full size 8.3-8.8e-4).
TOL_FORWARD_FP32_WEIGHTS = 1.0e-3
TOL_FORWARD_FP32_WEIGHTS_FULL = 1.0e-3
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _setup(cfg, B, T, h, w, seed=0, fp16_representable_weights=True):
    """fp16_representable_weights=True: the random checkpoint is rounded to fp16 FIRST and loaded into both models (what a
    `variant="fp16"` checkpoint is) -- the assumption behind the 1e-3-per-forward figures.  False = SURVEY §8d's protocol: the oracle
    keeps the un-rounded fp32 weights (the reference runs weight_dtype=float32, unified_loop_consistency.py:188), the HIP loader
    packs the same fp32 dict to fp16."""
    from evoworld_amd.unet import DEFAULT_CONFIG, UNetSpatioTemporalConditionModel, random_state_dict
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef
    sd = random_state_dict({**DEFAULT_CONFIG, **cfg}, seed)
    if fp16_representable_weights:
        sd = {k: v.half().float() for k, v in sd.items()}
    ref = UNetSpatioTemporalConditionModelRef(**cfg).eval()
    ref.load_state_dict(sd, strict=True)
    m = UNetSpatioTemporalConditionModel(**cfg).load_state_dict(sd, device="cuda")
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, cfg["in_channels"], h, w, generator=g)
    ehs = torch.randn(B, 1, cfg["cross_attention_dim"], generator=g)
    ehs[0] = 0  # CFG uncond row
    ids = torch.tensor([[6.0, 127.0, 0.02]] * B)
    return m, ref, x, ehs, ids


def test_unet_tiny_vs_oracle():
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    B, T, h, w = 2, 4, 16, 32
    m, ref, x, ehs, ids = _setup(cfg, B, T, h, w)
    t = torch.tensor(1.6377)
    rt = {}
    want = ref(x, t, ehs, ids, taps=rt)
    gt = {}
    got = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False, taps=gt)[0]
    worst = 0.0
    for k, (ten, H, W) in gt.items():
        a = ten.float().reshape(B * T, H, W, -1).permute(0, 3, 1, 2).cpu()
        e = rel_l2(a, rt[k])
        print(f"tap {k:8s} rel-L2 {e:.2e}")
        worst = max(worst, e)
    e = rel_l2(got.cpu(), want)
    print(f"unet tiny forward rel-L2 {e:.3e}")
    assert torch.isfinite(got).all()
    assert worst < TOL_TAP and e < TOL_FORWARD


@pytest.mark.parametrize("seed", [0, 3, 7])
def test_unet_tiny_vs_oracle_fp32_weights(seed):
    """SURVEY §8d protocol: un-rounded fp32 weights in the oracle, the same dict packed to fp16 by the HIP loader (three weight / input seeds)."""
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    B, T, h, w = 2, 4, 16, 32
    out = []
    for rep in (True, False):
        m, ref, x, ehs, ids = _setup(cfg, B, T, h, w, seed=seed, fp16_representable_weights=rep)
        t = torch.tensor(1.6377)
        out.append(rel_l2(m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False)[0].cpu(), ref(x, t, ehs, ids)))
    print(f"unet tiny forward (seed {seed}) rel-L2: fp16-representable checkpoint {out[0]:.3e} | fp32 checkpoint (SURVEY 8d protocol) {out[1]:.3e}")
    assert out[0] < TOL_FORWARD and out[1] < TOL_FORWARD_FP32_WEIGHTS


@pytest.mark.parametrize("seed", [0, 3])
def test_unet_split_operands_buy_parity(seed, monkeypatch):
    """Round 5: conv_in with both operands split inside its K padding, conv_out / level-0 proj_in / proj_out with W = W_hi + W_lo as a second K
    block (EW_SPLIT_OPERANDS=1); round 6 (=2, the default): the GroupNorm outputs feeding conv_out and the level-0 proj_in as [x_hi | x_lo] rows as well.
    Same weights, same inputs, EW_SPLIT_OPERANDS=0 against 1 and 2: the split build must be closer to the fp32 oracle under BOTH
    weight protocols (by >= 8 % of the squared distance under the fp32 one) and conv_in's tap must be at fp32-storage level."""
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    B, T, h, w = 2, 4, 16, 32
    t = torch.tensor(1.6377)
    res = {}
    for rep in (True, False):
        for split in ("0", "1", "2"):
            monkeypatch.setenv("EW_SPLIT_OPERANDS", split)
            m, ref, x, ehs, ids = _setup(cfg, B, T, h, w, seed=seed, fp16_representable_weights=rep)
            assert m.split_operands == (split != "0") and (m.in_split is not None) == (split != "0") and m.split_acts == (split == "2")
            rt, gt = {}, {}
            want = ref(x, t, ehs, ids, taps=rt)
            got = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False, taps=gt)[0]
            ten, H, W = gt["conv_in"]
            tap = rel_l2(ten.float().reshape(B * T, H, W, -1).permute(0, 3, 1, 2).cpu(), rt["conv_in"])
            res[(rep, split)] = (rel_l2(got.cpu(), want), tap)
    for (rep, split), (e, tap) in res.items():
        print(f"seed {seed}, {'fp16-representable' if rep else 'fp32':18s} checkpoint, EW_SPLIT_OPERANDS={split}: forward rel-L2 {e:.3e}, conv_in tap {tap:.2e}")
    for rep in (True, False):
        assert res[(rep, "2")][0] < res[(rep, "1")][0] < res[(rep, "0")][0]     # round 6: the activation side of conv_out / level-0 proj_in buys more
        # conv_in output: 8.5e-7 (hi + lo8 storage) with an fp16-representable checkpoint, 1.7e-5 with an fp32 one (its BIAS is still a single
        # fp16 vector: 2^-12 of a U(+-0.08) bias on O(1) outputs), against 2.1e-4 / 3.0e-4 with single-rounded operands
        assert res[(rep, "1")][1] < 3e-5 < res[(rep, "0")][1]
    assert res[(False, "1")][0] ** 2 < 0.92 * res[(False, "0")][0] ** 2 and res[(False, "2")][0] < TOL_FORWARD_FP32_WEIGHTS


@pytest.mark.parametrize("mode", [0, 1])
def test_unet_level0_320_fused_feed_forward_modes(mode, monkeypatch):
    """A shrunken config whose FIRST level has the real 320 channels (head_dim 64, hidden 1280), so that its three feed-forwards go through
    ew_ff_geglu320_f16: mode 0 = LayerNorm + two GEMMs (the A/B baseline), 1 = LayerNorm kernel + fused kernel (default).  Both are checked
    against the fp32 oracle at the forward tolerance."""
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    cfg["block_out_channels"] = (320, 128, 256, 256)
    cfg["num_attention_heads"] = (5, 2, 4, 4)
    B, T, h, w = 2, 4, 16, 32
    monkeypatch.setenv("EW_FUSED_FF", str(mode))     # read at construction
    m, ref, x, ehs, ids = _setup(cfg, B, T, h, w, seed=11)
    assert m.fused_ff == mode
    t = torch.tensor(0.9)
    want = ref(x, t, ehs, ids)
    got = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False)[0]
    e = rel_l2(got.cpu(), want)
    print(f"unet 320-wide level 0, fused_ff={mode}: forward rel-L2 {e:.3e}")
    assert torch.isfinite(got).all() and e < TOL_FORWARD


def test_unet_tiny_T25_ragged_spatial():
    """T=25 (the real frame count) and a latent size whose deepest level has S=8 (< one KV tile)."""
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    cfg["num_frames"] = 25
    B, T, h, w = 2, 25, 16, 32
    m, ref, x, ehs, ids = _setup(cfg, B, T, h, w, seed=3)
    t = torch.tensor(-0.7)
    want = ref(x, t, ehs, ids)
    got = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False)[0]
    e = rel_l2(got.cpu(), want)
    print(f"unet tiny T=25 forward rel-L2 {e:.3e}")
    assert e < TOL_FORWARD


def test_unet_forward_bit_reproducible():
    """deterministic GroupNorm statistics (no atomics): two runs of the same forward are bit-identical"""
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    B, T, h, w = 2, 4, 16, 32
    m, _, x, ehs, ids = _setup(cfg, B, T, h, w)
    t = torch.tensor(0.5)
    a = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False)[0]
    b = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False)[0]
    assert torch.equal(a, b)


def test_unet_dead_cross_attention_identity():
    """The reference executes the single-token cross attention in full; the folded form must agree (oracle both ways)."""
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef, tiny_config
    ref = UNetSpatioTemporalConditionModelRef(**tiny_config()).eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 2, 18, 8, 16, generator=g)
    ehs = torch.randn(2, 1, 64, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    a = ref(x, torch.tensor(0.3), ehs, ids)
    b = ref(x, torch.tensor(0.3), ehs, ids, exec_dead_cross_attn=True)
    assert rel_l2(a, b) < 1e-5


@pytest.mark.skipif(__import__("os").environ.get("EW_SKIP_FULL_PARITY") == "1",
                    reason="EW_SKIP_FULL_PARITY=1: skips the ~3-4 min fp32 CPU oracle forward at the full config-2 size (quick local runs)")
def test_unet_full_size_forward_vs_oracle():
    """BASELINE.json configs[1] at full size: the real SVD-Xtend architecture (1.52 B parameters, random init kept in fp32 by the oracle),
    B=2 (CFG), T=25, 72x128 latents -- one HIP forward against one fp32 CPU-oracle forward on the same weights and inputs,
    with the per-block taps.  The tiny-config tests above run on every box; this one is the same comparison at the sizes the
    headline is measured on (stream-K tail, 256x320 tiles, S=9216 attention, 50-slab GroupNorm all engaged)."""
    import time
    cfg = dict(in_channels=18, out_channels=4, block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
               projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
               num_attention_heads=(5, 10, 20, 20), num_frames=25)           # evoworld/trainer/unet_plucker.py:69-94, in_channels 18
    B, T, h, w = 2, 25, 72, 128
    torch.set_num_threads(min(int(__import__("os").environ.get("EW_ORACLE_THREADS", "32")), __import__("os").cpu_count() or 1))
    # Round 6: SURVEY §8d's weight protocol is the DEFAULT (the reference runs weight_dtype = torch.float32, unified_loop_consistency.py:188: the
    # oracle keeps the un-rounded fp32 checkpoint, the HIP loader packs the same dict); EW_FULL_FP16_WEIGHTS=1 = the fp16-representable variant
    fp32w = __import__("os").environ.get("EW_FULL_FP16_WEIGHTS") != "1"
    seed = int(__import__("os").environ.get("EW_FULL_SEED", "7"))            # (builder-run: a second weight / input seed for the fp32-weights figure)
    m, ref, x, ehs, ids = _setup(cfg, B, T, h, w, seed=seed, fp16_representable_weights=not fp32w)
    t = torch.tensor(1.6377)
    gt = {}
    got = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False, taps=gt)[0]
    torch.cuda.synchronize()
    got2 = m(x.cuda(), t, ehs.cuda(), ids.cuda(), return_dict=False)[0]
    assert torch.equal(got, got2)                       # bit-reproducible at full size too (stream-K split is deterministic)
    t0 = time.time()
    rt = {}
    with torch.no_grad():
        want = ref(x, t, ehs, ids, taps=rt)
    print(f"fp32 oracle forward at full size: {time.time() - t0:.0f} s on {torch.get_num_threads()} threads")
    worst = 0.0
    for k, (ten, H, W) in gt.items():
        a = ten.float().reshape(B * T, H, W, -1).permute(0, 3, 1, 2).cpu()
        e = rel_l2(a, rt[k])
        print(f"full-size tap {k:8s} rel-L2 {e:.2e}")
        worst = max(worst, e)
    e = rel_l2(got.cpu(), want)
    print(f"unet FULL-SIZE forward (B=2, T=25, 72x128 latents, 1.52 B parameters, seed {seed}, {'fp32 checkpoint' if fp32w else 'fp16-representable checkpoint'}) rel-L2 {e:.3e}")
    assert torch.isfinite(got).all()
    assert worst < (1.8e-3 if fp32w else TOL_TAP) and e < (TOL_FORWARD_FP32_WEIGHTS_FULL if fp32w else TOL_FORWARD)
