"""-m 'not gpu': pins the CLIP-path oracle (oracle/clip_ref.py) -- the encoder restatement against the output of the real
transformers.CLIPVisionModelWithProjection (tests/golden/clip_tiny.npz, oracle/make_goldens_clip.py), the antialiased resize
restatement against the reference's own `_resize_with_antialiasing` (tests/golden/resize_antialias.npz)."""
import numpy as np
import torch

from conftest import rel_l2


def test_clip_oracle_matches_transformers_golden(golden_dir):
    from evoworld_amd.clip import DEFAULT_CLIP_CONFIG, random_clip_state_dict
    from oracle.clip_ref import CLIPVisionRef, tiny_clip_config
    g = np.load(f"{golden_dir}/clip_tiny.npz")
    cfg = tiny_clip_config()
    sd = {k: v.half().float() for k, v in random_clip_state_dict({**DEFAULT_CLIP_CONFIG, **cfg}, 0).items()}
    y = CLIPVisionRef(**cfg).load_state_dict(sd)(torch.tensor(g["x"]))
    assert rel_l2(y, torch.tensor(g["image_embeds"])) < 1e-5


def test_resize_oracle_matches_reference_golden(golden_dir):
    from oracle.clip_ref import resize_with_antialiasing_ref
    g = np.load(f"{golden_dir}/resize_antialias.npz")
    y = resize_with_antialiasing_ref(torch.tensor(g["x"]), (28, 28))
    np.testing.assert_allclose(y.numpy(), g["y"], atol=2e-6)
