"""Golden for row N2 from the REAL third-party implementation: transformers.CLIPVisionModelWithProjection (installed in the
build container, 5.x; the reference pins 4.47 -- same architecture) instantiated with the tiny config of oracle/clip_ref.py
and the seeded weights of evoworld_amd.clip.random_clip_state_dict, evaluated on a seeded input.  Only the input and the
output are stored (weights are regenerated from the seed by the tests).
Usage (in the build container): python oracle/make_goldens_clip.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from evoworld_amd.clip import DEFAULT_CLIP_CONFIG, random_clip_state_dict
    from oracle.clip_ref import CLIPVisionRef, tiny_clip_config
    cfg = tiny_clip_config()
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(**cfg)).eval()
    sd = {k: v.half().float() for k, v in random_clip_state_dict({**DEFAULT_CLIP_CONFIG, **cfg}, 0).items()}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=g)
    with torch.no_grad():
        y = hf(pixel_values=x).image_embeds
        ref = CLIPVisionRef(**cfg).load_state_dict(sd)(x)
    print("HF vs oracle restatement rel-L2:", float((y - ref).norm() / y.norm()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_tiny.npz"), x=x.numpy(), image_embeds=y.numpy())


if __name__ == "__main__":
    main()
