"""TEST INFRASTRUCTURE -- the ORACLE side of tests/test_gpu_pipeline.py::test_full_size_clip_vs_oracle, run on host cores without a GPU
(round 5, VERDICT r4 item 6a: the full-size 25-step clip under the SURVEY 8d weight protocol -- oracle on UN-rounded fp32 weights).

The fp32 CPU oracle needs ~3 min per denoise step on 32 threads and an hour of a `gpurun` call would be spent on host arithmetic; the oracle
loop does not depend on the HIP path (same seeds -> same inputs), so it runs here instead and leaves every step's latents in a file in the
tree (tests/_ckpt/, git-ignored, shipped by gpurun).  The GPU test then resumes from `done == steps` (EW_FULL_PARITY_CKPT) and only compares.

    python tools/oracle_full_clip_cpu.py [--steps 25] [--fp32-weights 1] [--threads 6] [--out tests/_ckpt/clip_oracle_fp32w.pt]

Resumable: an existing output file with the same (steps, protocol) is continued after its last finished step."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--fp32-weights", type=int, default=1)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default="tests/_ckpt/clip_oracle_fp32w.pt")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from evoworld_amd.unet import DEFAULT_CONFIG, random_state_dict
    from oracle.pipeline_ref import oracle_loop
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef
    # the same construction as the test (seeds 11 / 13): keep the two in step
    cfg = dict(in_channels=18, out_channels=4, block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
               projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
               num_attention_heads=(5, 10, 20, 20), num_frames=25)
    fp32w = bool(a.fp32_weights)
    sd = {k: (v.float() if fp32w else v.half().float()) for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 11).items()}
    ref = UNetSpatioTemporalConditionModelRef(**cfg).eval()
    ref.load_state_dict(sd)
    del sd
    T, h, w = 25, 72, 128
    g = torch.Generator().manual_seed(13)
    lat0, il = torch.randn(1, T, 4, h, w, generator=g), torch.randn(1, T + 1, 4, h, w, generator=g)
    ehs, pl = torch.randn(1, 1, cfg["cross_attention_dim"], generator=g), torch.randn(1, T, 6, h, w, generator=g)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    st = {"steps": a.steps, "fp32w": fp32w, "done": 0, "lat": None, "trace": {}}
    if os.path.exists(a.out):
        old = torch.load(a.out)
        if old.get("steps") == a.steps and old.get("fp32w") == fp32w:
            st = old
            print(f"resuming after step {st['done']}", flush=True)
    t0 = time.time()
    KEEP = {1, 5, 10, 15, 20, 23, 24}

    def on_step(i, lat):
        st["done"], st["lat"] = i + 1, lat.clone()
        if (i + 1) in KEEP or i + 1 == a.steps:      # fp32 copies of a subset of the steps for the curve (fp16 copies would add 2.8e-4 of their own)
            st["trace"][i + 1] = lat.clone()
        torch.save(st, a.out + ".tmp")
        os.replace(a.out + ".tmp", a.out)
        print(f"oracle step {i + 1}/{a.steps} done ({time.time() - t0:.0f} s)", flush=True)

    with torch.no_grad():
        oracle_loop(ref, lat0, il, ehs, pl, T, a.steps, start=st["done"], lat_start=st["lat"], on_step=on_step)
    print("oracle clip complete:", a.out, flush=True)


if __name__ == "__main__":
    main()
