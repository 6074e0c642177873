#!/usr/bin/env python3
"""bench.py -- EvoWorld per-clip denoise throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
A "step" = ONE CLIP through the hot path at BASELINE.json configs[1]: 576x1024 panorama, 25 frames, 25
EulerDiscrete steps, CFG on (U-Net batch 2), random-init SVD-Xtend U-Net (in_channels 18), synthetic inputs
already resident in HBM.  N>1: one process per GPU (torch.distributed.run), one independent clip per rank
(weak scaling, no collective inside the loop; the final latents are all-gathered).  Rank 0 prints ONE JSON line.
`value` = frames/s over the whole job = N * 25 * K / max-over-ranks(seconds).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_TFLOP_PER_FORWARD = 154.31      # SURVEY.md §8d, config 2, dead cross-attention work removed
PEAK_F16_DENSE_TFLOPS = 2500.0       # MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16 MFMA
NOMINAL_CLOCK_GHZ = 2.4              # the clock the dense peak is quoted at (256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz = 2.5 PFLOP/s)


def source_fingerprint():
    """sha256 over the kernel sources and the files that decide which kernels a forward launches: a recorded PMC measurement is
    only attached to the bench line when it was taken on exactly this code."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "evoworld_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "evoworld_amd", "csrc", "*.h")))
    files += [os.path.join(ROOT, "evoworld_amd", f) for f in ("unet.py", "ops.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(residual_mode):
    """HBM bytes of ONE U-Net forward from the committed rocprofv3 PMC run (profiles/rNN_hbm_traffic.json, latest round: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.sh).
    PMC collection needs its own profiler passes, so this is a RECORDED measurement: it carries the fingerprint of the sources
    and the residual-stream mode it was taken on, and is reported as stale (traffic = null in the bench line) when either
    differs from what is running."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))      # the latest round's record
    try:
        tr = json.load(open(files[-1]))
        tr["file"] = os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, IndexError):
        return None
    rec = tr.get("recorded_at", {})
    tr["stale"] = not (rec.get("source_fingerprint") == source_fingerprint() and rec.get("residual_stream") == residual_mode)
    return tr


def committed_clock():
    """Effective graphics clock of the forward's long kernels (diagnostic): GRBM_GUI_ACTIVE / kernel duration from the committed
    `rocprofv3 --pmc GRBM_GUI_ACTIVE` pass (profiles/rNN_clock.json, tools/pmc_clock.sh), time-weighted over the launches of >= 300 us (the
    counter window is a few us longer than the kernel, so short launches read high).  The chip clocks to its power budget: MFMA- and VALU-dense
    kernels sustain 1.8-2.2 GHz of the 2.4 GHz the 2.5 PFLOP/s peak is quoted at.  A recorded measurement, like `traffic`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_clock.json")))
    try:
        c = json.load(open(files[-1]))
        ks = [k for k in c["kernels"] if k["ms"] / k["launches"] >= 0.3]
        ghz = sum(k["ghz"] * k["ms"] for k in ks) / sum(k["ms"] for k in ks)
        return {"file": os.path.relpath(files[-1], ROOT), "ghz": ghz,
                "per_kernel": {k["kernel"]: round(k["ghz"], 3) for k in sorted(ks, key=lambda k: -k["ms"])[:8]}}
    except (OSError, ValueError, IndexError, KeyError, ZeroDivisionError):
        return None


def synth_inputs(T, h, w, seed, device):
    """SURVEY.md §8d config 2 synthetic clip (seeded on CPU so that RNG differences between stacks vanish)."""
    from evoworld_amd.geometry import xyz_euler_to_three_by_four_matrix_batch
    from evoworld_amd.plucker import equirectangular_to_ray, ray_c2w_to_plucker
    g = lambda s: torch.Generator().manual_seed(seed * 100 + s)
    latents = torch.randn(1, T, 4, h, w, generator=g(1))
    image_latents = torch.randn(1, T + 1, 4, h, w, generator=g(2))
    ehs = torch.randn(1, 1, 1024, generator=g(3))
    i = torch.arange(T, dtype=torch.float32)
    psi = 95.0 + 3.6 * i
    pose = torch.stack([0.04 * i * torch.sin(psi * math.pi / 180), torch.zeros(T), 0.04 * i * torch.cos(psi * math.pi / 180),
                        torch.zeros(T), psi, torch.zeros(T)], dim=1)
    c2w = xyz_euler_to_three_by_four_matrix_batch(pose, relative=True)
    plucker = ray_c2w_to_plucker(torch.tensor(equirectangular_to_ray(h, w)).float().to(device), c2w.to(device))[None]
    return latents.to(device), image_latents.to(device), ehs.to(device), plucker


def cpu_baseline(max_threads=32, frames=4):
    """The reference's CPU path (diffusers fp32 on PyTorch) restated by the oracle, timed on the host cores on a
    BOUNDED sample (~15-20 s of CPU work): one full-resolution (72x128 latents) U-Net forward over `frames` frames of ONE CFG
    row (B=1, T=`frames` -- 4 by default, so the temporal convs / temporal attention see more than one frame; frames >= 50 runs
    the complete B=2, T=25 forward instead), incl. the reference's dead cross-attention work, scaled linearly to a clip step
    (B=2, T=25 = 50 frames).  A one-frame forward runs first, untimed (thread pool, allocator and weight pages warm).  The extrapolation error was measured once
    against a COMPLETE full-size CPU forward (`bench.py --cpu-baseline-full`).  `cores` = the threads
    used (torch's CPU convs do not scale past a few dozen threads on this problem: 256 threads measured 20x slower than 32);
    `host_cores` = os.cpu_count()."""
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef
    host = os.cpu_count() or 1
    cores = min(host, max_threads)
    torch.set_num_threads(cores)
    with torch.device("meta"):
        m = UNetSpatioTemporalConditionModelRef()
    m = m.to_empty(device="cpu").eval()
    with torch.no_grad():
        gw = torch.Generator().manual_seed(1)
        for n, p in m.named_parameters():      # timing only, but real-looking values: random matrices, unit norm scales, zero biases
            if p.ndim > 1:
                p.uniform_(-0.02, 0.02, generator=gw)
            else:
                p.fill_(1.0 if n.endswith("weight") else 0.0)
    g = torch.Generator().manual_seed(0)
    full = frames >= 50
    B, TS = (2, 25) if full else (1, frames)
    x = torch.randn(B, TS, 18, 72, 128, generator=g)
    ehs = torch.randn(B, 1, 1024, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * B)
    m(x[:1, :1], torch.tensor(1.0), ehs[:1], ids[:1], exec_dead_cross_attn=True)          # untimed warm-up
    t0 = time.time()
    m(x, torch.tensor(1.0), ehs, ids, exec_dead_cross_attn=True)
    dt = time.time() - t0
    t_forward = dt * 50.0 / (B * TS)
    fps = 25.0 / (25 * t_forward)
    what = (f"1 COMPLETE fp32 oracle U-Net forward (B=2, T=25 at 72x128 latents: {dt:.1f} s on {cores} threads)" if full else
            f"1 fp32 oracle U-Net forward at 72x128 latents over B=1 x T={TS} frames ({dt:.1f} s on {cores} threads), "
            f"scaled x{50 / TS:g} to B=2,T=25")
    return {"value": fps, "unit": "frames/s", "cores": cores, "host_cores": host, "kind": "port",
            "sample": what + " and x25 denoise steps per clip (steps are shape-identical)"}


def kernel_breakdown(unet, pipe, latents, image_latents, ehs, plucker, T, h, w, top=8):
    """One extra (untimed) U-Net forward with a HIP-event pair around every C-ABI launch, grouped by the kernel the library
    actually dispatched (names as rocprofv3 prints them, so the committed profiles/ summary can be compared line by line):
    per kernel the launches per forward, the average duration, and for the MFMA kernels algorithmic flops / time against the
    dense fp16 peak.  Event pairs add ~2 % to the chain; the headline numbers come from the untouched timed region above."""
    import collections
    import ctypes
    from evoworld_amd import _lib
    lib = _lib.load()
    rec = []
    names = ("ew_gemm_f16", "ew_groupnorm_stats_f16", "ew_groupnorm_finalize", "ew_groupnorm_apply_f16", "ew_groupnorm_apply_split_f16", "ew_layernorm_f16",
             "ew_attn_spatial_f16", "ew_attn_spatial_log2_f16", "ew_attn_temporal_f16", "ew_ff_geglu320_f16")
    kname = {"ew_groupnorm_stats_f16": "gn_stats_kernel", "ew_groupnorm_finalize": "gn_finalize_kernel",
             "ew_groupnorm_apply_f16": "gn_apply_kernel", "ew_groupnorm_apply_split_f16": "gn_apply_kernel",
             "ew_layernorm_f16": "ln_kernel", "ew_attn_spatial_f16": "attn_spatial_kernel", "ew_attn_spatial_log2_f16": "attn_spatial_kernel",
             "ew_attn_temporal_f16": "attn_temporal_kernel", "ew_ff_geglu320_f16": "ff320_kernel"}
    orig = {n: getattr(lib, n) for n in names}

    def wrap(n):
        fn = orig[n]

        def timed(*a):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a)
            e.record()
            fl = 0.0
            if n == "ew_gemm_f16":
                g = a[0]._obj
                key = lib.ew_gemm_last_kernel().decode()
                fl = 2.0 * g.M * g.N * (g.c1 + g.c2) * {0: 1, 1: 9, 2: 3}[g.mode]
                if os.environ.get("EW_BENCH_BY_SHAPE"):
                    key += f" M={g.M} N={g.N} K={(g.c1 + g.c2) * {0: 1, 1: 9, 2: 3}[g.mode]}"
            else:
                key = kname[n]
                if os.environ.get("EW_BENCH_BY_SHAPE"):
                    if n == "ew_layernorm_f16":
                        key += f" rows={a[9]} C={a[10]}"
                    elif n.startswith("ew_groupnorm") and n != "ew_groupnorm_finalize":
                        o_ = 8 if n.endswith('apply_split_f16') else (6 if n.endswith('apply_f16') else 3)
                        key += f" slabs={a[o_]} rows={a[o_ + 1]} C={a[o_ + 2]}" + (" +lo" if a[1] else "") + (" +split out" if n.endswith('apply_split_f16') else "")
                    elif n.startswith("ew_attn_spatial"):
                        key += f" S={a[5]}"
                if n.startswith("ew_attn_spatial"):
                    fl = 4.0 * a[4] * a[6] * a[5] * a[5] * 64          # n_seq * heads * S^2 * head_dim
                elif n == "ew_ff_geglu320_f16":
                    g = a[0]._obj
                    fl = 2.0 * g.M * g.C * 2 * g.hidden + 2.0 * g.M * g.hidden * g.C   # up-projection (value + gate) + down-projection
            rec.append((key, fl, s, e))
            return r
        return timed
    try:
        for n in names:
            setattr(lib, n, wrap(n))
        x_in = torch.zeros(2 * T * h * w, 64, dtype=torch.float16, device=latents.device)
        ids = torch.tensor([[6.0, 127.0, 0.02]] * 2, device=latents.device)
        e2 = torch.cat([torch.zeros_like(ehs), ehs], 0).to(torch.float16)
        unet.forward_nhwc(x_in, 1.0, e2, ids, 2, T, h, w)
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(lib, n, orig[n])
    agg = collections.OrderedDict()
    for key, fl, s, e in rec:
        t = agg.setdefault(key, [0, 0.0, 0.0])
        t[0] += 1
        t[1] += s.elapsed_time(e)
        t[2] += fl
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    if os.environ.get("EW_BENCH_FULL_BREAKDOWN"):
        tot = sum(v[1] for _, v in rows)
        for key, (n, ms, fl) in rows:
            print(f"  {key:48s} n={n:4d} total {ms:7.2f} ms  avg {ms / n * 1e3:8.1f} us" + (f"  {fl / ms / 1e9:7.1f} TF/s" if fl else ""), file=sys.stderr)
        print(f"  sum of launches {tot:.2f} ms", file=sys.stderr)
    rows = rows[:top]
    out = []
    for key, (n, ms, fl) in rows:
        d = {"kernel": key, "launches_per_forward": n, "avg_us": round(ms / n * 1e3, 1), "total_ms": round(ms, 2)}
        if fl:
            d["achieved_tflops"] = round(fl / ms / 1e9, 1)
            d["frac"] = round(fl / ms / 1e9 / PEAK_F16_DENSE_TFLOPS, 3)
        out.append(d)
    return out


def fp16_stream_forward_ms(unet, ehs, T, h, w, n=3):
    """The same forward with the residual stream kept in plain fp16 (EW_RESIDUAL=fp16: the round-1 layout, 1.4e-3 rel-L2 against
    the fp32 oracle instead of 8.9e-4) -- reported next to the headline so that what parity at 1e-3 costs is measured on the
    same box in the same process.  Not part of the timed region, never the reported value."""
    keep = (unet.split_residual, unet.split_heads)
    unet.split_residual = unet.split_heads = False
    try:
        dev = ehs.device
        x_in = torch.zeros(2 * T * h * w, 64, dtype=torch.float16, device=dev)
        ids = torch.tensor([[6.0, 127.0, 0.02]] * 2, device=dev)
        e2 = torch.cat([torch.zeros_like(ehs), ehs], 0).to(torch.float16)
        unet.forward_nhwc(x_in, 1.0, e2, ids, 2, T, h, w)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            unet.forward_nhwc(x_in, 1.0, e2, ids, 2, T, h, w)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    finally:
        unet.split_residual, unet.split_heads = keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="clips timed per rank")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--denoise-steps", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp16-stream", action="store_true", help="skip the extra fp16-stream forwards (profiling runs: keeps the launch count at steps x denoise-steps + 1)")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="time ONE complete full-size CPU forward (minutes) instead of the bounded sample")
    ap.add_argument("--tiny", action="store_true", help="shrunken U-Net (plumbing check only; result flagged invalid)")
    ap.add_argument("--split", choices=["clip", "cfg"], default=os.environ.get("EW_BENCH_SPLIT", "clip"),
                    help="multi-GPU axis: clip = one independent clip per rank (weak scaling, default); cfg = ranks 2p, 2p+1 share clip p, "
                         "one CFG row each with one all_gather of eps per denoise step (strong scaling per clip; --gpus 1: both rows as "
                         "two B=1 forwards on the one rank)")
    args = ap.parse_args()

    # --gpus N from a plain `python bench.py`: re-exec as N ranks (one process per GPU) under torch.distributed.run
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        have = torch.cuda.device_count()
        if have < args.gpus and os.environ.get("EW_SHARE_GPU") != "1":
            raise SystemExit(f"--gpus {args.gpus} but only {have} device(s) answer")
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
    from evoworld_amd import distributed as D
    rank, world, local = D.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if os.environ.get("EW_SHARE_GPU") == "1":          # test hook: every rank on the visible devices round-robin (with EW_DIST_BACKEND=gloo)
        local = local % torch.cuda.device_count()
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: local device {local} does not exist ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from evoworld_amd import ops
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from evoworld_amd.scheduler import EulerDiscreteScheduler
    from evoworld_amd.unet import UNetSpatioTemporalConditionModel
    cfg = {}
    if args.tiny:
        cfg = dict(block_out_channels=(64, 128, 256, 256), addition_time_embed_dim=64,
                   projection_class_embeddings_input_dim=192, cross_attention_dim=1024, num_attention_heads=(1, 2, 4, 4))
    # rank 0 owns the weights (random-init here, a checkpoint in production); the other ranks receive the packed 3 GB over
    # RCCL / xGMI once (north_star: "RCCL broadcast"), then every rank runs its own clips with no collective in the loop
    if rank == 0:
        unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device=dev, **cfg)
    else:
        unet = UNetSpatioTemporalConditionModel.from_zeros(device=dev, **cfg)
    if world > 1 or os.environ.get("EW_FORCE_DIST") == "1":
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        unet.broadcast_weights(src=0)
        torch.cuda.synchronize()
        t_b = time.perf_counter() - t_b
        cs = unet.weights_checksum()
        lo, hi = D.max_over_ranks(-cs, dev), D.max_over_ranks(cs, dev)
        if -lo != hi:
            raise SystemExit(f"weight broadcast mismatch: checksum range [{-lo}, {hi}]")
        if rank == 0:
            print(f"[bench] weight broadcast to {world} rank(s): {t_b * 1e3:.1f} ms, checksum {cs:.6e} on every rank", file=sys.stderr)
    pipe = StableVideoDiffusionPipeline(unet=unet, scheduler=EulerDiscreteScheduler())
    T, h, w = args.frames, args.height // 8, args.width // 8
    n_clips = world                                  # clips in flight per "step" over the whole job
    if args.split == "cfg":
        if world > 1 and world % 2:
            raise SystemExit(f"--split cfg needs an even rank count (got {world})")
        grp = D.CfgGroup(rank, world, size=2 if world > 1 else 1)
        pipe.cfg_group = grp
        n_clips = grp.n_pairs
        latents, image_latents, ehs, plucker = synth_inputs(T, h, w, seed=10 + grp.pair, device=dev)   # both members: the same clip
    else:
        latents, image_latents, ehs, plucker = synth_inputs(T, h, w, seed=10 + rank, device=dev)
    dummy_image = torch.zeros(1, 3, args.height, args.width, device=dev)

    def one_clip():
        out = pipe(dummy_image, height=args.height, width=args.width, num_frames=T, num_inference_steps=args.denoise_steps,
                   latents=latents, output_type="latent", plucker_embedding=plucker, image_latents=image_latents,
                   image_embeddings=ehs, mask_mem=False).frames
        return D.gather_results(out)

    for _ in range(args.warmup):
        one_clip()
    # live HIP-event timing of the U-Net forward (the kernel chain the roofline is quoted on), on the launch stream
    fw_events = []
    orig_forward = unet.forward_nhwc

    def timed_forward(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig_forward(*a, **k)
        e.record()
        fw_events.append((s, e))
        return r
    unet.forward_nhwc = timed_forward

    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_clip()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    per_rank_s = D.gather_floats(dt, dev)           # every rank's own clock over the same barrier-bracketed region
    dt = D.max_over_ranks(dt, dev)
    pg_world, pg_backend = D.world_info()
    unet.forward_nhwc = orig_forward
    finite = all(bool(torch.isfinite(r).all()) for r in res)
    if args.split == "cfg" and world > 1:              # both members of a CFG pair hold the whole clip: they must agree bit for bit
        for q in range(0, world, 2):
            if not torch.equal(res[q], res[q + 1]):
                raise SystemExit(f"CFG pair {q // 2}: the two members returned different latents")
            if q and torch.equal(res[q], res[0]):      # pairs carry different clips (seed 10 + pair)
                raise SystemExit(f"CFG pair {q // 2} returned the latents of pair 0: pairs must work on different clips")
    from evoworld_amd import ops as _ops
    _ops.streamk_check()                      # (pipe.denoise already checked after every clip; this covers the hooked forward too)

    kernels = kernel_breakdown(unet, pipe, latents, image_latents, ehs, plucker, T, h, w) if rank == 0 else None
    fp16_ms = fp16_stream_forward_ms(unet, ehs, T, h, w) if rank == 0 and unet.split_residual and not args.no_fp16_stream else None

    if rank == 0:
        fw_ms = sum(s.elapsed_time(e) for s, e in fw_events) / max(1, len(fw_events))
        full = (not args.tiny) and (T, args.height, args.width, args.denoise_steps) == (25, 576, 1024, 25)
        algo = ALGO_TFLOP_PER_FORWARD / (2 if args.split == "cfg" else 1)     # cfg split: a forward is ONE CFG row (B = 1)
        ach = algo / (fw_ms / 1e3) if full else None
        tr = committed_traffic("split" if unet.split_residual else "fp16")
        clk = committed_clock()
        line = {
            "metric": "panoramic frames/sec per clip (576x1024x25f, 25 denoise steps)",
            "value": n_clips * T * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (args.split == "cfg" and world == 2) else "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            # what torch.distributed itself reports (N > 1: backend "nccl" = RCCL), and each rank's own time per step: a SCALE record can
            # show that RCCL saw N ranks and how far apart they finished; `value` uses the max
            "rccl_world": pg_world, "dist_backend": pg_backend, "per_rank_ms_per_step": [round(t / args.steps * 1e3, 3) for t in per_rank_s],
            "config": {"workload": f"configs[1]: single clip {args.height}x{args.width}x{T}f, {args.denoise_steps} EulerDiscrete "
                                   "steps, CFG batch 2, random-init SVD-Xtend U-Net (in_channels 18), "
                                   + ("one clip per GPU" if args.split == "clip" else
                                      f"CFG-pair split: {n_clips} clip(s) over {world} rank(s), one CFG row per rank, all_gather of eps per step"),
                       "parallelism": (f"dp{world}" if args.split == "clip" else f"cfg{min(2, world)} x dp{n_clips}"),
                       "unet_forward_ms": fw_ms, "residual_stream": "split fp16 + int8 (3 B/elt)" if unet.split_residual else "fp16",
                       "unet_forward_ms_fp16_stream": fp16_ms, "valid": bool(full and finite)},
            "roofline": {"bound": "mfma", "kernel": "U-Net denoise step (all launches of one forward, HIP events on the launch stream)",
                         "achieved": ach, "peak": PEAK_F16_DENSE_TFLOPS, "unit": "TFLOP/s",
                         "frac": (ach / PEAK_F16_DENSE_TFLOPS) if ach else None,
                         "traffic": (tr or {}).get("bytes_per_forward") if full and tr and not tr["stale"] else None, "traffic_detail": tr,
                         "algorithmic_tflop_per_launch": algo,
                         # diagnostic, not the headline: the peak scaled to the clock the long kernels actually sustain under the power limit
                         "effective_clock_ghz": clk["ghz"] if clk else None, "nominal_clock_ghz": NOMINAL_CLOCK_GHZ,
                         "frac_at_effective_clock": (ach / (PEAK_F16_DENSE_TFLOPS * clk["ghz"] / NOMINAL_CLOCK_GHZ)) if (ach and clk) else None,
                         "clock_detail": clk,
                         "kernels": kernels},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(frames=50 if args.cpu_baseline_full else 4)
        print(json.dumps(line), flush=True)
    D.barrier()
    D.shutdown()


if __name__ == "__main__":
    main()
