#!/usr/bin/env python
"""Reprojection-stage benchmark (BASELINE.json configs[2], the segment-1 -> segment-2 hand-off of the 3-clip loop): every
HIP kernel of the stage at full size, timed with HIP events on the launch stream, against its algorithmic HBM bytes.

Workload (segment 1): S = 49 perspective frames of 392x518 depth -> 9.95 M lifted points ->
percentile-50 confidence filter (~5 M points kept) -> splat into V = 24 views x 6 faces x 512^2 z-buffers -> resolve ->
cube->equirect 24 x 1000 x 2000 -> Pillow-exact resize to 24 x 576 x 1024 -> fp32 CHW in [-1,1]; plus the pano->perspective
gather that feeds the depth network (49 x 576x1024 -> 384x512) and the 8-bit quantisation of the generated frames.
Reference stage: evoworld/reprojection/reproject_vggt_open3d_utils.py:294-310,617-711, unified_loop_consistency.py:299-368.

Prints a table to stderr and ONE JSON line to stdout:
  {"stage": "reprojection", "total_ms": ..., "kernels": [{"kernel", "ms", "bytes", "GBps", "frac_of_8TBps"}, ...],
   "roofline": {...dominant kernel (the splat) against the 8 TB/s HBM roof...},
   "cpu_baseline": {...the numpy oracle (oracle/reproject_ref.py: lift, filter, splat, cube->equirect) timed on one host core on a bounded sample...}}
`bytes` = algorithmic bytes (each tensor the op must read / write, once), NOT measured traffic.
Usage: python bench_reproject.py [--iters 5] [--points-frames 49] [--no-cpu-baseline]
"""
import argparse
import json
import math
import sys

import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PEAK_HBM_GBPS = 8000.0


def cpu_baseline(depth, conf, images, extr, intr, w2c, lut, V, res, fx, gpu_ms):
    """The reference's CPU work for this stage restated by the numpy oracle (oracle/reproject_ref.py; Open3D's rasteriser and the vggt lift are
    absent from the image -- kind "port"), on ONE host core (numpy is single-threaded here), rows R1-R6 of SURVEY.md §8a: depth lift and
    percentile filter over ALL S frames, splat + resolve + cube->equirect for ONE of the V target views (the per-view work is identical),
    scaled x V.  ~20-30 s of CPU work.  A reported baseline, not a target."""
    from oracle import reproject_ref as R
    t = {}
    t0 = time.time()
    xyz = R.depth_unproject_ref(depth.cpu().numpy(), extr.cpu().numpy(), intr.cpu().numpy())
    t["lift"] = time.time() - t0
    t0 = time.time()
    v, c = R.confidence_filter_ref(xyz, conf.cpu().numpy(), R.extract_colors_ref(images.cpu().numpy()), 50.0)
    t["filter"] = time.time() - t0
    t0 = time.time()
    faces, _ = R.splat_ref(v, c, w2c[:1].cpu().numpy(), res, fx, fx, fx, fx, 0.1)
    t["splat_1view"] = time.time() - t0
    t0 = time.time()
    R.cube2equi_gather_ref(faces, lut.cpu().numpy())
    t["cube2equi_1view"] = time.time() - t0
    total = t["lift"] + t["filter"] + V * (t["splat_1view"] + t["cube2equi_1view"])
    return {"value": round(total * 1e3, 1), "unit": "ms per hand-off (R1-R6)", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"numpy oracle: lift {t['lift']:.2f} s + filter {t['filter']:.2f} s over all frames, splat {t['splat_1view']:.2f} s + cube->equirect "
                      f"{t['cube2equi_1view']:.2f} s for 1 of {V} views scaled x{V}",
            "gpu_ms_same_rows": round(gpu_ms, 3), "ratio": round(total * 1e3 / gpu_ms, 1)}


def timed(fn, iters, warmup=1):
    for _ in range(warmup):
        out = fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--points-frames", type=int, default=49, help="S: frames lifted to points (25 = segment 0, 49 = segment 1)")
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_reproject.py needs an MI355X (no CPU path)")
    from evoworld_amd import ops
    from evoworld_amd import reprojection as RP
    dev = "cuda"
    S, Hd, Wd, V, res, Hp, Wp = args.points_frames, 392, 518, args.views, 512, 1000, 2000
    g = torch.Generator().manual_seed(0)
    n = S * Hd * Wd
    # synthetic VGGT outputs: room-like depth, cameras on a small circle, fov-90 intrinsics
    v, u = torch.meshgrid(torch.linspace(-1, 1, Hd), torch.linspace(-1, 1, Wd), indexing="ij")
    depth = (3.0 / torch.maximum(u.abs(), v.abs() * 1.5).clamp_min(0.25))[None].repeat(S, 1, 1)
    depth = (depth * (1 + 0.05 * torch.rand(S, Hd, Wd, generator=g))).to(dev).contiguous()
    conf = torch.rand(S, Hd, Wd, generator=g).to(dev)
    images = torch.rand(S, 3, Hd, Wd, generator=g).to(dev)
    ang = torch.linspace(0, 2 * math.pi, S + 1)[:-1]
    c2w = torch.eye(4).repeat(S, 1, 1)
    c2w[:, 0, 3], c2w[:, 2, 3] = 0.5 * torch.cos(ang), 0.5 * torch.sin(ang)
    c2w[:, 0, 0], c2w[:, 0, 2], c2w[:, 2, 0], c2w[:, 2, 2] = torch.cos(ang), torch.sin(ang), -torch.sin(ang), torch.cos(ang)
    extr = torch.linalg.inv(c2w)[:, :3, :4].contiguous().to(dev)
    f = Wd / 2.0
    intr = torch.tensor([[f, 0, Wd / 2.0], [0, f, Hd / 2.0], [0, 0, 1]]).repeat(S, 1, 1).to(dev)
    tang = torch.linspace(0, 2 * math.pi, V + 1)[:-1]
    tc2w = torch.eye(4).repeat(V, 1, 1)
    tc2w[:, 0, 3], tc2w[:, 2, 3] = 0.8 * torch.cos(tang), 0.8 * torch.sin(tang)
    w2c = torch.tensor(RP.face_w2c(tc2w.numpy()), dtype=torch.float32).contiguous().to(dev)
    lut = RP.build_cube2equi_lut(Wp, Hp, res).to(dev)
    fx = res / 2.0
    rows = []

    def rec(name, ms, nbytes, note=""):
        rows.append({"kernel": name, "ms": round(ms, 4), "bytes": int(nbytes), "GBps": round(nbytes / ms / 1e6, 1),
                     "frac_of_8TBps": round(nbytes / ms / 1e6 / PEAK_HBM_GBPS, 4), "note": note})

    it = args.iters
    # R1 depth lift
    ms, xyz = timed(lambda: ops.depth_unproject(depth, extr, intr), it)
    rec("depth_unproject_kernel", ms, n * (4 + 12), "depth f32 in, xyz f32 out")
    # R2 select + compaction
    cf = conf.reshape(-1).contiguous()
    k = (n - 1) // 2
    ms, ab = timed(lambda: ops.select_kth(cf, k), it)
    rec("ew_select_kth_f32 (sel_hist x4 + sel_tail)", ms, n * 4 * 5, "5 streaming passes over conf")
    thr = float(RP.percentile_threshold(cf, 50.0))
    pts = xyz.reshape(-1, 3)
    ms, (vk, rgbx) = timed(lambda: ops.filter_compact(cf, thr, pts, images, Hd * Wd), it)
    m = vk.shape[0]
    rec("ew_filter_compact (count + scan + scatter)", ms, n * 4 * 2 + m * (12 + 12 + 12 + 4), "conf twice; kept xyz/img in, xyz/rgbx out")
    vk = vk.contiguous()
    # R4 splat: ew_splat_cubemap = zfill_kernel (the 0xFF.. init, inside the call since round 6) + splat_kernel
    zb_bytes = V * 6 * res * res * 8
    zbuf = torch.empty((V, 6, res, res), dtype=torch.int64, device=dev)
    lib = ops._lib.load()
    ms_fill, _ = timed(lambda: ops._lib.check(lib.ew_splat_cubemap(None, 0, ops._ptr(w2c), ops._ptr(zbuf), V, res, fx, fx, fx, fx, 0.1, ops._stream()), "zfill"), it)
    rec("zfill_kernel (ew_splat_cubemap, empty cloud)", ms_fill, zb_bytes, "0xFF.. init of the z-buffers")
    ms, _ = timed(lambda: ops._lib.check(lib.ew_splat_cubemap(ops._ptr(vk), m, ops._ptr(w2c), ops._ptr(zbuf), V, res, fx, fx, fx, fx, 0.1, ops._stream()), "splat"), it)
    frag = m * V
    ms_splat, splat_bytes = ms - ms_fill, m * 12 + zb_bytes
    rec("splat_kernel", ms_splat, splat_bytes, f"{m} points x {V} views = {frag / 1e6:.0f} M fragments; points read once + z-buffer touched once")
    faces4 = torch.empty(V, 6, res, res, 4, dtype=torch.uint8, device=dev)
    ms, _ = timed(lambda: ops._lib.check(lib.ew_splat_resolve(ops._ptr(zbuf), ops._ptr(rgbx), 4, ops._ptr(faces4), 4, V, res, ops._stream()), "resolve"), it)
    npix = V * 6 * res * res
    rec("resolve_kernel<4,4>", ms, npix * (8 + 4) + min(m, npix) * 4, "z-buffer in, RGBX faces out, colour gather")
    # R6 cube -> equirect
    ms, panos = timed(lambda: ops.cube2equi_gather(faces4, lut, Hp, Wp), it)
    rec("cube2equi_kernel<4>", ms, Hp * Wp * 6 + V * Hp * Wp * 3 + npix * 4, "LUT once, panoramas out, faces in (each texel ~once)")
    # R7 resize + convert
    ch = tuple(t.to(dev) for t in RP.resample_coeffs(Wp, 1024))
    cv = tuple(t.to(dev) for t in RP.resample_coeffs(Hp, 576))
    ms, small = timed(lambda: ops.resize_aa_u8(panos, ch, cv, 576, 1024), it)
    rec("resample_h_kernel + resample_v_kernel", ms, V * 3 * (Hp * Wp + 2 * Hp * 1024 + 576 * 1024), "src in, tmp out+in, dst out")
    ms, mem = timed(lambda: ops.u8_hwc_to_f32_chw(small), it)
    rec("u8_hwc_to_f32_chw_kernel", ms, V * 576 * 1024 * (3 + 12))
    # R0 pano -> perspective for the depth network, and the 8-bit quantisation of generated frames
    frames = (torch.rand(S, 3, 576, 1024, generator=g) * 2 - 1).to(dev)
    ms, fu8 = timed(lambda: ops.f32_chw_to_u8_hwc(frames), it)
    rec("f32_chw_to_u8_hwc_kernel", ms, S * 576 * 1024 * (12 + 3))
    e2p = RP.Equi2Pers(height=384, width=512, fov_x=90.0, mode="bilinear")
    rots = [{"pitch": 0, "roll": 0, "yaw": 0.1 * i} for i in range(S)]
    ms, pers = timed(lambda: e2p.batch(fu8, rots), it)
    rec("equi2pers_kernel", ms, S * 384 * 512 * 3 * (1 + 4), "out + 4 bilinear taps (the panorama region seen is ~1/6 of the sphere)")
    total = sum(r["ms"] for r in rows)
    print(f"{'kernel':52s} {'ms':>9s} {'MB':>9s} {'GB/s':>9s} {'of 8 TB/s':>10s}", file=sys.stderr)
    for r in rows:
        print(f"{r['kernel']:52s} {r['ms']:9.3f} {r['bytes'] / 1e6:9.1f} {r['GBps']:9.1f} {r['frac_of_8TBps']:10.3f}   {r['note']}", file=sys.stderr)
    print(f"{'total':52s} {total:9.3f}", file=sys.stderr)
    line = {"stage": "reprojection", "workload": f"configs[2] hand-off: S={S} frames of {Hd}x{Wd} depth ({n} points, {m} kept), "
            f"V={V} views x 6 x {res}^2, {Hp}x{Wp} panoramas -> 576x1024", "total_ms": round(total, 3), "peak_GBps": PEAK_HBM_GBPS, "kernels": rows,
            # dominant kernel of the stage: the splat -- algorithmic bytes (points once + every z-buffer word once) over its HIP-event time.  It is bound
            # by the rate of 64-bit atomics (fragments/s), not by bytes: the fraction of the HBM roof is reported as the contract asks, the fragment
            # rate next to it says what actually limits it
            "roofline": {"bound": "hbm", "kernel": "splat_kernel", "achieved": round(splat_bytes / ms_splat / 1e6, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(splat_bytes / ms_splat / 1e6 / PEAK_HBM_GBPS, 4), "traffic": None,
                         "fragments_per_s": round(frag / ms_splat * 1e3), "limit": "memory-side 64-bit atomicMin rate"}}
    if not args.no_cpu_baseline:
        gpu_rows = ("depth_unproject_kernel", "ew_select_kth_f32", "ew_filter_compact", "zfill_kernel", "splat_kernel", "resolve_kernel", "cube2equi_kernel")
        gpu_ms = sum(r["ms"] for r in rows if r["kernel"].startswith(gpu_rows))
        line["cpu_baseline"] = cpu_baseline(depth, conf, images, extr, intr, w2c, lut, V, res, fx, gpu_ms)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
