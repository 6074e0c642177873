#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden-vector generator (runs ONLY in the build container).

Imports the reference's *importable* geometry code from /root/reference (read-only) with
sys.modules stubs for the third-party packages that are absent here, runs it on seeded
inputs and writes the input/output pairs as data fixtures to tests/golden/*.npz.

Nothing from /root/reference is copied: the fixtures are inputs and expected outputs only.
The reference cannot travel to the GPU box, so tests consume the committed fixtures.

Reference functions exercised (file:line in /root/reference):
  utils/plucker_embedding.py:56   equirectangular_to_ray
  utils/plucker_embedding.py:221  ray_c2w_to_plucker
  dataset/CameraTrajDataset.py:643 xyz_euler_to_three_by_four_matrix_batch
  utils/geometry.py:5             xyz_euler_to_four_by_four_matrix_batch
  evoworld/reprojection/reproject_vggt_open3d_utils.py:542  CubemapRenderer.cube_to_equirectangular_cuda
  evoworld/reprojection/reproject_vggt_open3d_utils.py:1176 align_first_and_last_points
  evoworld/reprojection/reproject_vggt_open3d_utils.py:1126 rotation_from_vectors
  evoworld/reprojection/reproject_vggt_open3d_utils.py:286,294 _extract_colors/_apply_confidence_filter
  evoworld/reprojection/reproject_vggt_open3d_utils.py:472  SceneBuilder.align_extrinsics (numpy part)
  evoworld/reprojection/pano_to_pers_utils.py:5  calculate_segment_indices
  evoworld/inference/navigator_evoworld.py:303   Navigator.split_curve_into_segments
  evoworld/pipeline/pipeline_evoworld.py:746     _resize_with_antialiasing

Usage:  python oracle/make_goldens.py   (from the repo root)
"""
import hashlib
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_reference():
    scratch = tempfile.mkdtemp(prefix="ew_gold_")
    open(os.path.join(scratch, "skyseg.onnx"), "wb").close()  # blocks the import-time download
    os.chdir(scratch)
    sys.path.insert(0, REF)
    from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection  # noqa: F401  (before the torchvision stub)
    for name in ["cv2", "onnxruntime", "open3d", "trimesh", "equilib", "imageio", "torchvision",
                 "torchvision.transforms", "torchvision.transforms.functional", "matplotlib",
                 "matplotlib.pyplot", "requests", "PIL.ImageOps"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    # diffusers stubs: plain classes so that the pipeline module can subclass them
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _C:  # generic base
        def __init__(self, *a, **k):
            pass

    mod("diffusers")
    mod("diffusers.image_processor", PipelineImageInput=object)
    mod("diffusers.models", AutoencoderKLTemporalDecoder=_C, UNetSpatioTemporalConditionModel=_C)
    mod("diffusers.schedulers", EulerDiscreteScheduler=_C)
    lg = types.SimpleNamespace(get_logger=lambda *_a, **_k: MagicMock())
    mod("diffusers.utils", BaseOutput=_C, logging=lg,
        replace_example_docstring=lambda *_a, **_k: (lambda f: f))
    mod("diffusers.utils.torch_utils", is_compiled_module=lambda m: False, randn_tensor=None)
    mod("diffusers.video_processor", VideoProcessor=_C)
    mod("diffusers.pipelines")
    mod("diffusers.configuration_utils", ConfigMixin=type("ConfigMixin",(),{}), register_to_config=lambda f: f)
    mod("diffusers.loaders", UNet2DConditionLoadersMixin=type("LoadersMixin",(),{}))
    mod("diffusers.models.attention_processor", CROSS_ATTENTION_PROCESSORS=(), AttentionProcessor=_C, AttnProcessor=_C)
    mod("diffusers.models.embeddings", TimestepEmbedding=_C, Timesteps=_C)
    mod("diffusers.models.modeling_utils", ModelMixin=type("ModelMixin",(torch.nn.Module,),{}))
    mod("diffusers.models.unets")
    mod("diffusers.models.unets.unet_3d_blocks", UNetMidBlockSpatioTemporal=_C, get_down_block=None, get_up_block=None)
    mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=_C)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    out = os.path.abspath(OUT)
    _import_reference()
    from utils.plucker_embedding import equirectangular_to_ray, ray_c2w_to_plucker
    from utils.geometry import xyz_euler_to_four_by_four_matrix_batch
    from dataset.CameraTrajDataset import xyz_euler_to_three_by_four_matrix_batch
    import evoworld.reprojection.reproject_vggt_open3d_utils as R
    from evoworld.reprojection.pano_to_pers_utils import calculate_segment_indices

    # ---------------- K1: poses -> c2w -> Plücker (example/case_000, frames 102..126) -------------
    rows = open(os.path.join(REF, "example/case_000/camera_poses.txt")).read().strip().split("\n")[1:]
    poses_unity = np.array([[float(v) for v in r.split(",")[1:]] for r in rows], dtype=np.float64)  # [126,6]
    flip = np.array([1, -1, 1, -1, 1, -1], dtype=np.float64)  # utils/constant.py:3
    gold = {"poses_unity": poses_unity}
    rays = equirectangular_to_ray(72, 128)
    gold["rays_72x128"] = rays.astype(np.float32)
    for tag, ps in (("ps01", 0.1), ("ps10", 1.0)):
        p = poses_unity[101:126] * flip
        p[:, :3] *= ps  # dataset/CameraTrajDataset.py:348 (pos_scale on xyz only)
        cam = torch.tensor(p, dtype=torch.float32)
        c2w = xyz_euler_to_three_by_four_matrix_batch(cam, relative=True)
        pl = ray_c2w_to_plucker(torch.tensor(rays).float(), c2w)
        gold[f"cam_{tag}"] = cam.numpy()
        gold[f"c2w_{tag}"] = c2w.numpy()
        gold[f"plucker_{tag}_f0_12_24"] = pl[[0, 12, 24]].numpy()
        gold[f"plucker_{tag}_sum"] = np.array([pl.double().sum().item(), pl.double().abs().sum().item()])
        gold[f"plucker_{tag}_rowsum"] = pl.double().sum(dim=(2, 3)).numpy()  # [25,6]
    # random poses, absolute + relative, 3x4 and 4x4
    g = torch.Generator().manual_seed(7)
    rp = torch.cat([torch.randn(9, 3, generator=g) * 3, (torch.rand(9, 3, generator=g) - 0.5) * 360], dim=1)
    gold["rand_poses"] = rp.numpy()
    gold["rand_c2w_abs"] = xyz_euler_to_three_by_four_matrix_batch(rp, relative=False).numpy()
    gold["rand_c2w_rel"] = xyz_euler_to_three_by_four_matrix_batch(rp, relative=True).numpy()
    gold["rand_c2w4_rel"] = xyz_euler_to_four_by_four_matrix_batch(rp, relative=True).numpy()
    gold["rand_c2w4_abs"] = xyz_euler_to_four_by_four_matrix_batch(rp, relative=False).numpy()
    rays_s = equirectangular_to_ray(8, 16)
    gold["rays_8x16"] = rays_s.astype(np.float32)
    gold["rand_plucker_8x16"] = ray_c2w_to_plucker(
        torch.tensor(rays_s).float(), torch.tensor(gold["rand_c2w_rel"])).numpy()
    np.savez_compressed(os.path.join(out, "plucker.npz"), **gold)

    # ---------------- K2: cube -> equirect index LUT ---------------------------------------------
    cr = R.CubemapRenderer()
    order = ["right", "left", "bottom", "top", "front", "back"]

    def lut_for(W, H, res):
        # index-coded faces: channel0 = face id, channel1/2 = v,u split into bytes
        luts = []
        for code in range(3):  # three passes: face id, v, u (values up to res-1 may exceed 255)
            faces = {}
            for fi, f in enumerate(order):
                vv, uu = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
                src = [np.full((res, res), fi), vv, uu][code]
                arr = np.stack([src & 255, (src >> 8) & 255, np.zeros_like(src)], 0).astype(np.uint8)
                faces[f] = torch.from_numpy(arr[None])
            pano = cr.cube_to_equirectangular_cuda(faces, W, H, device="cpu")[0]
            luts.append(pano[..., 0].astype(np.int16) + (pano[..., 1].astype(np.int16) << 8))
        return np.stack(luts, -1)  # [H,W,3] = face, v, u

    lut_full = lut_for(2000, 1000, 512)
    lut_small = lut_for(64, 32, 16)
    lut_mid = lut_for(256, 128, 64)
    hist = np.bincount(lut_full[..., 0].reshape(-1), minlength=6)
    probes = [(0, 0), (500, 1000), (500, 0), (500, 1500), (250, 500), (999, 1999), (100, 700),
              (499, 250), (500, 250), (499, 1750), (750, 1250), (1, 1), (998, 3)]
    pv = np.array([[r, c, *lut_full[r, c]] for r, c in probes], dtype=np.int32)
    # full LUT stored as row-delta-coded int16 (compresses well)
    np.savez_compressed(os.path.join(out, "cube2equi_lut.npz"),
                        lut_2000x1000x512=lut_full, lut_64x32x16=lut_small, lut_256x128x64=lut_mid,
                        hist_full=hist, probes_full=pv,
                        sha256_full=np.frombuffer(bytes.fromhex(sha(lut_full)), dtype=np.uint8))
    print("LUT sha256", sha(lut_full), "hist", hist.tolist())
    # a pixel-level gather golden on random faces (small config)
    g = torch.Generator().manual_seed(11)
    faces = {f: torch.randint(0, 256, (3, 3, 16, 16), generator=g, dtype=torch.uint8) for f in order}
    pano = cr.cube_to_equirectangular_cuda(faces, 64, 32, device="cpu")
    np.savez_compressed(os.path.join(out, "cube2equi_gather.npz"),
                        faces=np.stack([faces[f].numpy() for f in order], 1),  # [B,6,3,res,res] order=right,left,bottom,top,front,back
                        pano=pano)

    # ---------------- K3: 2-point similarity alignment -------------------------------------------
    cases_A, cases_B, outs = [], [], []
    rng = np.random.default_rng(3)
    raw = [
        (np.array([[0, 0, 0], [1, 0, 0], [2, 0, 1.0]]), np.array([[1, 1, 1], [1, 2, 1], [1, 3, 3.0]])),
        (rng.normal(size=(5, 3)), rng.normal(size=(5, 3))),
        (np.array([[0, 0, 0], [0, 0, 2.0]]), np.array([[1, 0, 0], [1, 0, 5.0]])),      # parallel
        (np.array([[0, 0, 0], [0, 0, 2.0]]), np.array([[1, 0, 0], [1, 0, -5.0]])),     # antiparallel
        (np.array([[0, 0, 0], [2.0, 0, 0]]), np.array([[0, 1, 0], [-3.0, 1, 0]])),     # antiparallel, x axis
        (np.array([[1, 2, 3.0], [1, 2, 3.0]]), np.array([[0, 0, 0], [1.0, 1, 1]])),    # degenerate A
        (rng.normal(size=(25, 3)) * 4, rng.normal(size=(25, 3)) * 0.3),
    ]
    al = {}
    for i, (A, B) in enumerate(raw):
        s, Rm, t = R.align_first_and_last_points(A, B)
        al[f"A{i}"], al[f"B{i}"] = A, B
        al[f"s{i}"], al[f"R{i}"], al[f"t{i}"] = np.float64(s), np.asarray(Rm, np.float64), np.asarray(t, np.float64)
    al["n"] = np.int64(len(raw))
    # align_extrinsics numpy part restated by calling the pieces (the method itself calls .cuda())
    S = 49
    gt = xyz_euler_to_four_by_four_matrix_batch(
        torch.tensor(poses_unity[:126] * flip * np.array([0.1, 0.1, 0.1, 1, 1, 1]), dtype=torch.float32),
        relative=True).numpy()
    ang = 0.3
    Rw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    sc, tw = 1.7, np.array([0.5, -0.2, 0.9])
    c2w_v = np.repeat(np.eye(4)[None], S, 0)
    c2w_v[:, :3, :3] = Rw @ gt[:S, :3, :3]
    c2w_v[:, :3, 3] = (sc * (Rw @ gt[:S, :3, 3].T)).T + tw
    extr = np.linalg.inv(c2w_v)[:, :3, :4]  # world->cam, what VGGT returns
    seg = 1
    start = (seg + 1) * 24 + 1
    inv = np.stack([np.linalg.inv(np.vstack([e, [0, 0, 0, 1]])) for e in extr])
    s, Rm, t = R.align_first_and_last_points(gt[:start][:, :3, 3], inv[:, :3, 3])
    T = np.eye(4); T[:3, :3] = s * Rm; T[:3, 3] = t
    tgt = np.einsum("ij,bjk->bik", T, gt[start:start + 24])
    al["ax_gt"], al["ax_extr"], al["ax_seg"], al["ax_target"] = gt, extr, np.int64(seg), tgt
    np.savez_compressed(os.path.join(out, "align.npz"), **al)

    # ---------------- R2: colour extraction + percentile filter -----------------------------------
    pp = R.PointCloudProcessor()
    g = torch.Generator().manual_seed(5)
    imgs = torch.rand(3, 3, 14, 18, generator=g).numpy()
    pts = torch.randn(3, 14, 18, 3, generator=g).numpy()
    conf = torch.rand(3, 14, 18, generator=g).numpy()
    conf[0, :3] = conf[0, 3]  # ties
    cols = pp._extract_colors(imgs)
    v50, c50 = pp._apply_confidence_filter(pts, conf, cols, 50.0)
    v30, c30 = pp._apply_confidence_filter(pts, conf, cols, 30.0)
    v0, c0 = pp._apply_confidence_filter(pts, conf, cols, 0.0)
    np.savez_compressed(os.path.join(out, "filter.npz"), images=imgs, points=pts, conf=conf, colors=cols,
                        v50=v50, c50=c50, v30=v30, c30=c30, v0=v0, c0=c0,
                        scale50=np.float64(pp._calculate_scene_scale(v50)))

    # ---------------- K4: segment index math ------------------------------------------------------
    segidx = np.array([calculate_segment_indices(i) for i in range(5)], dtype=np.int64)
    import evoworld.inference.navigator_evoworld as NV
    split = {}
    for L in (10, 25, 26, 49, 73, 126):
        segs = NV.Navigator.split_curve_into_segments(None, list(range(L)))
        split[f"L{L}"] = np.array([[s_[0], s_[-1] + 1] for s_ in segs], dtype=np.int64)
    np.savez_compressed(os.path.join(out, "segments.npz"), calculate_segment_indices=segidx, **split)

    # ---------------- N2 helper: antialias resize (kept for the CLIP "next" row) -------------------
    import evoworld.pipeline.pipeline_evoworld as P
    g = torch.Generator().manual_seed(13)
    x = torch.rand(1, 3, 72, 128, generator=g) * 2 - 1
    y = P._resize_with_antialiasing(x, (28, 28))
    np.savez_compressed(os.path.join(out, "resize_antialias.npz"), x=x.numpy(), y=y.numpy())
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    main()
