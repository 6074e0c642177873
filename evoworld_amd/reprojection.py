"""Reprojection stage: 3D memory (VGGT depth/pose predictions) -> 24 target panoramas, on the GPU.

Call surface of evoworld/reprojection/reproject_vggt_open3d_utils.py (predictions_to_target_view :1216,
PointCloudProcessor :165, SceneBuilder :345, CubemapRenderer :522, align_first_and_last_points :1176,
rotation_from_vectors :1126), pano_to_pers_utils.py:5 (calculate_segment_indices) and the pano->perspective step
of unified_loop_consistency.py:299-334.  The reference round-trips through numpy, Open3D's GL context and PNG
files; here the point cloud, z-buffers, cube faces and panoramas stay in HBM:
    depth lift (ew_depth_unproject) -> percentile filter (order statistics on device) -> 2-point similarity alignment
    (host, float64, 3 points matter) -> point splat into 24x6 z-buffers (ew_splat_cubemap, 64-bit atomicMin)
    -> cube->equirect gather through the integer LUT (ew_cube2equi_gather).
The Open3D / pyequilib / vggt arithmetic is third-party and unpinned; the LUT index path, alignment,
filter and segment math are pinned bit-exactly by goldens generated from the reference (tests/golden).
"""
import math
import os

import numpy as np
import torch

from . import ops

FACE_ORDER = ("right", "left", "bottom", "top", "front", "back")    # face ids of the cube->equirect LUT (:583-590)
CUBEMAP_TRANSFORMS = {                                               # :29-36
    "front": np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64),
    "right": np.array([[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
    "back": np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64),
    "left": np.array([[0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
    "top": np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
    "bottom": np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
}
CUBEMAP = CUBEMAP_TRANSFORMS
Z_NEAR = 0.1   # Open3D's camera near plane floor (unpinned)

_lut_cache = {}


def build_cube2equi_lut(width, height, res):
    """int16 [H,W,3] = (face, v_px, u_px): the integer index path of cube_to_equirectangular_cuda (:542-607).
    Evaluated ONCE per (W,H,res) with torch CPU float32 ops in the reference's order, so it is bit-identical to what the
    reference computes on this software stack (golden: sha256 e45a27dc... for (2000,1000,512))."""
    key = (width, height, res)
    if key in _lut_cache:
        return _lut_cache[key]
    col = torch.linspace(0, width - 1, width)
    row = torch.linspace(0, height - 1, height)
    rr, cc = torch.meshgrid(row, col, indexing="ij")
    lon = (-cc / width) * 2 * torch.pi - torch.pi + torch.pi / 2
    lat = (rr / height) * torch.pi - torch.pi / 2
    X, Y, Z = torch.cos(lat) * torch.cos(lon), torch.sin(lat), torch.cos(lat) * torch.sin(lon)
    aX, aY, aZ = X.abs(), Y.abs(), Z.abs()
    dom = {"right": (aX >= aY) & (aX >= aZ) & (X > 0), "left": (aX >= aY) & (aX >= aZ) & (X < 0),
           "bottom": (aY >= aX) & (aY >= aZ) & (Y > 0), "top": (aY >= aX) & (aY >= aZ) & (Y < 0),
           "front": (aZ >= aX) & (aZ >= aY) & (Z > 0), "back": (aZ >= aX) & (aZ >= aY) & (Z < 0)}
    uv = {"right": (-Z / aX, -Y / aX), "left": (Z / aX, -Y / aX), "bottom": (-X / aY, -Z / aY), "top": (-X / aY, Z / aY),
          "front": (X / aZ, -Y / aZ), "back": (-X / aZ, -Y / aZ)}
    face = torch.zeros((height, width), dtype=torch.int64)
    u, v = torch.zeros_like(X), torch.zeros_like(Y)
    for fi, f in enumerate(FACE_ORDER):          # dict order of the reference: ties go to the LAST face written
        m = dom[f]
        face[m] = fi
        u[m], v[m] = uv[f][0][m], uv[f][1][m]
    u_px = (((u + 1) / 2) * (res - 1)).long()
    v_px = ((1 - (v + 1) / 2) * (res - 1)).long()
    lut = torch.stack([face, v_px, u_px], -1).to(torch.int16).contiguous()
    _lut_cache[key] = lut
    return lut


# ------------------------------------------------------------------ alignment (host, float64; :1126-1213)
def rotation_from_vectors(u, v):
    nu, nv = np.linalg.norm(u), np.linalg.norm(v)
    if nu < 1e-15 or nv < 1e-15:
        return np.eye(3)
    uh, vh = u / nu, v / nv
    dot = np.clip(np.dot(uh, vh), -1.0, 1.0)
    if np.isclose(dot, 1.0):
        return np.eye(3)
    if np.isclose(dot, -1.0):
        tmp = np.array([1.0, 0.0, 0.0])
        if np.abs(np.dot(uh, tmp)) > 0.9:
            tmp = np.array([0.0, 1.0, 0.0])
        w = np.cross(uh, tmp)
        w /= np.linalg.norm(w)
        return np.eye(3) - 2.0 * np.outer(w, w)
    angle = np.arccos(dot)
    w = np.cross(uh, vh)
    wh = w / np.linalg.norm(w)
    K = np.array([[0, -wh[2], wh[1]], [wh[2], 0, -wh[0]], [-wh[1], wh[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def align_first_and_last_points(A, B):
    """(s, R, t) with B0 = s R A0 + t and B_last = s R A_last + t."""
    A0, A1, B0, B1 = A[0], A[-1], B[0], B[-1]
    vA, vB = A1 - A0, B1 - B0
    lenA, lenB = np.linalg.norm(vA), np.linalg.norm(vB)
    if lenA < 1e-15:
        return 1.0, np.eye(3), B0 - np.eye(3) @ A0
    s = lenB / lenA
    R = rotation_from_vectors(vA, vB)
    return s, R, B0 - s * R @ A0


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class SceneBuilder:
    def align_extrinsics(self, camera_pose, predictions_extrinsic, num_target_view, outdir, only_render_last_24_frame):
        """Target camera-to-world matrices in the VGGT frame (:472-519)."""
        cam = _np(predictions_extrinsic)
        n = len(cam)
        E = np.zeros((n, 4, 4))
        E[:, :3, :4] = cam
        E[:, 3, 3] = 1
        E_inv = np.stack([np.linalg.inv(e) for e in E])
        try:
            segment_id = int(str(outdir).rstrip("/").split("_")[-1])
        except Exception:
            segment_id = 1
        start = (segment_id + 1) * num_target_view + 1 if not only_render_last_24_frame else -num_target_view
        pose = _np(camera_pose)
        gt = pose[:start]
        target_gt = pose[start:start + num_target_view] if not only_render_last_24_frame else pose[start:]
        s, R, t = align_first_and_last_points(gt[:, :3, 3], E_inv[:, :3, 3])
        T = np.eye(4)
        T[:3, :3] = s * R
        T[:3, 3] = t
        return np.einsum("ij, bjk -> bik", T, target_gt)


def face_w2c(target_c2w):
    """float32 [V,6,3,4] world->camera per (view, face): inv(c2w @ T_face [@ Rz(180) for top/bottom]) (:617-666)."""
    Fz = np.eye(4)
    Fz[:3, :3] = np.diag([-1.0, -1.0, 1.0])
    out = np.zeros((len(target_c2w), 6, 3, 4), dtype=np.float32)
    for v, c2w in enumerate(np.asarray(target_c2w, dtype=np.float64)):
        for fi, f in enumerate(FACE_ORDER):
            pose = c2w @ CUBEMAP_TRANSFORMS[f]
            if f in ("top", "bottom"):
                pose = pose @ Fz
            out[v, fi] = np.linalg.inv(pose)[:3, :4].astype(np.float32)
    return out


# ------------------------------------------------------------------ point cloud filter (:165-337)
def _dev(x, device, dtype=None):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=device, dtype=dtype) if dtype is not None else t.to(device)


def percentile_rank(n, q):
    """(lo, hi, t): np.percentile's 'linear' virtual index (n-1)*q/100 in numpy >= 2 float32 arithmetic -> the two
    neighbouring order statistics and the interpolation weight."""
    qq = np.true_divide(q, np.float32(100))
    k = (n - 1) * qq
    lo = int(np.floor(k))
    hi = min(lo + 1, n - 1)
    return lo, hi, np.asanyarray(k - lo, dtype=np.float32)


def percentile_lerp(a, b, t):
    """numpy's _lerp in float32: a + (b-a)*t, with the t >= 0.5 form b - (b-a)*(1-t)."""
    a, b = np.float32(a), np.float32(b)
    d = b - a
    r = a + d * t
    if t >= 0.5:
        r = b - d * (1 - t)
    return np.float32(r)


def percentile_threshold(conf_flat, q):
    """np.percentile(conf, q) (method 'linear') for a float32 device tensor: the two neighbouring order statistics are
    selected on the device (ew_select_kth_f32: radix select, exact), the interpolation replays numpy's own float32
    arithmetic (`percentile_rank` / `percentile_lerp`, verified equal to np.percentile on this stack by
    tests/test_cpu_reprojection_host.py; the device selection by tests/test_gpu_reprojection.py)."""
    n = conf_flat.numel()
    lo, hi, t = percentile_rank(n, q)
    ab = ops.select_kth(conf_flat.contiguous(), lo).cpu().numpy()      # x_(lo), x_(lo+1)
    return percentile_lerp(ab[0], ab[1] if hi > lo else ab[0], t)


class PointCloudProcessor:
    def __init__(self, device="cuda"):
        self.device = torch.device(device)

    def filter_predictions(self, predictions, conf_thres=50.0, filter_by_frames="all", mask_black_bg=False,
                           mask_white_bg=False, mask_sky=False, target_dir=None, image_subdir=None,
                           prediction_mode="Predicted Pointmap", only_render_last_24_frame=False):
        """-> (vertices [N,3] f32 cuda, colors [N,3] u8 cuda, scene_scale float).  Sky masking (an onnx model in the
        reference, :51-163) is out of scope: mask_sky must be False, as on the inference path."""
        if mask_sky:
            raise NotImplementedError("mask_sky=True needs the skyseg onnx model (out of scope; the pipeline passes False)")
        dev = self.device
        if "Pointmap" in prediction_mode and "world_points" in predictions:
            pts, conf = predictions["world_points"], predictions.get("world_points_conf")
        else:
            pts, conf = predictions["world_points_from_depth"], predictions.get("depth_conf")
        pts = _dev(pts, dev, torch.float32)
        conf = torch.ones(pts.shape[:-1], device=dev) if conf is None else _dev(conf, dev, torch.float32)
        images = _dev(predictions["images"], dev, torch.float32)
        if filter_by_frames not in ("all", "All"):
            try:
                idx = int(str(filter_by_frames).split(":")[0])
                pts, conf, images = pts[idx:idx + 1], conf[idx:idx + 1], images[idx:idx + 1]
            except (ValueError, IndexError):
                pass
        cf = conf.reshape(-1).contiguous()
        thr = 0.0 if conf_thres == 0.0 else float(percentile_threshold(cf, conf_thres))
        # colours = (images NHWC * 255) truncated to uint8 (:286-292), fused into the order-preserving compaction
        nchw = images.ndim == 4 and images.shape[1] == 3
        hw = images.shape[2] * images.shape[3] if nchw else 0
        v, rgbx = ops.filter_compact(cf, thr, pts.reshape(-1, 3).contiguous(), images.contiguous(), hw)
        c = rgbx[:, :3]                                                          # view of the RGBX words
        if v.shape[0] == 0:
            v = torch.tensor([[1.0, 0.0, 0.0]], device=dev)
            c = torch.tensor([[255, 255, 255]], dtype=torch.uint8, device=dev)
        if mask_black_bg or mask_white_bg:
            m = torch.ones(len(v), dtype=torch.bool, device=dev)
            if mask_black_bg:
                m &= c.sum(dim=1, dtype=torch.int32) >= 16
            if mask_white_bg:
                m &= ~((c[:, 0] > 240) & (c[:, 1] > 240) & (c[:, 2] > 240))
            if m.any():
                v, c = v[m].contiguous(), c[m].contiguous()
        v = v.contiguous()
        lo = torch.quantile(v[:: max(1, len(v) // 2_000_000)].double(), 0.05, dim=0)
        hi = torch.quantile(v[:: max(1, len(v) // 2_000_000)].double(), 0.95, dim=0)
        return v, c, float((hi - lo).norm())


class CubemapRenderer:
    """24 target views x 6 faces point splat + cube->equirect, all on the device (:522-711)."""

    def __init__(self, face_res=512, z_near=Z_NEAR):
        self.face_res, self.z_near = face_res, z_near

    def render_cubemaps(self, vertices, colors, target_extrinsic, face_channels=3):
        res = self.face_res
        f = res / (2 * math.tan(math.radians(90.0) / 2))                         # fx = fy = 256, cx = cy = 256 (:624-627)
        w2c = torch.from_numpy(face_w2c(target_extrinsic)).to(vertices.device)
        faces, _ = ops.splat_cubemap(vertices, colors, w2c, res, f, f, res / 2, res / 2, self.z_near, face_channels)
        return faces                                                             # uint8 [V,6,res,res,3|4], FACE_ORDER

    def render_cubemaps_to_panoramas(self, vertices, colors, target_extrinsic, num_target_view=24, outdir=None,
                                     width=2000, height=1000):
        faces = self.render_cubemaps(vertices, colors, target_extrinsic, face_channels=4)   # RGBX words: aligned gathers
        lut = build_cube2equi_lut(width, height, self.face_res).to(vertices.device)
        panos = ops.cube2equi_gather(faces, lut, height, width)                   # uint8 [V,H,W,3] on the device
        if outdir:
            os.makedirs(outdir, exist_ok=True)
            from PIL import Image
            for i, p in enumerate(panos.cpu().numpy()):
                Image.fromarray(p).save(os.path.join(outdir, f"{i:02}.png"))      # RGB on disk, as cv2.imwrite(BGR(pano))
        return panos


def predictions_to_target_view(predictions, camera_pose, conf_thres=50.0, filter_by_frames="all", point_processor=None,
                               scene_builder=None, cubemap_renderer=None, mask_black_bg=False, mask_white_bg=False,
                               show_cam=True, mask_sky=False, target_dir=None, image_subdir=None,
                               prediction_mode="Predicted Pointmap", num_target_view=24, outdir="demo_pyrender_render",
                               only_render_last_24_frame=False, return_device_tensor=False, save_png=True):
    """VGGT predictions -> 24 reprojected panoramas uint8 [24,1000,2000,3] (:1216-1282).  `outdir`'s `_{segment}` suffix
    selects the target poses, as in the reference (:487-492)."""
    if not isinstance(predictions, dict):
        raise ValueError("predictions must be a dictionary")
    pp = point_processor or PointCloudProcessor()
    sb = scene_builder or SceneBuilder()
    cr = cubemap_renderer or CubemapRenderer()
    if "world_points_from_depth" not in predictions and "world_points" not in predictions:
        predictions = dict(predictions)
        predictions["world_points_from_depth"] = depth_to_world_points(predictions["depth"], predictions["extrinsic"],
                                                                       predictions["intrinsic"])
    v, c, _scale = pp.filter_predictions(predictions, conf_thres, filter_by_frames, mask_black_bg, mask_white_bg, mask_sky,
                                         target_dir, image_subdir, prediction_mode, only_render_last_24_frame)
    target = sb.align_extrinsics(camera_pose, predictions["extrinsic"], num_target_view, outdir, only_render_last_24_frame)
    panos = cr.render_cubemaps_to_panoramas(v, c, target, num_target_view, outdir if save_png else None)
    return panos if return_device_tensor else panos.cpu().numpy()


def depth_to_world_points(depth, extrinsic, intrinsic, device="cuda"):
    """vggt unproject_depth_map_to_point_map (unified_loop_consistency.py:365-367): [S,H,W(,1)] -> [S,H,W,3] on device."""
    d = _dev(depth, device, torch.float32)
    if d.ndim == 4:
        d = d[..., 0]
    return ops.depth_unproject(d.contiguous(), _dev(extrinsic, device, torch.float32).contiguous(),
                               _dev(intrinsic, device, torch.float32).contiguous())


# ------------------------------------------------------------------ pano -> perspective (unified_loop_consistency.py:299-334)
def calculate_segment_indices(segment_id):
    """(start_idx, end_idx, look_at_idx)  (evoworld/reprojection/pano_to_pers_utils.py:5-14)."""
    look_at_idx = (segment_id + 1) * 24 + 24
    start_idx = segment_id * 24 + 1
    if segment_id == 0:
        start_idx -= 1
    return start_idx, start_idx + 25, look_at_idx


def calculate_target_yaws(camera_params, n_frames, segment_id):
    """yaw_diff per frame in radians: rad(yaw_i) - atan2(xL - x_i, zL - z_i)  (unified_loop_consistency.py:308-324)."""
    look_at_idx = (segment_id + 1) * 24 + 24
    out = []
    for i in range(n_frames):
        if i + 1 <= len(camera_params):
            L = camera_params[min(look_at_idx, len(camera_params) - 1)]
            tgt = math.atan2(L[0] - camera_params[i][0], L[2] - camera_params[i][2])
            out.append(math.radians(camera_params[i][4]) - tgt)
        else:
            out.append(0.0)
    return np.array(out, dtype=float)


class Equi2Pers:
    """pyequilib Equi2Pers(height, width, fov_x, mode='bilinear') restated as one HIP gather kernel (ew_equi2pers).
    __call__(equi CHW uint8, rots={'pitch','roll','yaw'}) -> CHW uint8.  Positive yaw turns the view to the LEFT
    (pyequilib z-up convention; the reference passes current_yaw - target_yaw, unified_loop_consistency.py:321)."""

    def __init__(self, height=384, width=512, fov_x=90.0, mode="bilinear", **_):
        if mode != "bilinear":
            raise NotImplementedError("only bilinear sampling (what the reference configures) is implemented")
        self.height, self.width, self.fov_x = height, width, fov_x

    @staticmethod
    def rotation(rots):
        y, p, r = -float(rots.get("yaw", 0.0)), float(rots.get("pitch", 0.0)), float(rots.get("roll", 0.0))
        Ry = np.array([[math.cos(y), 0, math.sin(y)], [0, 1, 0], [-math.sin(y), 0, math.cos(y)]])
        Rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]])
        Rz = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]])
        return (Ry @ Rx @ Rz).astype(np.float32)

    def batch(self, equi_hwc, rots_list):
        """equi uint8 [F,He,We,3] on the device, list of rots dicts -> uint8 [F,Hp,Wp,3] on the device."""
        rot = torch.from_numpy(np.stack([self.rotation(r) for r in rots_list])).to(equi_hwc.device)
        return ops.equi2pers(equi_hwc.contiguous(), rot.contiguous(), self.height, self.width, self.fov_x)

    def __call__(self, equi, rots):
        e = equi if isinstance(equi, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(equi))
        out = self.batch(e.permute(1, 2, 0)[None].cuda().contiguous(), [rots])[0].permute(2, 0, 1)
        return out if isinstance(equi, torch.Tensor) else out.cpu().numpy()


def split_curve_into_segments(path):
    """25-pose windows with 1-pose overlap (evoworld/inference/navigator_evoworld.py:303-318)."""
    total = len(path)
    if total < 25:
        return [path]
    segs, s, e = [], 0, 25
    while e <= total:
        segs.append(path[s:e])
        s = e - 1
        e = s + 25
    if e - s > 1 and s < total:
        segs.append(path[s:])
    return segs


# ------------------------------------------------------------------ R7: memory panoramas -> pipeline input (PIL-exact resize)
_coeff_cache = {}


def resample_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0) over the full box:
    returns (kk int32 [out, ksize], bounds int32 [out, 2] = (xmin, count)).  This is what
    torchvision.transforms.Resize((576,1024)) does to the 1000x2000 PIL memory panoramas
    (dataset/CameraTrajDataset.py:586-619, unified_loop_consistency.py:422); verified bit-exact against PIL itself."""
    key = (in_size, out_size)
    if key in _coeff_cache:
        return _coeff_cache[key]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)], dtype=np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        q = w * (1 << 22)
        kk[xx, :xmax] = np.where(q < 0, q - 0.5, q + 0.5).astype(np.int64)      # C cast: truncation toward zero
        bounds[xx] = (xmin, xmax)
    out = (torch.from_numpy(kk), torch.from_numpy(bounds))
    _coeff_cache[key] = out
    return out


def memory_to_pixel_values(panos_u8, height=576, width=1024):
    """uint8 [V,Hi,Wi,3] (device) -> fp32 [V,3,height,width] in [-1,1]: Resize (PIL bilinear, antialiased) -> ToTensor ->
    x*2-1, without leaving the GPU (the reference goes numpy -> PIL -> torch per panorama, unified_loop_consistency.py:422)."""
    V, Hi, Wi, _ = panos_u8.shape
    dev = panos_u8.device
    ch = tuple(t.to(dev) for t in resample_coeffs(Wi, width))
    cv = tuple(t.to(dev) for t in resample_coeffs(Hi, height))
    small = ops.resize_aa_u8(panos_u8.contiguous(), ch, cv, height, width) if (Hi, Wi) != (height, width) else panos_u8.contiguous()
    return ops.u8_hwc_to_f32_chw(small)
