"""TEST INFRASTRUCTURE -- CPU (numpy / torch-CPU) restatement of the reprojection stage and the denoise glue.
A checker only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by evoworld_amd/.

Pinned by goldens generated from the reference itself (oracle/make_goldens.py -> tests/golden/):
  cube2equi_lut_ref / cube2equi_gather_ref  <- reproject_vggt_open3d_utils.py:542-614   (bit-exact LUT, K2)
  align_first_and_last_points_ref, rotation_from_vectors_ref <- :1126-1213               (K3)
  target_c2w_ref                            <- SceneBuilder.align_extrinsics :472-519
  extract_colors_ref / confidence_filter_ref <- :286-310
  calculate_segment_indices_ref / split_curve_into_segments_ref <- pano_to_pers_utils.py:5, navigator_evoworld.py:303
PARITY UNPINNED (third-party engines absent from /root/reference, no golden available):
  splat_ref        <- Open3D 0.18 OffscreenRenderer point rendering (Filament GL), driven by :617-666.
                      Restated as: w2c transform, u = fx*x/z+cx, v = fy*y/z+cy, pixel = floor, z > near,
                      nearest depth wins (ties -> lowest point index), colour = point RGB, background 0.
  depth_unproject_ref <- facebookresearch/vggt unproject_depth_map_to_point_map (unified_loop_consistency.py:365)
  equi2pers_ref    <- pyequilib==0.5.8 Equi2Pers (unified_loop_consistency.py:178-183,329), bilinear.
The fp32 expression trees of splat_ref mirror evoworld_amd/csrc/geometry.hip op for op (no FMA), so the integer
pixel-index path is compared bit-exactly.
"""
import numpy as np
import torch

FACE_ORDER = ["right", "left", "bottom", "top", "front", "back"]           # face ids 0..5 (:583-590)
CUBEMAP_TRANSFORMS = {                                                       # :29-36
    "front": np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64),
    "right": np.array([[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
    "back": np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64),
    "left": np.array([[0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
    "top": np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
    "bottom": np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64),
}


def cube2equi_lut_ref(W, H, res):
    """int16 [H,W,3] = (face, v_px, u_px), computed with torch CPU float32 ops in the reference's order (:546-607)."""
    x = torch.linspace(0, W - 1, W)
    y = torch.linspace(0, H - 1, H)
    xv, yv = torch.meshgrid(y, x, indexing="ij")          # xv = row index, yv = column index (names as in the reference)
    lon = (-yv / W) * 2 * torch.pi - torch.pi + torch.pi / 2
    lat = (xv / H) * torch.pi - torch.pi / 2
    X = torch.cos(lat) * torch.cos(lon)
    Y = torch.sin(lat)
    Z = torch.cos(lat) * torch.sin(lon)
    aX, aY, aZ = X.abs(), Y.abs(), Z.abs()
    face = torch.zeros((H, W), dtype=torch.int64)
    u = torch.zeros_like(X)
    v = torch.zeros_like(Y)
    masks = {
        "right": (aX >= aY) & (aX >= aZ) & (X > 0), "left": (aX >= aY) & (aX >= aZ) & (X < 0),
        "bottom": (aY >= aX) & (aY >= aZ) & (Y > 0), "top": (aY >= aX) & (aY >= aZ) & (Y < 0),
        "front": (aZ >= aX) & (aZ >= aY) & (Z > 0), "back": (aZ >= aX) & (aZ >= aY) & (Z < 0),
    }
    for fi, f in enumerate(FACE_ORDER):                    # later faces overwrite earlier ones on ties
        m = masks[f]
        face[m] = fi
        if f in ("right", "left"):
            u[m] = -Z[m] / aX[m] if f == "right" else Z[m] / aX[m]
            v[m] = -Y[m] / aX[m]
        elif f in ("bottom", "top"):
            u[m] = -X[m] / aY[m]
            v[m] = -Z[m] / aY[m] if f == "bottom" else Z[m] / aY[m]
        else:
            u[m] = X[m] / aZ[m] if f == "front" else -X[m] / aZ[m]
            v[m] = -Y[m] / aZ[m]
    u = (u + 1) / 2
    v = (v + 1) / 2
    u_px = (u * (res - 1)).long()
    v_px = ((1 - v) * (res - 1)).long()
    return torch.stack([face, v_px, u_px], -1).to(torch.int16).numpy()


def cube2equi_gather_ref(faces, lut):
    """faces uint8 [V,6,res,res,3] (FACE_ORDER), lut [H,W,3] -> uint8 [V,H,W,3]."""
    f, v, u = lut[..., 0].astype(np.int64), lut[..., 1].astype(np.int64), lut[..., 2].astype(np.int64)
    return faces[:, f, v, u, :]


def rotation_from_vectors_ref(u, v):
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
    ang = np.arccos(dot)
    w = np.cross(uh, vh)
    wh = w / np.linalg.norm(w)
    K = np.array([[0, -wh[2], wh[1]], [wh[2], 0, -wh[0]], [-wh[1], wh[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1.0 - np.cos(ang)) * (K @ K)


def align_first_and_last_points_ref(A, B):
    A0, A1, B0, B1 = A[0], A[-1], B[0], B[-1]
    vA, vB = A1 - A0, B1 - B0
    lA, lB = np.linalg.norm(vA), np.linalg.norm(vB)
    if lA < 1e-15:
        return 1.0, np.eye(3), B0 - A0
    s = lB / lA
    R = rotation_from_vectors_ref(vA, vB)
    return s, R, B0 - s * R @ A0


def target_c2w_ref(camera_pose, extrinsic, segment_id, num_target_view=24):
    """SceneBuilder.align_extrinsics (:472-519) for only_render_last_24_frame=False, float64."""
    n = len(extrinsic)
    E = np.zeros((n, 4, 4))
    E[:, :3, :4] = extrinsic
    E[:, 3, 3] = 1
    inv = np.stack([np.linalg.inv(e) for e in E])
    start = (segment_id + 1) * num_target_view + 1
    gt = np.asarray(camera_pose)      # dtype preserved: the reference feeds float32 poses, so vA = A1 - A0 is a float32 op
    s, R, t = align_first_and_last_points_ref(gt[:start][:, :3, 3], inv[:, :3, 3])
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = t
    return np.einsum("ij,bjk->bik", T, gt[start:start + num_target_view])


def face_w2c_ref(target_c2w):
    """[V,6,3,4] float32 world->camera per (view, face) in FACE_ORDER: inv(c2w @ T_face [@ Rz180 for top/bottom])
    (render_cubemap :636-666, render_face :617-623)."""
    Fz = np.eye(4)
    Fz[:3, :3] = np.diag([-1.0, -1.0, 1.0])                # Rotation.from_euler('z', 180 deg)
    out = np.zeros((len(target_c2w), 6, 3, 4), dtype=np.float32)
    for v, c2w in enumerate(target_c2w):
        for fi, f in enumerate(FACE_ORDER):
            pose = c2w @ CUBEMAP_TRANSFORMS[f]
            if f in ("top", "bottom"):
                pose = pose @ Fz
            out[v, fi] = np.linalg.inv(pose)[:3, :4].astype(np.float32)
    return out


def splat_ref(xyz, rgb, w2c, res, fx, fy, cx, cy, z_near):
    """xyz f32 [N,3], rgb u8 [N,3], w2c f32 [V,6,3,4] -> faces u8 [V,6,res,res,3], zbuf u64 [V,6,res,res]."""
    f32 = np.float32
    xyz = np.asarray(xyz, dtype=f32)
    V = w2c.shape[0]
    zbuf = np.full((V, 6, res, res), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    idx = np.arange(xyz.shape[0], dtype=np.uint64)
    fx, fy, cx, cy, z_near = f32(fx), f32(fy), f32(cx), f32(cy), f32(z_near)
    for v in range(V):
        for f in range(6):
            M = w2c[v, f].astype(f32)
            zc = ((M[2, 0] * x + M[2, 1] * y) + M[2, 2] * z) + M[2, 3]
            ok = zc > z_near
            xc = ((M[0, 0] * x + M[0, 1] * y) + M[0, 2] * z) + M[0, 3]
            yc = ((M[1, 0] * x + M[1, 1] * y) + M[1, 2] * z) + M[1, 3]
            with np.errstate(divide="ignore", invalid="ignore"):
                pu = (fx * xc) / zc + cx
                pv = (fy * yc) / zc + cy
            fu, fv = np.floor(pu), np.floor(pv)
            ok &= (fu >= 0) & (fu < res) & (fv >= 0) & (fv < res)
            iu, iv = fu[ok].astype(np.int64), fv[ok].astype(np.int64)
            key = (zc[ok].view(np.uint32).astype(np.uint64) << np.uint64(32)) | idx[ok]
            np.minimum.at(zbuf[v, f].reshape(-1), iv * res + iu, key)
    faces = np.zeros((V, 6, res, res, 3), dtype=np.uint8)
    hit = zbuf != np.uint64(0xFFFFFFFFFFFFFFFF)
    faces[hit] = np.asarray(rgb)[(zbuf[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)]
    return faces, zbuf


def depth_unproject_ref(depth, extr, intr):
    """depth [S,H,W] f32, extr [S,3,4] (world->cam), intr [S,3,3] -> [S,H,W,3]: Xw = R^T (Xc - t)."""
    S, H, W = depth.shape
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    out = np.zeros((S, H, W, 3), dtype=np.float64)
    for s in range(S):
        K, E = intr[s].astype(np.float64), extr[s].astype(np.float64)
        z = depth[s].astype(np.float64)
        xc = (u - K[0, 2]) * z / K[0, 0]
        yc = (v - K[1, 2]) * z / K[1, 1]
        a = np.stack([xc - E[0, 3], yc - E[1, 3], z - E[2, 3]], -1)
        out[s] = a @ E[:, :3]                                # R^T a  (row-vector form)
    return out.astype(np.float32)


def equi2pers_ref(equi, rot, Hp, Wp, fov_x_deg):
    """equi u8 [F,He,We,3], rot [F,3,3] -> u8 [F,Hp,Wp,3], float64 restatement of the sampling grid + bilinear."""
    F_, He, We, _ = equi.shape
    focal = Wp / (2.0 * np.tan(np.radians(fov_x_deg) / 2.0))
    px, py = np.meshgrid(np.arange(Wp, dtype=np.float64), np.arange(Hp, dtype=np.float64))
    cam = np.stack([(px - Wp * 0.5) / focal, (py - Hp * 0.5) / focal, np.ones_like(px)], -1)
    out = np.zeros((F_, Hp, Wp, 3), dtype=np.float64)
    for f in range(F_):
        d = cam @ rot[f].astype(np.float64).T
        lon = np.arctan2(d[..., 0], d[..., 2])
        lat = np.arcsin(d[..., 1] / np.linalg.norm(d, axis=-1))
        ui = (lon * We / (2 * np.pi) + We * 0.5 + 0.5) % We
        uj = np.clip(lat * He / np.pi + He * 0.5 + 0.5, 0, He - 1)
        x0, y0 = np.floor(ui).astype(np.int64) % We, np.floor(uj).astype(np.int64)
        ax, ay = (ui - np.floor(ui))[..., None], (uj - np.floor(uj))[..., None]
        x1, y1 = (x0 + 1) % We, np.minimum(y0 + 1, He - 1)
        img = equi[f].astype(np.float64)
        top = img[y0, x0] * (1 - ax) + img[y0, x1] * ax
        bot = img[y1, x0] * (1 - ax) + img[y1, x1] * ax
        out[f] = top * (1 - ay) + bot * ay
    return out                                                # float; caller compares to uint8 with +-1 tolerance


def extract_colors_ref(images):
    c = np.transpose(images, (0, 2, 3, 1)) if (images.ndim == 4 and images.shape[1] == 3) else images
    return (c.reshape(-1, 3) * 255).astype(np.uint8)


def confidence_filter_ref(points, conf, colors, conf_thres):
    cf = conf.reshape(-1)
    thr = 0.0 if conf_thres == 0.0 else np.percentile(cf, conf_thres)
    m = cf >= thr
    if not np.any(m):
        return np.array([[1, 0, 0]]), np.array([[255, 255, 255]])
    return points.reshape(-1, 3)[m], colors[m]


def calculate_segment_indices_ref(segment_id):
    look_at = (segment_id + 1) * 24 + 24
    start = segment_id * 24 + 1
    if segment_id == 0:
        start -= 1
    return start, start + 25, look_at


def split_curve_into_segments_ref(n):
    if n < 25:
        return [(0, n)]
    segs, s, e = [], 0, 25
    while e <= n:
        segs.append((s, e))
        s = e - 1
        e = s + 25
    if e - s > 1 and s < n:
        segs.append((s, n))
    return segs


def euler_cfg_step_ref(eps_u, eps_c, latents, guidance, sigma, sigma_next):
    """fp32 torch restatement of pipeline_evoworld.py:709-714 (CFG + EulerDiscreteScheduler.step, v-prediction)."""
    e = eps_u + guidance.view(1, -1, 1, 1, 1) * (eps_c - eps_u)
    s = torch.tensor(sigma, dtype=torch.float32)
    x0 = e * (-s / (s ** 2 + 1) ** 0.5) + latents / (s ** 2 + 1)
    return latents + (latents - x0) / s * (sigma_next - sigma)
