"""-m gpu: reprojection kernels through the C ABI against the CPU oracle (oracle/reproject_ref.py).
Integer paths (splat pixel index + z-test winner, LUT gather, order-preserving filter) are compared BIT-EXACTLY;
floating-point paths state their tolerance."""
import numpy as np
import pytest
import torch

from oracle import reproject_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cloud(n, seed, spread=4.0):
    rng = np.random.default_rng(seed)
    xyz = (rng.normal(size=(n, 3)) * spread).astype(np.float32)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    return xyz, rgb


def _views(V, seed):
    rng = np.random.default_rng(seed)
    c2w = np.repeat(np.eye(4)[None], V, 0)
    for v in range(V):
        a = rng.uniform(-np.pi, np.pi)
        c2w[v, :3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) * 1.3   # scale folded in, as align_extrinsics does
        c2w[v, :3, 3] = rng.normal(size=3)
    return c2w


@pytest.mark.parametrize("n,V,res", [(20000, 3, 64), (200000, 2, 128), (1, 1, 16), (5000, 24, 32)])
def test_splat_bit_exact_vs_oracle(n, V, res):
    from evoworld_amd import ops
    xyz, rgb = _cloud(n, 1)
    w2c = R.face_w2c_ref(_views(V, 2))
    f = res / 2.0
    faces, zbuf = ops.splat_cubemap(torch.tensor(xyz).to(DEV), torch.tensor(rgb).to(DEV), torch.tensor(w2c).to(DEV), res, f, f, f, f, 0.1)
    want_faces, want_z = R.splat_ref(xyz, rgb, w2c, res, f, f, f, f, 0.1)
    assert np.array_equal(zbuf.cpu().numpy().view(np.uint64), want_z)       # depth bits + winning point index per pixel
    assert np.array_equal(faces.cpu().numpy(), want_faces)


def test_splat_empty_cloud_and_behind_camera():
    from evoworld_amd import ops
    w2c = torch.tensor(R.face_w2c_ref(np.eye(4)[None])).to(DEV)
    xyz = torch.tensor([[0.0, 0.0, 0.05], [0.0, 0.0, 2.0]]).to(DEV)          # first point is closer than z_near on every face? no: only front
    rgb = torch.tensor([[255, 0, 0], [0, 255, 0]], dtype=torch.uint8).to(DEV)
    faces, _ = ops.splat_cubemap(xyz, rgb, w2c, 8, 4.0, 4.0, 4.0, 4.0, 0.1)
    f = faces.cpu().numpy()
    assert (f[0, 4, 4, 4] == [0, 255, 0]).all() and f.reshape(-1, 3).any(1).sum() == 1
    faces0, z0 = ops.splat_cubemap(xyz[:0].contiguous(), rgb[:0].contiguous(), w2c, 8, 4.0, 4.0, 4.0, 4.0, 0.1)
    assert not faces0.any() and (z0 == -1).all()


def test_cube2equi_full_size_vs_oracle(golden_dir):
    from evoworld_amd import ops
    from evoworld_amd.reprojection import build_cube2equi_lut
    lut = build_cube2equi_lut(2000, 1000, 512)
    assert np.array_equal(lut.numpy(), np.load(f"{golden_dir}/cube2equi_lut.npz")["lut_2000x1000x512"])
    g = torch.Generator().manual_seed(0)
    faces = torch.randint(0, 256, (2, 6, 512, 512, 3), generator=g, dtype=torch.uint8)
    pano = ops.cube2equi_gather(faces.to(DEV), lut.to(DEV), 1000, 2000)
    assert np.array_equal(pano.cpu().numpy(), R.cube2equi_gather_ref(faces.numpy(), lut.numpy()))


def test_depth_unproject_vs_oracle():
    from evoworld_amd import ops
    rng = np.random.default_rng(0)
    S, H, W = 3, 28, 36
    depth = rng.uniform(1, 20, size=(S, H, W)).astype(np.float32)
    intr = np.repeat(np.array([[[259.0, 0, W / 2], [0, 259.0, H / 2], [0, 0, 1]]], dtype=np.float32), S, 0)
    extr = np.linalg.inv(_views(S, 5) / np.array([1.3, 1.3, 1.3, 1])[None, None, :] * 1.0)[:, :3, :4].astype(np.float32)
    got = ops.depth_unproject(torch.tensor(depth).to(DEV), torch.tensor(extr).to(DEV), torch.tensor(intr).to(DEV)).cpu().numpy()
    want = R.depth_unproject_ref(depth, extr, intr)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)                # fp32 kernel vs float64 oracle


def test_equi2pers_vs_oracle():
    from evoworld_amd.reprojection import Equi2Pers
    g = torch.Generator().manual_seed(1)
    # smooth image (bilinear differences between fp32 and fp64 stay < 1 LSB except at exact .0 truncation ties)
    yy, xx = torch.meshgrid(torch.arange(72.0), torch.arange(144.0), indexing="ij")
    img = torch.stack([(xx * 1.7) % 256, (yy * 3.1) % 256, (xx + yy) % 256], -1).to(torch.uint8)
    e2p = Equi2Pers(48, 64, 90.0)
    rots = [{"yaw": 0.0, "pitch": 0.0, "roll": 0.0}, {"yaw": 0.7, "pitch": 0, "roll": 0}, {"yaw": -2.9, "pitch": 0, "roll": 0}]
    equi = img[None].repeat(3, 1, 1, 1).to(DEV)
    got = e2p.batch(equi, rots).cpu().numpy().astype(np.int32)
    want = R.equi2pers_ref(equi.cpu().numpy(), np.stack([e2p.rotation(r) for r in rots]), 48, 64, 90.0)
    diff = np.abs(got - np.floor(want))
    assert (diff <= 1).mean() > 0.995 and np.median(diff) == 0                 # tolerance: +-1 LSB (wrap seams excepted)
    # yaw=0 looks at the panorama centre: the central output pixel samples around column We/2 + 0.5
    assert abs(int(got[0, 24, 32, 0]) - int(((72 + 0.5) * 1.7) % 256)) <= 2


def test_predictions_to_target_view_end_to_end_vs_oracle(tmp_path):
    """Synthetic VGGT-like predictions -> 24 panoramas, HIP path vs oracle composition."""
    from evoworld_amd import reprojection as RP
    from evoworld_amd.geometry import xyz_euler_to_four_by_four_matrix_batch
    rng = np.random.default_rng(0)
    S, H, W = 25, 28, 36
    i = torch.arange(60, dtype=torch.float32)
    poses = torch.stack([0.04 * i * torch.sin(i / 9), torch.zeros(60), 0.04 * i * torch.cos(i / 9), torch.zeros(60), 95 + 3.6 * i, torch.zeros(60)], 1)
    gt = xyz_euler_to_four_by_four_matrix_batch(poses, relative=True)                    # float32 [60,4,4]
    extr = np.linalg.inv(gt[:S].numpy().astype(np.float64))[:, :3, :4].astype(np.float32)    # VGGT frame == GT frame (s=1)
    depth = rng.uniform(1, 8, size=(S, H, W, 1)).astype(np.float32)
    conf = rng.uniform(0, 1, size=(S, H, W)).astype(np.float32)
    images = rng.uniform(0, 1, size=(S, 3, H, W)).astype(np.float32)
    intr = np.repeat(np.array([[[W / 2, 0, W / 2], [0, W / 2, H / 2], [0, 0, 1]]], dtype=np.float32), S, 0)
    preds = {"depth": depth, "depth_conf": conf, "images": images, "extrinsic": extr, "intrinsic": intr}
    outdir = str(tmp_path / "rendered_panorama_vggt_open3d_0")
    cr = RP.CubemapRenderer(face_res=64)
    pan = RP.CubemapRenderer.render_cubemaps_to_panoramas   # noqa: F841 (surface check)
    got = RP.predictions_to_target_view(dict(preds), gt, conf_thres=50.0, prediction_mode="depth_unproject", num_target_view=24,
                                        outdir=outdir, cubemap_renderer=_Small(cr), save_png=False)
    # oracle composition on the SAME lifted points (the lift itself is checked in test_depth_unproject_vs_oracle)
    from evoworld_amd import ops
    xyz = ops.depth_unproject(torch.tensor(depth[..., 0]).to(DEV), torch.tensor(extr).to(DEV), torch.tensor(intr).to(DEV)).cpu().numpy()
    cols = R.extract_colors_ref(images)
    v, c = R.confidence_filter_ref(xyz, conf, cols, 50.0)
    tgt = R.target_c2w_ref(gt.numpy(), extr, 0)
    faces, _ = R.splat_ref(v, c, R.face_w2c_ref(tgt), 64, 32.0, 32.0, 32.0, 32.0, 0.1)
    want = R.cube2equi_gather_ref(faces, R.cube2equi_lut_ref(256, 128, 64))
    assert got.shape == (24, 128, 256, 3) and got.dtype == np.uint8
    assert np.array_equal(got, want)


class _Small:
    """CubemapRenderer at 64^2 faces / 256x128 panoramas so that the oracle finishes in seconds."""

    def __init__(self, cr):
        self.cr = cr

    def render_cubemaps_to_panoramas(self, v, c, target, n, outdir):
        return self.cr.render_cubemaps_to_panoramas(v, c, target, n, outdir, width=256, height=128)


@pytest.mark.parametrize("n", [1, 2, 5, 1000, 4097, 1_000_003])
def test_radix_select_and_percentile_match_numpy(n):
    """ew_select_kth_f32 against a numpy sort (ties, negatives, +-0, denormals), and the full percentile threshold against
    np.percentile (reproject_vggt_open3d_utils.py:297-310)."""
    from evoworld_amd import ops
    from evoworld_amd import reprojection as RP
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n).astype(np.float32)
    if n >= 1000:
        a[::7] = a[3]                       # heavy ties
        a[1:50] = 0.0
        a[51:60] = -0.0
        a[60:70] = 1e-42                    # denormals
    srt = np.sort(a)
    x = torch.tensor(a).to(DEV)
    for k in sorted({0, n // 2, max(0, n - 2), n - 1, min(n - 1, 17)}):
        got = ops.select_kth(x, k).cpu().numpy()
        assert got[0] == srt[k], (n, k)
        assert got[1] == srt[min(k + 1, n - 1)], (n, k)
    for q in (50.0, 30.0, 99.5):
        assert RP.percentile_threshold(x, q) == np.percentile(a, q), (n, q)


@pytest.mark.parametrize("nchw", [True, False])
def test_filter_compact_matches_boolean_mask(nchw):
    """order-preserving compaction + colour extraction == numpy boolean-mask semantics (:286-310)"""
    from evoworld_amd import ops
    S, H, W = 3, 37, 53
    n = S * H * W
    g = torch.Generator().manual_seed(0)
    conf = torch.rand(n, generator=g)
    conf[::5] = 0.5
    xyz = torch.randn(n, 3, generator=g)
    img = torch.rand(S, 3, H, W, generator=g)
    img[0, :, 0, :4] = torch.tensor([0.0, 1.0, 0.999999, 0.5])[None]
    thr = 0.5
    imgs = img if nchw else img.permute(0, 2, 3, 1).contiguous()
    v, rgbx = ops.filter_compact(conf.to(DEV), thr, xyz.to(DEV), imgs.to(DEV), H * W if nchw else 0)
    keep = conf.numpy() >= np.float32(thr)
    cols = (img.permute(0, 2, 3, 1).reshape(-1, 3).numpy() * 255).astype(np.uint8)
    assert v.shape[0] == int(keep.sum())
    assert np.array_equal(v.cpu().numpy(), xyz.numpy()[keep])
    assert np.array_equal(rgbx[:, :3].cpu().numpy(), cols[keep])


@pytest.mark.parametrize("H,W,ch", [(5, 9, 3), (5, 9, 4), (6, 10, 3)])
def test_cube2equi_gather_any_pixel_count(H, W, ch):
    """H*W % 4 != 0: the dword-store path would be misaligned for views >= 1 -- every pixel takes the byte path (round-2 advisor)."""
    from evoworld_amd import ops
    g = torch.Generator().manual_seed(3)
    V, res = 3, 8
    faces = torch.randint(0, 256, (V, 6, res, res, ch), dtype=torch.uint8, generator=g)
    lut = torch.stack([torch.randint(0, 6, (H, W), generator=g), torch.randint(0, res, (H, W), generator=g),
                       torch.randint(0, res, (H, W), generator=g)], -1).to(torch.int16)
    got = ops.cube2equi_gather(faces.to(DEV), lut.to(DEV), H, W).cpu()
    f, vv, uu = lut[..., 0].long(), lut[..., 1].long(), lut[..., 2].long()
    want = faces[:, f, vv, uu][..., :3]
    assert torch.equal(got, want)


def test_select_and_splat_accept_unaligned_views():
    """conf[frames] / xyz[1:] style views of tensors whose H*W is not a multiple of 4 start at odd offsets: the wrappers copy
    them to an aligned buffer instead of failing the 16-byte alignment check (round-2 advisor)."""
    from evoworld_amd import ops
    g = torch.Generator().manual_seed(4)
    base = torch.rand(3 * 37 + 5, generator=g).to(DEV)
    view = base[37 + 1:]                                            # data_ptr offset 152 bytes: 8-byte aligned only
    assert view.data_ptr() % 16 != 0
    k = 20
    got = ops.select_kth(view, k).cpu()
    srt = torch.sort(view.cpu()).values
    assert torch.equal(got, srt[k:k + 2])
    xyz_all = (torch.rand(50, 3, generator=g) * 4 - 2).to(DEV)
    xyz = xyz_all[1:]
    assert xyz.data_ptr() % 16 != 0
    rgb = torch.randint(0, 256, (49, 3), dtype=torch.uint8, generator=g).to(DEV)
    w2c = torch.eye(4)[:3][None, None].repeat(1, 6, 1, 1).contiguous().to(DEV)
    f1, z1 = ops.splat_cubemap(xyz, rgb, w2c, 16, 8.0, 8.0, 8.0, 8.0, 0.1)
    f2, z2 = ops.splat_cubemap(xyz.clone(), rgb, w2c, 16, 8.0, 8.0, 8.0, 8.0, 0.1)
    assert torch.equal(f1, f2) and torch.equal(z1, z2)
