"""-m 'not gpu': the CPU oracle restatements against the golden vectors generated from the reference itself
(oracle/make_goldens.py).  These pin the oracle before it is trusted as the checker of the HIP path."""
import hashlib

import numpy as np
import torch

from oracle import reproject_ref as R


def test_cube2equi_lut_bit_exact(golden_dir):
    g = np.load(f"{golden_dir}/cube2equi_lut.npz")
    for (W, H, res), key in (((64, 32, 16), "lut_64x32x16"), ((256, 128, 64), "lut_256x128x64"),
                             ((2000, 1000, 512), "lut_2000x1000x512")):
        lut = R.cube2equi_lut_ref(W, H, res)
        assert lut.dtype == np.int16 and np.array_equal(lut, g[key]), key
    full = g["lut_2000x1000x512"]
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == \
        "e45a27dcc803835a3e6438d5cacf93806c71e076be94eaf3e80430da9ac96d50"
    assert np.bincount(full[..., 0].reshape(-1), minlength=6).tolist() == [232206, 232206, 533804, 535806, 232989, 232989]
    probes = {(0, 0): (3, 255, 255), (500, 1000): (4, 255, 255), (500, 0): (5, 255, 255), (500, 1500): (0, 255, 255),
              (250, 500): (3, 255, 511), (999, 1999): (2, 254, 255), (100, 700): (3, 206, 322)}
    for (r, c), want in probes.items():
        assert tuple(int(x) for x in full[r, c]) == want


def test_cube2equi_gather_matches_reference(golden_dir):
    g = np.load(f"{golden_dir}/cube2equi_gather.npz")
    lut = np.load(f"{golden_dir}/cube2equi_lut.npz")["lut_64x32x16"]
    faces = np.transpose(g["faces"], (0, 1, 3, 4, 2))
    assert np.array_equal(R.cube2equi_gather_ref(faces, lut), g["pano"])


def test_alignment_goldens(golden_dir):
    g = np.load(f"{golden_dir}/align.npz")
    for i in range(int(g["n"])):
        s, Rm, t = R.align_first_and_last_points_ref(g[f"A{i}"], g[f"B{i}"])
        np.testing.assert_allclose(s, g[f"s{i}"], rtol=1e-12)
        np.testing.assert_allclose(Rm, g[f"R{i}"], atol=1e-12)
        np.testing.assert_allclose(t, g[f"t{i}"], atol=1e-12)
    np.testing.assert_allclose(g["s0"], 1.2649110640673518, rtol=1e-14)
    tgt = R.target_c2w_ref(g["ax_gt"], g["ax_extr"], int(g["ax_seg"]))
    np.testing.assert_allclose(tgt, g["ax_target"], atol=1e-10)


def test_filter_goldens(golden_dir):
    g = np.load(f"{golden_dir}/filter.npz")
    cols = R.extract_colors_ref(g["images"])
    assert np.array_equal(cols, g["colors"])
    for thr, kv, kc in ((50.0, "v50", "c50"), (30.0, "v30", "c30"), (0.0, "v0", "c0")):
        v, c = R.confidence_filter_ref(g["points"], g["conf"], cols, thr)
        assert np.array_equal(v, g[kv]) and np.array_equal(c, g[kc])


def test_segment_math_goldens(golden_dir):
    g = np.load(f"{golden_dir}/segments.npz")
    assert [R.calculate_segment_indices_ref(i) for i in range(5)] == [tuple(r) for r in g["calculate_segment_indices"].tolist()]
    assert R.calculate_segment_indices_ref(0) == (0, 25, 48) and R.calculate_segment_indices_ref(2) == (49, 74, 96)
    for L in (10, 25, 26, 49, 73, 126):
        assert R.split_curve_into_segments_ref(L) == [tuple(r) for r in g[f"L{L}"].tolist()], L


def test_splat_ref_basic_properties():
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(5000, 3)).astype(np.float32) * 3
    rgb = rng.integers(1, 256, size=(5000, 3), dtype=np.uint8)
    c2w = np.repeat(np.eye(4)[None], 2, 0)
    c2w[1, :3, 3] = [0.2, -0.1, 0.3]
    w2c = R.face_w2c_ref(c2w)
    faces, zbuf = R.splat_ref(xyz, rgb, w2c, 32, 16.0, 16.0, 16.0, 16.0, 0.1)
    hit = zbuf != np.uint64(0xFFFFFFFFFFFFFFFF)
    assert hit.sum() > 1000 and (faces[~hit] == 0).all()
    # nearest point wins: add a very near duplicate of point 0 along the same ray -> it must own that pixel
    p = xyz[0] * 0.5
    faces2, zbuf2 = R.splat_ref(np.vstack([xyz, p[None]]), np.vstack([rgb, [[9, 8, 7]]]).astype(np.uint8), w2c[:1], 32, 16.0, 16.0, 16.0, 16.0, 0.01)
    assert (faces2.reshape(-1, 3) == np.array([9, 8, 7])).all(1).sum() == 1
