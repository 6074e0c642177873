"""CPU tests of the drop-in entry point's host logic (no GPU, no HIP library calls): the reference's CLI flags and defaults
(unified_loop_consistency.py:542-571), camera_poses.txt parsing with the Unity->OpenCV flip (:370-395), episode discovery,
and the stage-provider hook."""
import os
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_loop_consistency as cli  # noqa: E402


def test_cli_flags_and_defaults_match_reference():
    a = cli.parse_arguments(["--unet_path", "CKPT"])
    assert a.svd_path == "stabilityai/stable-video-diffusion-img2vid-xt-1-1"
    assert (a.base_folder, a.save_dir, a.dataset_name) == ("data/Curve_Loop/test", "unified_output", "CameraTrajDataset")
    assert (a.num_data, a.start_idx, a.num_segments, a.num_frames, a.seed) == (1, 0, 3, 25, 42)
    assert not (a.save_frames or a.curve_path or a.single_segment)
    b = cli.parse_arguments(["--unet_path", "X", "--save_frames", "--curve_path", "--single_segment", "--num_segments", "5"])
    assert b.save_frames and b.curve_path and b.single_segment and b.num_segments == 5
    with pytest.raises(SystemExit):
        cli.parse_arguments([])                                   # --unet_path is required, as in the reference


def test_camera_poses_parsing(tmp_path):
    ep = tmp_path / "case_000"
    ep.mkdir()
    (ep / "camera_poses.txt").write_text("Frame,PosX,PosY,PosZ,RotX,RotY,RotZ\n1,0.25,1.78,14.9,0.0,95.5,0.0\n\n2, 0.65 ,1.78,14.92,1.0,97.75,2.0\nbad,row\n")
    cam = cli.load_camera_poses(str(ep))
    assert cam.shape == (2, 6)
    np.testing.assert_allclose(cam[0], [0.25, -1.78, 14.9, -0.0, 95.5, -0.0])      # y and the x/z rotations flip sign
    np.testing.assert_allclose(cam[1], [0.65, -1.78, 14.92, -1.0, 97.75, -2.0])
    with pytest.raises(FileNotFoundError):
        cli.load_camera_poses(str(tmp_path / "nope"))
    (ep / "camera_poses.txt").write_text("Frame,PosX\n")
    with pytest.raises(ValueError):
        cli.load_camera_poses(str(ep))


def test_episode_discovery(tmp_path):
    assert cli.list_episodes(str(tmp_path / "missing")) == []
    for n in ("b_ep", "a_ep"):
        (tmp_path / n).mkdir()
        (tmp_path / n / "camera_poses.txt").write_text("1,0,0,0,0,0,0\n")
    (tmp_path / "no_poses").mkdir()
    eps = cli.list_episodes(str(tmp_path))
    assert [os.path.basename(e) for e in eps] == ["a_ep", "b_ep"]
    assert cli.list_episodes(str(tmp_path / "a_ep")) == [str(tmp_path / "a_ep")]   # a folder that IS an episode
    assert cli.synthetic_episode(33).shape == (33, 6)


def test_stage_provider_hook():
    from evoworld_amd.stages import SyntheticStages, load_stages
    assert isinstance(load_stages(None, None), SyntheticStages)
    mod = types.ModuleType("fake_stage_provider")
    mod.make_stages = lambda args: ("made", args)
    mod.other = lambda args: ("other", args)
    sys.modules["fake_stage_provider"] = mod
    try:
        assert load_stages("fake_stage_provider", 7) == ("made", 7)
        assert load_stages("fake_stage_provider:other", 8) == ("other", 8)
    finally:
        del sys.modules["fake_stage_provider"]
