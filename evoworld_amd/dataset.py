"""Episode loader of the entry points: the build's counterpart of the slice of dataset/CameraTrajDataset.py the inference
scripts use (CameraTrajDataset.__getitem__ :292-371, load_images :414-445, load_reprojection :450-510, load_traj :512-526,
convert_to_opencv_rdf :373-395, pos_scale :223,348).  Host-side file I/O only; every image is resized on the device with
the Pillow-exact kernel (reprojection.memory_to_pixel_values = Resize -> ToTensor -> x*2-1).

Two modes, as unified_loop_consistency.py:215-239 builds the dataset:
  * single_segment=True   -> sampling_method "reprojection", load_complete_episode False: the episode's LAST 25 frames
                             (1-based file ids L-24 .. L), memory = [panorama/001] + rendered_panorama_vggt_open3d/00..NN.png
  * single_segment=False  -> "empty_with_traj", load_complete_episode True: the whole episode, zero memory.
`cam_traj` carries positions x pos_scale (the Navigator / Plucker path); the UNSCALED poses are what
unified_loop_consistency.py:370-395 re-reads from camera_poses.txt for yaws and reprojection alignment."""
import os

import numpy as np
import torch

from .geometry import UNITY_TO_OPENCV

POS_SCALE = 0.1            # CameraTrajDataset default (dataset/CameraTrajDataset.py:223)
SEQUENCE_LENGTH = 25


def load_camera_poses(episode_path):
    """camera_poses.txt 'Frame,PosX,PosY,PosZ,RotX,RotY,RotZ' -> float64 [P,6] in the OpenCV (RDF) convention, UNSCALED
    (unified_loop_consistency.py:370-395)."""
    f = os.path.join(episode_path, "camera_poses.txt")
    if not os.path.isfile(f):
        raise FileNotFoundError(f"camera_poses.txt not found under {episode_path}")
    rows = []
    for line in open(f):
        parts = [s.strip() for s in line.strip().split(",")]
        if len(parts) >= 7 and "rame" not in parts[0]:
            rows.append([float(x) for x in parts[1:7]])
    if not rows:
        raise ValueError(f"No valid camera pose rows parsed from {f}")
    return np.asarray(rows, dtype=float) * np.asarray(UNITY_TO_OPENCV, dtype=float)


def _open_rgb(path):
    from PIL import Image
    if not os.path.exists(path):
        alt = os.path.splitext(path)[0] + ".jpg"                     # the reference falls back to .jpg (:430-433)
        if os.path.exists(alt):
            path = alt
    return np.array(Image.open(path).convert("RGB"))


def _to_pixel_values(arrays, height, width, device):
    """list of uint8 [H,W,3] (equal sizes) -> fp32 [N,3,height,width] in [-1,1] on the device (Pillow-exact resize)."""
    from . import reprojection as RP
    u8 = torch.tensor(np.stack(arrays)).to(device)
    return RP.memory_to_pixel_values(u8, height, width)


def load_single_segment_batch(episode_path, height=576, width=1024, device="cuda", sequence_length=SEQUENCE_LENGTH,
                              pos_scale=POS_SCALE, reprojection_name="rendered_panorama_vggt_open3d", no_images=False):
    """The batch CameraTrajDataset(sampling 'reprojection', load_complete_episode=False)[idx] hands to process_batch, with the
    DataLoader's leading batch dim: pixel_values [1,25,3,H,W], cam_traj [1,25,6] (pos-scaled), memorized_pixel_values
    [1,1+N,3,H,W], episode_path."""
    cam = load_camera_poses(episode_path)
    L = cam.shape[0]
    if L < sequence_length:
        raise ValueError(f"episode has {L} poses, need at least {sequence_length}")
    start = L - sequence_length + 1                                   # 1-based id of the first frame (:313-328)
    ids = list(range(start, start + sequence_length))
    traj = torch.tensor(cam[start - 1: start - 1 + sequence_length], dtype=torch.float32)
    traj[:, :3] *= pos_scale                                          # :348
    rdir = os.path.join(episode_path, reprojection_name)
    if no_images:
        pix = torch.zeros(sequence_length, 3, height, width, device=device)
        mem = torch.zeros(sequence_length, 3, height, width, device=device)
    else:
        pix = _to_pixel_values([_open_rgb(os.path.join(episode_path, "panorama", f"{i:03}.png")) for i in ids], height, width, device)
        names = sorted(f for f in os.listdir(rdir) if f.endswith(".png"))
        renders = [_open_rgb(os.path.join(rdir, f"{i:02}.png")) for i in range(len(names))]      # :478-505
        first = _to_pixel_values([_open_rgb(os.path.join(episode_path, "panorama", "001.png"))], height, width, device)
        # first frame inserted at 0 (:506-514).  With sequence_length < 25 (BASELINE configs[0]: 8 frames) the reference's dataset still returns all
        # 1 + 24 memory frames and its pipeline then fails the channel concat at pipeline_evoworld.py:643; here the memory is cut to the window
        # length ([001] + renders 00 .. sequence_length - 2), which for the default 25 is the identity
        mem = torch.cat([first, _to_pixel_values(renders[:sequence_length - 1], height, width, device)], dim=0)
    return {"pixel_values": pix[None], "cam_traj": traj[None], "memorized_pixel_values": mem[None],
            "memorized_cam_traj": traj[None].clone(), "episode_path": [episode_path]}


def load_complete_episode_batch(episode_path, height=576, width=1024, device="cuda", pos_scale=POS_SCALE, cam=None):
    """CameraTrajDataset(sampling 'empty_with_traj', load_complete_episode=True)[idx] as far as process_episode reads it
    (unified_loop_consistency.py:241-268): the first frame and the complete, pos-scaled trajectory.  (The reference also
    loads every ground-truth panorama for its predictions_gt dumps; the generator never sees them.)"""
    cam = load_camera_poses(episode_path) if cam is None else np.asarray(cam, dtype=float)
    traj = torch.tensor(cam, dtype=torch.float32)
    traj[:, :3] *= pos_scale
    f = os.path.join(episode_path, "panorama", "001.png")
    if os.path.isfile(f) or os.path.isfile(f[:-4] + ".jpg"):
        first = _to_pixel_values([_open_rgb(f)], height, width, device)[0]
    else:
        first = None
    return {"first_frame": first, "cam_traj": traj[None], "camera_params": cam, "episode_path": [episode_path]}
