"""-m 'not gpu': the host-side coefficient tables of the antialiased resize (R7) reproduce Pillow bit-exactly
(PIL is the reference's own resize engine: torchvision.transforms.Resize on PIL images, CameraTrajDataset.py:586-619)."""
import numpy as np
from PIL import Image

from evoworld_amd.reprojection import resample_coeffs


def _emulate(img, Ho, Wo):
    Hi, Wi, _ = img.shape
    kh, bh = [t.numpy() for t in resample_coeffs(Wi, Wo)]
    kv, bv = [t.numpy() for t in resample_coeffs(Hi, Ho)]
    tmp = np.zeros((Hi, Wo, 3), np.uint8)
    for xo in range(Wo):
        x0, n = bh[xo]
        tmp[:, xo] = np.clip(((1 << 21) + (img[:, x0:x0 + n].astype(np.int64) * kh[xo, :n][None, :, None]).sum(1)) >> 22, 0, 255)
    out = np.zeros((Ho, Wo, 3), np.uint8)
    for yo in range(Ho):
        y0, n = bv[yo]
        out[yo] = np.clip(((1 << 21) + (tmp[y0:y0 + n].astype(np.int64) * kv[yo, :n][:, None, None]).sum(0)) >> 22, 0, 255)
    return out


def test_resample_tables_match_pillow():
    rng = np.random.default_rng(0)
    for (Hi, Wi, Ho, Wo) in ((100, 200, 57, 102), (50, 100, 72, 128), (125, 250, 72, 128), (33, 77, 33, 20)):
        img = rng.integers(0, 256, size=(Hi, Wi, 3), dtype=np.uint8)
        ref = np.array(Image.fromarray(img).resize((Wo, Ho), Image.BILINEAR))
        assert np.array_equal(_emulate(img, Ho, Wo), ref), (Hi, Wi, Ho, Wo)
    kk, bounds = resample_coeffs(2000, 1024)
    assert kk.shape == (1024, 5) and int(kk.sum(1).min()) >= (1 << 22) - 4 and int(kk.sum(1).max()) <= (1 << 22) + 4
