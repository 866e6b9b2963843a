"""CPU oracle of the 2.5D slice-stack view pipeline (TEST INFRASTRUCTURE -- imported only by tests/ and smoke()).

Restates what the reference's input pipeline computes per view (scripts/phase5_big_run.py:471-570):
  _load_hu01 (:516-528)      stored u16 -> HU = (u16 - 32768) * 0.1 -> clip((HU - (level - width/2)) / max(width, 1), 0, 1)
  transforms (:493-497)      RandomResizedCrop(img, scale, BICUBIC) -> RandomHorizontalFlip -> Normalize(ImageNet mean/std)
The random draws are inputs here (level, width, crop box, flip); the arithmetic is what is pinned.

RandomResizedCrop on a float tensor goes crop -> torchvision ``resize`` -> ``torch.nn.functional.interpolate(mode="bicubic",
align_corners=False, antialias=True)`` (torchvision 0.24 pinned by the reference's requirements.txt:13-16 is not installed here;
its tensor path is that one call and does not clamp float outputs).  ``interpolate`` is the third-party kernel
``_upsample_bicubic2d_aa`` of PyTorch; ``aa_bicubic_weights`` / ``resize_aa_bicubic`` below restate its published algorithm
(separable, Keys cubic a = -0.5 stretched by the down-scale factor, taps clipped at the image edge and renormalised) in
NumPy, and tests/test_slice_views.py checks the restatement against torch's kernel itself.  The reference holds no fixture
for this stage: parity is pinned to torch's kernel, not to outputs of the reference.
"""
from __future__ import annotations

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def hu_window01(u16: np.ndarray, level: float, width: float) -> np.ndarray:
    """scripts/phase5_big_run.py:516-528 (float32 arithmetic, python-float scalars)."""
    hu = (u16.astype(np.float32) - np.float32(32768.0)) * np.float32(0.1)
    wmin = level - width / 2.0
    return np.clip((hu - np.float32(wmin)) / np.float32(max(width, 1.0)), 0.0, 1.0).astype(np.float32)


def _cubic(x: np.ndarray) -> np.ndarray:
    a = np.float32(-0.5)
    x = np.abs(x).astype(np.float32)
    near = ((a + 2) * x - (a + 3)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1, near, np.where(x < 2, far, 0)).astype(np.float32)


def aa_bicubic_weights(in_size: int, out_size: int):
    """Per output index: (first input index, taps, normalised weights) of torch's antialiased bicubic (align_corners=False)."""
    scale = np.float32(in_size) / np.float32(out_size)
    support = np.float32(2.0) * scale if scale >= 1 else np.float32(2.0)
    inv = np.float32(1.0) / scale if scale >= 1 else np.float32(1.0)
    out = []
    for i in range(out_size):
        center = scale * np.float32(i + 0.5)
        xmin = max(int(center - support + np.float32(0.5)), 0)
        xsize = min(int(center + support + np.float32(0.5)), in_size) - xmin
        w = _cubic((np.arange(xsize, dtype=np.float32) + np.float32(xmin) - center + np.float32(0.5)) * inv)
        out.append((xmin, xsize, (w / w.sum(dtype=np.float32)).astype(np.float32)))
    return out


def resize_aa_bicubic(img: np.ndarray, size: int) -> np.ndarray:
    """(C,h,w) float32 -> (C,size,size): horizontal pass, then vertical pass."""
    C, h, w = img.shape
    wx, wy = aa_bicubic_weights(w, size), aa_bicubic_weights(h, size)
    tmp = np.empty((C, h, size), dtype=np.float32)
    for ox, (x0, n, wt) in enumerate(wx):
        tmp[:, :, ox] = (img[:, :, x0:x0 + n] * wt).sum(-1, dtype=np.float32)
    out = np.empty((C, size, size), dtype=np.float32)
    for oy, (y0, n, wt) in enumerate(wy):
        out[:, oy, :] = (tmp[:, y0:y0 + n, :] * wt[None, :, None]).sum(1, dtype=np.float32)
    return out


def make_view(stack_u16: np.ndarray, level: float, width: float, top: int, left: int, h: int, w: int, flip: bool, size: int,
              resize=resize_aa_bicubic) -> np.ndarray:
    """One augmented view: (3,H,W) u16 stack -> (3,size,size) float32."""
    x = np.stack([hu_window01(s, level, width) for s in stack_u16], 0)
    y = resize(np.ascontiguousarray(x[:, top:top + h, left:left + w]), size)
    if flip:
        y = y[:, :, ::-1]
    return ((y - MEAN[:, None, None]) / STD[:, None, None]).astype(np.float32)


def torch_resize(img: np.ndarray, size: int) -> np.ndarray:
    """The kernel the reference reaches through torchvision: F.interpolate(bicubic, antialias=True)."""
    import torch
    import torch.nn.functional as F
    return F.interpolate(torch.from_numpy(img)[None], size=(size, size), mode="bicubic", align_corners=False, antialias=True)[0].numpy()
