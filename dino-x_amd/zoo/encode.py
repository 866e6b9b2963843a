"""``zoo.encode`` surface: raw HU array + physical spacing -> features, forward on the HIP engine.

Host-side preprocessing follows the reference (zoo/encode.py:34-72,129-169): convert to HU
(``hu16_png``: (u16 - 32768) * 0.1), window to [0,1] (level 40 / width 400 by default), replicate or
split into 3 channels, PIL bilinear resize to ``model.img_size``, ImageNet normalise, spacing tensor
only for scale-aware models.  ``encode`` returns ``(1, 1, D)`` (CLS) or all tokens ``(1, N, D)``;
``ValueError`` for an unknown ``input_format``, an unsupported shape or mismatched list lengths.
"""
from __future__ import annotations

from typing import List, Literal, Sequence, Tuple, Union

import numpy as np
import torch

from zoo.arch import PatchViT

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)
_FORMATS = ("hu_float", "hu16_png", "windowed_float")


def _to_hu(arr: np.ndarray, input_format: str) -> np.ndarray:
    if input_format not in _FORMATS:
        raise ValueError(f"Unknown input_format: '{input_format}'. Supported: 'hu_float', 'hu16_png', 'windowed_float'")
    a = arr.astype(np.float32)
    return (a - 32768.0) * 0.1 if input_format == "hu16_png" else a


def _hu_window(arr: np.ndarray, level: float = 40.0, width: float = 400.0) -> np.ndarray:
    lo, hi = level - width / 2, level + width / 2
    return (np.clip(arr, lo, hi) - lo) / (hi - lo)


def _resize(arr: np.ndarray, size: int) -> np.ndarray:
    from PIL import Image
    return np.array(Image.fromarray(arr).resize((size, size), Image.BILINEAR))


def _channels(arr: np.ndarray) -> List[np.ndarray]:
    if arr.ndim == 2:
        return [arr, arr, arr]
    if arr.ndim == 3 and arr.shape[2] == 3:
        return [arr[:, :, i] for i in range(3)]
    if arr.ndim == 3 and arr.shape[0] == 3:
        return [arr[i] for i in range(3)]
    raise ValueError(f"Unsupported image shape: {arr.shape}. Expected (H, W), (H, W, 3), or (3, H, W).")


def preprocess(image: np.ndarray, img_size: int, input_format: str, hu_level: float, hu_width: float) -> torch.Tensor:
    """-> (3, img_size, img_size) fp32, ImageNet-normalised."""
    arr = _to_hu(image, input_format)
    if input_format != "windowed_float":
        arr = _hu_window(arr, level=hu_level, width=hu_width)
    stack = np.stack([_resize(np.ascontiguousarray(c, dtype=np.float32), img_size) for c in _channels(arr)], axis=0)
    return torch.from_numpy(((stack.astype(np.float32) - _MEAN) / _STD).astype(np.float32))


def encode(model: PatchViT, image: np.ndarray, pixel_spacing: Tuple[float, float] = (1.0, 1.0), slice_thickness: float = 1.0, *,
           input_format: Literal["hu_float", "hu16_png", "windowed_float"] = "hu_float", hu_level: float = 40.0,
           hu_width: float = 400.0, return_all_tokens: bool = False,
           device: Union[str, torch.device, None] = None) -> torch.Tensor:
    if device is None:
        device = next(model.parameters()).device
    x = preprocess(image, model.img_size, input_format, hu_level, hu_width).unsqueeze(0).to(device)
    spacing = None
    if model.scale_aware:
        spacing = torch.tensor([[pixel_spacing[0], pixel_spacing[1], slice_thickness]], dtype=torch.float32, device=device)
    with torch.no_grad():
        feats = model(x, spacing=spacing)
    return feats if return_all_tokens else feats[:, 0:1, :]


def encode_batch(model: PatchViT, images: Sequence[np.ndarray], spacings: Sequence[Tuple[float, float, float]], *,
                 input_format: Literal["hu_float", "hu16_png", "windowed_float"] = "hu_float", hu_level: float = 40.0,
                 hu_width: float = 400.0, return_all_tokens: bool = False,
                 device: Union[str, torch.device, None] = None) -> torch.Tensor:
    """Same preprocessing per image as ``encode`` but ONE batched forward through the HIP engine."""
    if len(images) != len(spacings):
        raise ValueError(f"images ({len(images)}) and spacings ({len(spacings)}) must have same length")
    if device is None:
        device = next(model.parameters()).device
    x = torch.stack([preprocess(im, model.img_size, input_format, hu_level, hu_width) for im in images], 0).to(device)
    spacing = None
    if model.scale_aware:
        spacing = torch.tensor([list(s) for s in spacings], dtype=torch.float32, device=device)
    with torch.no_grad():
        feats = model(x, spacing=spacing)
    return feats if return_all_tokens else feats[:, 0:1, :]
