"""Device-side view pipeline for the 2.5D slice stacks.

The reference builds every training view on DataLoader workers (scripts/phase5_big_run.py:493-497, 516-528, 549-555): HU
decode, random window, RandomResizedCrop (bicubic), flip, normalise -- ~500 img/s on its box.  Here the workers only decode
PNGs to u16 stacks; the random draws happen on the host (cheap scalars, same order and distributions as the reference's
transform stack) and ONE kernel (csrc/views.hip, ``dinox_slice_views``) turns the resident u16 stacks into the fp32
``(2B, 3, S, S)`` batch that ``PatchViT.forward`` takes.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

RATIO = (3.0 / 4.0, 4.0 / 3.0)          # torchvision RandomResizedCrop default


def draw_crop_box(H: int, W: int, scale: Tuple[float, float], ratio: Tuple[float, float] = RATIO, rng=random) -> Tuple[int, int, int, int]:
    """torchvision RandomResizedCrop.get_params: (top, left, h, w); 10 tries, then the central crop at the clamped aspect."""
    area = H * W
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        ar = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
        cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < cw <= W and 0 < ch <= H:
            return rng.randint(0, H - ch), rng.randint(0, W - cw), ch, cw
    in_ratio = W / H
    h, w = H, W
    if in_ratio < ratio[0]:
        w, h = W, int(round(W / ratio[0]))
    elif in_ratio > ratio[1]:
        h, w = H, int(round(H * ratio[1]))
    return (H - h) // 2, (W - w) // 2, h, w


@dataclass
class ViewParams:
    """The random draws of one view (reference: level/width at :549-550, crop + flip inside the transform stack)."""
    level: float
    width: float
    top: int
    left: int
    h: int
    w: int
    flip: bool


def draw_view(H: int, W: int, rw_level=(-400.0, 400.0), rw_width=(800.0, 2000.0), crop_scale=(0.3, 1.0), rng=random) -> ViewParams:
    """Same draw order as the CPU pipeline: window level, window width, crop box, flip."""
    level = rng.uniform(rw_level[0], rw_level[1])
    width = rng.uniform(rw_width[0], rw_width[1])
    top, left, h, w = draw_crop_box(H, W, crop_scale, rng=rng)
    return ViewParams(level, width, top, left, h, w, rng.random() < 0.5)


@dataclass
class StackBatch:
    """A batch of u16 (3,H,W) slice stacks packed into one flat buffer (stacks may differ in size) plus the view draws:
    ``views[k][i]`` = k-th view of sample i.  ``spacing``: (B,3) fp32."""
    raw: torch.Tensor                    # flat int16/uint16 storage of every stack, C-order (3,H,W) each
    offsets: List[int]
    shapes: List[Tuple[int, int]]
    views: List[List[ViewParams]]
    spacing: torch.Tensor

    def to(self, device, non_blocking=True) -> "StackBatch":
        return StackBatch(self.raw.to(device, non_blocking=non_blocking), self.offsets, self.shapes, self.views,
                          self.spacing.to(device, non_blocking=non_blocking))


def collate_stacks(items: Sequence[Tuple[np.ndarray, Sequence[ViewParams], torch.Tensor]]) -> StackBatch:
    """DataLoader collate_fn: items are (u16 stack (3,H,W), [ViewParams per view], spacing (3,))."""
    offsets, shapes, chunks, total = [], [], [], 0
    for stack, _, _ in items:
        a = np.ascontiguousarray(stack, dtype=np.uint16)
        assert a.ndim == 3 and a.shape[0] == 3, a.shape
        offsets.append(total)
        shapes.append((a.shape[1], a.shape[2]))
        chunks.append(a.reshape(-1))
        total += a.size
    raw = torch.from_numpy(np.concatenate(chunks).view(np.int16))          # bit pattern; the kernel reads it as u16
    n_views = len(items[0][1])
    views = [[it[1][k] for it in items] for k in range(n_views)]
    return StackBatch(raw, offsets, shapes, views, torch.stack([it[2] for it in items], 0))


def make_views(batch: StackBatch, size: int, out: Optional[torch.Tensor] = None, views: Optional[List[List[ViewParams]]] = None) -> torch.Tensor:
    """(n_views * B, 3, size, size) fp32 on the device of ``batch.raw``, ordered [view 0 of every sample; view 1 ...] like
    ``torch.cat(views, 0)`` in the reference loop (scripts/phase5_big_run.py:1711).  ``views`` selects a subset of
    ``batch.views`` (e.g. the global views at one size, the local crops of the multi-crop extension at another)."""
    from . import ops
    from ._lib import check, lib
    raw = batch.raw
    ops._need_cuda(raw)
    assert raw.dtype in (torch.int16, torch.uint16) and raw.is_contiguous()
    rows_i, rows_f, max_crop = [], [], 1
    for vs in (batch.views if views is None else views):
        for i, p in enumerate(vs):
            H, W = batch.shapes[i]
            if not (0 <= p.top and p.top + p.h <= H and 0 <= p.left and p.left + p.w <= W and p.h > 0 and p.w > 0):
                raise ValueError(f"crop box {(p.top, p.left, p.h, p.w)} outside a {H}x{W} stack")
            rows_i.append((batch.offsets[i], H, W, p.top, p.left, p.h, p.w, int(p.flip)))
            rows_f.append((np.float32(p.level - p.width / 2.0), np.float32(max(p.width, 1.0))))
            max_crop = max(max_crop, p.h, p.w)
    V = len(rows_i)
    dev = raw.device
    vi = torch.tensor(rows_i, dtype=torch.int64).to(dev, non_blocking=True)
    vf = torch.tensor(np.asarray(rows_f, dtype=np.float32)).to(dev, non_blocking=True)
    if out is None:
        out = torch.empty((V, 3, size, size), dtype=torch.float32, device=dev)
    else:
        assert out.shape == (V, 3, size, size) and out.dtype == torch.float32 and out.is_contiguous()
    check(lib.dinox_slice_views(ops._p(raw), ops._p(vi), ops._p(vf), ops._p(out), V, size, max_crop, ops._stream()), "dinox_slice_views")
    return out
