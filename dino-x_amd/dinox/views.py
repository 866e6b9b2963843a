"""Device-side view pipeline for the 2.5D slice stacks.

The reference builds every training view on DataLoader workers (scripts/phase5_big_run.py:493-497, 516-528, 549-555): HU
decode, random window, RandomResizedCrop (bicubic), flip, normalise -- ~500 img/s on its box.  Here the workers only decode
PNGs to u16 stacks; the random draws happen on the host (cheap scalars, same order and distributions as the reference's
transform stack) and ONE kernel (csrc/views.hip, ``dinox_slice_views``) turns the resident u16 stacks into the fp32
``(2B, 3, S, S)`` batch that ``PatchViT.forward`` takes.
"""
from __future__ import annotations

import math
import os
import random
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import collections

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


_mapped: "collections.OrderedDict" = collections.OrderedDict()      # shared ring buffers of the loader workers seen by this process


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

    def pin_memory(self) -> "StackBatch":
        """Called by the DataLoader's pinning thread (pin_memory=True): page-locked memory, so that ``to(device)`` is a true
        asynchronous DMA instead of a staged copy that blocks the host.  With the workers' shared-memory ring (SHM_RING > 0) the ring
        buffer ITSELF is page-locked once (hipHostRegister on this process's mapping of it) and handed on as it is: no second host
        copy of the batch (400 MB per 256 stacks of 512 x 512 -- at 25 batches a second more than one thread can copy).  The
        DevicePrefetcher bounds the copies in flight so that a worker never rewrites a buffer the DMA engine still reads."""
        if SHM_RING > 0 and self.raw.is_shared():      # (ring only: with a fresh shared tensor per batch this would pin them all down)
            # keep the worker's ring buffer MAPPED in this process: torch finds a shared storage it already holds by its file identity,
            # a dropped one is unmapped and mapped again for the next batch in it (120 MB of page-table faults per batch)
            st = self.raw.untyped_storage()
            key = st.data_ptr()
            hit = _mapped.get(key)
            if hit is None or hit[0].nbytes() != st.nbytes():
                if hit is not None:
                    _unregister(hit)
                hit = _mapped[key] = (st, _register(st))
            _mapped.move_to_end(key)
            while len(_mapped) > 128:
                _unregister(_mapped.popitem(last=False)[1])
            if hit[1] and self.raw.is_pinned():
                return StackBatch(self.raw, self.offsets, self.shapes, self.views, self.spacing.pin_memory())
        return StackBatch(self.raw.pin_memory(), self.offsets, self.shapes, self.views, self.spacing.pin_memory())


def _register(st) -> bool:
    """Page-lock this process's mapping of a shared storage (False: not possible here -- the caller copies to pinned memory instead)."""
    if os.environ.get("DINOX_PIN_COPY") or not torch.cuda.is_available():
        return False
    try:
        return int(torch.cuda.cudart().cudaHostRegister(st.data_ptr(), st.nbytes(), 0)) == 0
    except Exception:
        return False


def _unregister(entry) -> None:
    st, registered = entry
    if registered:
        try:
            torch.cuda.cudart().cudaHostUnregister(st.data_ptr())
        except Exception:
            pass


class DevicePrefetcher:
    """Keeps ONE batch ahead on the device: while step s runs, the raw stacks of step s + 1 cross PCIe on a copy stream of their own
    (120 MB per 256 stacks of 3 x 280 x 280; 400 MB at 512 x 512).  ``next()`` hands out a batch whose copy the current stream has
    been told to wait for, and starts the copy of the one after.  Restarts the loader at the end of an epoch (the training loop is
    step-driven, as in the reference)."""

    def __init__(self, loader, device, ahead: bool = True) -> None:
        self.loader, self.device, self.run_ahead = loader, torch.device(device), ahead
        self.it = iter(loader)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.ahead = None
        self._inflight: "collections.deque" = collections.deque()       # events of the copies issued last (see _start)
        self.timing = None           # a list -> (start, end) events of every copy (the training script's DINOX_CLI_PROFILE)
        # (the loader iterator is created HERE -- that draws the loader's base seed from torch's global generator, and the training script
        #  creates it at the same point of its start-up as the reference creates its iterator; the first batch is fetched on first use)

    def _host_next(self):
        try:
            return next(self.it)
        except StopIteration:
            self.it = iter(self.loader)
            return next(self.it)

    def _start(self) -> None:
        item = self._host_next()
        if self.stream is None:
            self.ahead = (item.to(self.device), None)
            return
        # At most two copies in flight: the batch may still sit in a loader worker's ring buffer (StackBatch.pin_memory), and the worker
        # gets that buffer back SHM_RING of ITS batches later -- by then at least two more of its batches have been handed out here,
        # i.e. this wait has covered the copy that read it.  (The copy stream holds nothing but these copies: the wait is short.)
        while len(self._inflight) >= 2:
            self._inflight.popleft().synchronize()
        with torch.cuda.stream(self.stream):
            if self.timing is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            dev = item.to(self.device, non_blocking=True)
            ev = torch.cuda.Event(enable_timing=self.timing is not None)
            ev.record(self.stream)
            if self.timing is not None:
                self.timing.append((e0, ev))
        self._inflight.append(ev)
        self.ahead = (dev, ev)

    def next(self) -> StackBatch:
        if not self.run_ahead:                           # (A/B and the draw-order test: fetch, copy and hand out in program order)
            dev = self._host_next().to(self.device)
            if self.device.type == "cuda":               # the same bound on copies in flight (here they queue behind the compute)
                ev = torch.cuda.Event()
                ev.record()
                self._inflight.append(ev)
                while len(self._inflight) > 2:
                    self._inflight.popleft().synchronize()
            return dev
        if self.ahead is None:
            self._start()
        dev, ev = self.ahead
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            dev.raw.record_stream(cur)                 # allocated on the copy stream, consumed on this one
            dev.spacing.record_stream(cur)
        self._start()
        return dev


# Inside a DataLoader worker the batch buffer is a SHARED-MEMORY tensor taken from a small per-worker ring and written in one pass
# (slice by slice): first-touch page faults dominate the host side of this pipeline (120 MB of fresh pages cost ~0.6 s in this sandbox,
# 15 ms once touched), and torch re-sends a shared storage it has sent before by handle.  Reuse is safe under the DataLoader's own
# flow control IF the consumer copies the batch out before asking for more -- which pin_memory=True does (the pinning thread copies every
# batch to page-locked memory before the loop sees it); the training script switches the ring on only then (SHM_RING > 0).
SHM_RING = 0                 # buffers per worker (0: a fresh shared tensor per batch); needs > prefetch_factor + 1
_ring: List[torch.Tensor] = []
_ring_pos = 0


def _batch_buffer(total: int) -> torch.Tensor:
    global _ring_pos
    if torch.utils.data.get_worker_info() is None:
        return torch.empty(total, dtype=torch.int16)
    if SHM_RING <= 0:
        return torch.empty(0, dtype=torch.int16).set_(torch.UntypedStorage._new_shared(2 * total), 0, (total,))
    if len(_ring) < SHM_RING:
        _ring.append(torch.empty(0, dtype=torch.int16))
    k = _ring_pos % SHM_RING
    _ring_pos += 1
    if _ring[k].numel() < total:
        cap = int(total * 1.25) + 1024                                     # ragged batches: grow rarely
        _ring[k] = torch.empty(0, dtype=torch.int16).set_(torch.UntypedStorage._new_shared(2 * cap), 0, (cap,))
    return _ring[k][:total]


def collate_stacks(items: Sequence[Tuple[object, Sequence[ViewParams], torch.Tensor]]) -> StackBatch:
    """DataLoader collate_fn: items are (stack, [ViewParams per view], spacing (3,)) with ``stack`` a u16 array (3,H,W) or the three
    (H,W) slices as they come out of the decoder (no intermediate np.stack)."""
    offsets, shapes, total = [], [], 0
    for stack, _, _ in items:
        H, W = (stack.shape[1], stack.shape[2]) if isinstance(stack, np.ndarray) else stack[0].shape
        if isinstance(stack, np.ndarray):
            assert stack.ndim == 3 and stack.shape[0] == 3, stack.shape
        else:
            assert len(stack) == 3 and all(sl.shape == (H, W) for sl in stack), [getattr(sl, "shape", None) for sl in stack]
        offsets.append(total)
        shapes.append((H, W))
        total += 3 * H * W
    raw = _batch_buffer(total)
    dst = raw.numpy().view(np.uint16)
    for off, (stack, _, _), (H, W) in zip(offsets, items, shapes):
        if isinstance(stack, np.ndarray):
            dst[off:off + 3 * H * W] = np.asarray(stack, dtype=np.uint16).reshape(-1)      # bit pattern; the kernel reads it as u16
        else:
            for c, sl in enumerate(stack):
                dst[off + c * H * W:off + (c + 1) * H * W] = np.asarray(sl, dtype=np.uint16).reshape(-1)
    n_views = len(items[0][1])
    views = [[it[1][k] for it in items] for k in range(n_views)]
    return StackBatch(raw, offsets, shapes, views, torch.stack([it[2] for it in items], 0))


def make_views(batch: StackBatch, size: int, out: Optional[torch.Tensor] = None, views: Optional[List[List[ViewParams]]] = None,
               patch: Optional[int] = None, operand_dtype: Optional[torch.dtype] = None):
    """(n_views * B, 3, size, size) fp32 on the device of ``batch.raw``, ordered [view 0 of every sample; view 1 ...] like
    ``torch.cat(views, 0)`` in the reference loop (scripts/phase5_big_run.py:1711).  ``views`` selects a subset of
    ``batch.views`` (e.g. the global views at one size, the local crops of the multi-crop extension at another).
    With ``patch`` (and ``operand_dtype``: torch.bfloat16 under --amp, else torch.float32) the result is an ``ops.PatchOperand``: the
    same views written directly as the patch-embed operand [V * (size/patch)^2, patch_cols] -- the image batch is never stored."""
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
    # page-locked staging: a copy from pageable memory is synchronous -- it waits for the stream to drain, i.e. the host could never
    # run ahead of the device (measured in the training script: 49 ms per step instead of the engine's 38.6)
    vi = torch.tensor(rows_i, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
    vf = torch.tensor(np.asarray(rows_f, dtype=np.float32)).pin_memory().to(dev, non_blocking=True)
    if patch is not None:
        dt = operand_dtype or torch.float32
        if size % patch:
            raise ValueError(f"view size {size} is not a multiple of the patch size {patch}")
        cols = ops.patch_cols(patch, dt)
        u = out if out is not None else torch.empty((V * (size // patch) ** 2, cols), dtype=dt, device=dev)
        assert u.shape == (V * (size // patch) ** 2, cols) and u.dtype == dt and u.is_contiguous()
        check(lib.dinox_slice_views_patches(ops._p(raw), ops._p(vi), ops._p(vf), ops._p(u), V, size, max_crop, patch, cols, ops._code(dt),
                                            ops._stream()), "dinox_slice_views_patches")
        return ops.PatchOperand(u, V, size, patch)
    if out is None:
        out = torch.empty((V, 3, size, size), dtype=torch.float32, device=dev)
    else:
        assert out.shape == (V, 3, size, size) and out.dtype == torch.float32 and out.is_contiguous()
    check(lib.dinox_slice_views(ops._p(raw), ops._p(vi), ops._p(vf), ops._p(out), V, size, max_crop, ops._stream()), "dinox_slice_views")
    return out
