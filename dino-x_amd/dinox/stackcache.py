"""Decoded-slice cache for the training script's PNG path (not in the reference).

The reference decodes three 16-bit PNGs per sample per epoch inside its DataLoader workers
(scripts/phase5_big_run.py:516-528 ``_load_hu01`` + :539-547): ~2-3 ms of zlib inflate per 512 x 512 slice, i.e. about a thousand
samples a second on the 16 host cores of a one-GPU box -- a sixth of what the MI355X step consumes.  A decoded slice is a pure
function of its file, so it is decoded ONCE and kept as raw uint16 in one memory-mapped file:

  <root>/<key>/data.u16   every slice back to back (C order, its own H x W; offsets in meta.json); a sparse file that fills lazily
  <root>/<key>/valid.u8   one byte per slice: 1 once its pixels are in data.u16 (written AFTER the pixels)
  <root>/<key>/meta.json  paths, shapes, offsets, the key's ingredients

``key`` = sha256 over (path, file size, mtime_ns) of every PNG of the index: a changed, replaced or re-ordered dataset gets a fresh
cache, never stale pixels.  DataLoader workers share the mapping through the page cache; a slice decoded by two workers at once is
written twice with identical bytes (harmless).  After the first epoch (or ``prefill``) a worker's ``_read`` is a page-cache memcpy
straight into the batch buffer (views.collate_stacks), and the loader keeps up with the device."""
from __future__ import annotations

import hashlib
import json
import os
import struct
import zlib
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def png_shape(path) -> Tuple[int, int]:
    """(H, W) from the IHDR chunk: 24 bytes of I/O, no decode."""
    with open(path, "rb") as f:
        head = f.read(24)
    if len(head) < 24 or head[:8] != _PNG_SIG or head[12:16] != b"IHDR":
        raise ValueError(f"{path}: not a PNG file")
    w, h = struct.unpack(">II", head[16:24])
    return int(h), int(w)


def decode_png_u16(path) -> np.ndarray:
    """The decode the dataset does (PngDataset._read): first channel if the file has several, dtype as stored."""
    from PIL import Image
    arr = np.array(Image.open(path))
    return arr[:, :, 0] if arr.ndim == 3 else arr


class SliceCache:
    """uint16 memmap of decoded slices, one entry per distinct PNG file of an index."""

    def __init__(self, paths: Sequence, root, create: bool = True) -> None:
        self.paths: List[str] = sorted({str(p) for p in paths})
        self.index: Dict[str, int] = {p: i for i, p in enumerate(self.paths)}
        h = hashlib.sha256()
        stats = []
        for p in self.paths:
            st = os.stat(p)
            stats.append((st.st_size, st.st_mtime_ns))
            h.update(f"{p}|{st.st_size}|{st.st_mtime_ns}\n".encode())
        self.key = h.hexdigest()[:20]
        self.dir = Path(root) / self.key
        meta_path = self.dir / "meta.json"
        if meta_path.exists():
            meta = json.loads(meta_path.read_text())
            if meta.get("paths") != self.paths:
                raise RuntimeError(f"{meta_path}: key collision (different file list under the same key)")
            self.shapes = [tuple(s) for s in meta["shapes"]]
        else:
            if not create:
                raise FileNotFoundError(meta_path)
            self.shapes = [png_shape(p) for p in self.paths]
            self.dir.mkdir(parents=True, exist_ok=True)
            tmp = self.dir / f"meta.json.{os.getpid()}.tmp"
            tmp.write_text(json.dumps({"version": 1, "paths": self.paths, "shapes": [list(s) for s in self.shapes],
                                       "stats": [list(s) for s in stats]}))
            os.replace(tmp, meta_path)                   # (several ranks may race here: each writes the same bytes, rename is atomic)
        sizes = np.array([h_ * w_ for h_, w_ in self.shapes], dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.total = int(self.offsets[-1])
        self._ensure_file(self.dir / "data.u16", 2 * max(self.total, 1))
        self._ensure_file(self.dir / "valid.u8", max(len(self.paths), 1))
        self._data: Optional[np.memmap] = None           # opened per process (after the DataLoader fork)
        self._valid: Optional[np.memmap] = None
        self._pid = -1
        self.hits = self.misses = 0

    @staticmethod
    def _ensure_file(path: Path, nbytes: int) -> None:
        fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
        try:
            if os.fstat(fd).st_size < nbytes:
                os.ftruncate(fd, nbytes)                 # sparse: blocks appear as slices are written
        finally:
            os.close(fd)

    def _maps(self):
        if self._pid != os.getpid():
            self._data = np.memmap(self.dir / "data.u16", dtype=np.uint16, mode="r+", shape=(max(self.total, 1),))
            self._valid = np.memmap(self.dir / "valid.u8", dtype=np.uint8, mode="r+", shape=(max(len(self.paths), 1),))
            self._pid = os.getpid()
        return self._data, self._valid

    def __len__(self) -> int:
        return len(self.paths)

    def __getstate__(self):                              # (spawned workers: re-open the maps there)
        d = dict(self.__dict__)
        d["_data"] = d["_valid"] = None
        d["_pid"] = -1
        return d

    def get(self, path) -> np.ndarray:
        """The decoded (H, W) uint16 slice of ``path``: a view of the mapping (read-only use), decoding and storing it on first touch."""
        i = self.index[str(path)]
        data, valid = self._maps()
        H, W = self.shapes[i]
        view = data[self.offsets[i]:self.offsets[i + 1]].reshape(H, W)
        if valid[i]:
            self.hits += 1
            return view
        arr = decode_png_u16(self.paths[i])
        if arr.shape != (H, W):
            raise ValueError(f"{self.paths[i]}: decoded {arr.shape}, header said {(H, W)}")
        if arr.dtype != np.uint16:                       # an 8-bit or signed file is not HU-encoded u16: keep its VALUES, as np.asarray(.., uint16) would
            arr = arr.astype(np.uint16)
        view[...] = arr
        valid[i] = 1                                     # after the pixels (x86 stores are not reordered; a reader that sees 1 sees the slice)
        self.misses += 1
        return view

    def filled(self) -> int:
        return int(np.count_nonzero(self._maps()[1][:len(self.paths)]))

    def checksum(self, i: int) -> int:
        """crc32 of slice i's cached pixels (tests, ``verify``)."""
        data, _ = self._maps()
        return zlib.crc32(np.ascontiguousarray(data[self.offsets[i]:self.offsets[i + 1]]).tobytes())

    def prefill(self, workers: int = 0, say=None) -> int:
        """Decode every slice that is not cached yet (``workers`` processes; 0 = in this one).  Returns how many were decoded."""
        _, valid = self._maps()
        todo = [i for i in range(len(self.paths)) if not valid[i]]
        if not todo:
            return 0
        if workers <= 1:
            for i in todo:
                self.get(self.paths[i])
        else:
            import multiprocessing as mp
            chunks = [todo[k::workers] for k in range(workers)]
            with mp.get_context("fork").Pool(workers) as pool:
                for n, _ in enumerate(pool.imap_unordered(_fill_chunk, [(self, c) for c in chunks])):
                    if say:
                        say(f"stack_cache prefill: {n + 1}/{workers} worker lists done")
        self._maps()[0].flush()
        self._maps()[1].flush()
        return len(todo)


def _fill_chunk(job) -> int:
    cache, idxs = job
    for i in idxs:
        cache.get(cache.paths[i])
    return len(idxs)
