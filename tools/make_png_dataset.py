#!/usr/bin/env python3
"""Writes a CT-like dataset of 16-bit HU PNG slices + the index.csv the training script reads (png_path, series_dir, slice_index,
encoding, spacing_x/y/z, dataset) -- for timing the real-data path of scripts/phase5_big_run.py where no dataset can be fetched.
Slices are smooth "anatomy" (low-frequency blobs, shifted a little from slice to slice) plus scanner-like noise, stored as
HU * 10 + 32768: they deflate about as badly as real CT (the decode cost is what the test is about).

  python tools/make_png_dataset.py OUT_DIR [--series 64] [--slices 64] [--size 512] [--workers 16]"""
import argparse
import csv
import os
from multiprocessing import Pool
from pathlib import Path

import numpy as np


def _series(job):
    from PIL import Image
    out, s, n_slices, size = job
    g = np.random.default_rng(1000 + s)
    d = Path(out) / f"series{s:04d}"
    d.mkdir(parents=True, exist_ok=True)
    coarse = g.normal(0, 1, (n_slices // 4 + 2, size // 32 + 2, size // 32 + 2)).astype(np.float32)
    rows = []
    sp_xy, sp_z = float(g.uniform(0.46, 0.98)), float(g.uniform(0.625, 5.0))
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / 32.0
    y0, x0 = yy.astype(np.int32), xx.astype(np.int32)
    fy, fx = yy - y0, xx - x0
    for z in range(n_slices):
        zc, fz = z // 4, (z % 4) / 4.0
        plane = (1 - fz) * coarse[zc] + fz * coarse[zc + 1]
        sm = ((1 - fy) * (1 - fx) * plane[y0, x0] + (1 - fy) * fx * plane[y0, x0 + 1] + fy * (1 - fx) * plane[y0 + 1, x0]
              + fy * fx * plane[y0 + 1, x0 + 1])
        hu = -300.0 + 500.0 * sm + g.normal(0, 25.0, (size, size))              # soft tissue / lung-ish range + noise
        body = (yy - size / 64.0) ** 2 + (xx - size / 64.0) ** 2 < (size / 64.0 * 0.9) ** 2
        hu = np.where(body, hu, -1000.0)                                         # air outside the body: the compressible part
        u16 = np.clip(np.rint(hu * 10.0 + 32768.0), 0, 65535).astype(np.uint16)
        p = d / f"slice{z:04d}.png"
        Image.fromarray(u16).save(p)
        rows.append(dict(png_path=str(p), series_dir=str(d), slice_index=z, encoding="hu16_png", spacing_x=sp_xy, spacing_y=sp_xy,
                         spacing_z=sp_z, dataset="synthetic_ct"))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out", type=Path)
    ap.add_argument("--series", type=int, default=64)
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    a = ap.parse_args()
    a.out.mkdir(parents=True, exist_ok=True)
    with Pool(a.workers) as pool:
        rows = [r for rs in pool.map(_series, [(str(a.out), s, a.slices, a.size) for s in range(a.series)]) for r in rs]
    with open(a.out / "index.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    nbytes = sum(os.path.getsize(r["png_path"]) for r in rows)
    print(f"{len(rows)} slices of {a.size}x{a.size} in {a.series} series under {a.out}: {nbytes / 1e6:.0f} MB of PNG, "
          f"{len(rows) * a.size * a.size * 2 / 1e6:.0f} MB decoded; index {a.out / 'index.csv'}")


if __name__ == "__main__":
    main()
