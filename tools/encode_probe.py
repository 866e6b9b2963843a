#!/usr/bin/env python3
"""Latency of zoo.encode.encode (one image) and throughput of encode_batch on ViT-S/16 224 (the inference surface of the drop-in)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import numpy as np, torch
import zoo.arch as arch
from zoo.encode import encode, encode_batch, preprocess
m = arch.PatchViT(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True).to("cuda").eval()
r = np.random.default_rng(0)
img = (r.standard_normal((512, 512)) * 300).astype(np.float32)
for _ in range(3): encode(m, img, pixel_spacing=(0.7, 0.7), slice_thickness=2.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): f = encode(m, img, pixel_spacing=(0.7, 0.7), slice_thickness=2.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
t1 = time.perf_counter()
for _ in range(20): x = preprocess(img, 224, "hu_float", 40.0, 400.0)
tp = (time.perf_counter() - t1) / 20
print(f"encode(1 x 512x512): {dt*1e3:.2f} ms per call (host preprocess alone {tp*1e3:.2f} ms)")
imgs = [img] * 64; sps = [(0.7, 0.7, 2.0)] * 64
for _ in range(2): encode_batch(m, imgs, sps)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): encode_batch(m, imgs, sps)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
x = torch.randn(64, 3, 224, 224, device="cuda"); sp = torch.rand(64, 3, device="cuda") + 0.5
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3): m(x, spacing=sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x, spacing=sp)
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 10
    m32 = None
with torch.no_grad():
    for _ in range(3): m(x, spacing=sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x, spacing=sp)
    torch.cuda.synchronize(); d32 = (time.perf_counter() - t0) / 10
print(f"encode_batch(64): {dt*1e3:.1f} ms = {64/dt:.0f} img/s;  forward alone on 64 device-resident images: bf16 {df*1e3:.2f} ms, fp32 {d32*1e3:.2f} ms")
