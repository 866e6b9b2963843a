#!/usr/bin/env python3
"""Step time at other widths of the preset table (ViT-B/16: 768 x 12 heads; ViT-S is the tuned one): does anything fall off a cliff?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
import zoo.arch as arch
from dinox.engine import StepHyperParams, TrainEngine
dev = torch.device("cuda", 0)
CASES = (("ViT-S/16", 384, 12, 6, 128), ("ViT-B/16", 768, 12, 12, 128), ("ViT-S/14 (261 tokens)", 384, 12, 6, 128),
         ("ViT-T/14 (192 x 3 heads)", 192, 12, 3, 128), ("ViT-L/14, 6 of 24 blocks", 1024, 6, 16, 64), ("ViT-g/14 (1408 x 16 heads of 88), 4 of 40 blocks", 1408, 4, 16, 32),
         ("ViT-S/16 at 448 px (789 tokens)", 384, 12, 6, 32))
sel = os.environ.get("CASES")
for name, dim, depth, heads, B in [c for c in CASES if not sel or any(x in c[0] for x in sel.split(","))]:
    patch = 14 if "/14" in name else 16
    img = 448 if "448" in name else 224
    kw = dict(img_size=img, patch=patch, dim=dim, depth=depth, heads=heads, num_registers=4, scale_aware=True)
    torch.manual_seed(0)
    s = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192); t = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
    t.load_state_dict(s.state_dict())
    eng = TrainEngine(s.to(dev), t.to(dev), 8192, StepHyperParams(max_steps=100, warmup_steps=5), amp_dtype=torch.bfloat16)
    x = torch.randn(2 * B, 3, img, img, device=dev); sp = torch.rand(2 * B, 3, device=dev) + 0.5
    for _ in range(3): eng.step(x, sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): eng.step(x, sp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    P = (img // patch) ** 2; N = P + 5
    f = 2 * (P * 3 * patch * patch * dim + depth * (N * dim * 3 * dim + 2 * N * N * dim + N * dim * dim + 2 * N * dim * 4 * dim) + dim * dim + dim * 8192 + (N - 1) ** 2 * dim)
    print(f"{name:24s} bs {B}: {dt*1e3:7.2f} ms/step = {B/dt:7.0f} samples/s = {B/dt*8*f/1e12:6.1f} TFLOP/s")
    del eng, s, t; torch.cuda.empty_cache()
