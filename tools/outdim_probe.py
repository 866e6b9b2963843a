#!/usr/bin/env python3
"""The DINO head at other prototype counts (the reference's presets use 8192; DINO's own default is 65536)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
import zoo.arch as arch
from dinox.engine import StepHyperParams, TrainEngine
dev = torch.device("cuda", 0)
B = 256
for out_dim, koleo in ((8192, 0.0), (65536, 0.0), (65536, 0.1)):
    kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
    torch.manual_seed(0)
    s = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim); t = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim)
    t.load_state_dict(s.state_dict())
    eng = TrainEngine(s.to(dev), t.to(dev), out_dim, StepHyperParams(max_steps=100, warmup_steps=5, koleo_weight=koleo), amp_dtype=torch.bfloat16)
    x = torch.randn(2 * B, 3, 224, 224, device=dev); sp = torch.rand(2 * B, 3, device=dev) + 0.5
    for _ in range(3): eng.step(x, sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): eng.step(x, sp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"out_dim {out_dim:6d} koleo {koleo}: {dt*1e3:7.2f} ms/step = {B/dt:6.0f} samples/s  loss {float(eng.last['loss']):.4f}")
    del eng, s, t; torch.cuda.empty_cache()
