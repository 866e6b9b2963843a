#!/usr/bin/env python3
"""Host-side enqueue time of one engine step (no synchronisation inside the timed call) vs the GPU time of the step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
import zoo.arch as arch
from dinox.engine import StepHyperParams, TrainEngine
from dinox.hostinfo import usable_cpus
torch.set_num_threads(usable_cpus())
B = int(os.environ.get("B", 256))
kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
torch.manual_seed(0)
s = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192); t = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
t.load_state_dict(s.state_dict())
eng = TrainEngine(s.cuda(), t.cuda(), 8192, StepHyperParams(), amp_dtype=torch.bfloat16)
x = torch.randn(2 * B, 3, 224, 224, device="cuda"); sp = torch.rand(2 * B, 3, device="cuda") + 0.5
for _ in range(3): eng.step(x, sp)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(8):
    a = time.perf_counter(); eng.step(x, sp); host.append(time.perf_counter() - a)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 8
print(f"B={B}: host enqueue per step {1e3 * sorted(host)[len(host) // 2]:.1f} ms (min {1e3 * min(host):.1f}); wall per step {1e3 * wall:.1f} ms")
