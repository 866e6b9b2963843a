#!/usr/bin/env python3
"""Host-side enqueue time of one engine step (no synchronisation inside the timed call) vs the GPU time of the step.
Each timed step starts on an EMPTY queue (device synchronised first), so the figure is what the host needs to issue a step, not how
long the launch queue made it wait; DINOX_BLOCK_NATIVE=0 (block launches issued one by one from Python) for the A/B.  PROFILE=1
adds a cProfile listing of one step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
import zoo.arch as arch
from dinox import ops
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
for native in (True, False, True, False):
    ops._BLOCK_NATIVE = native
    for _ in range(3): eng.step(x, sp)
    torch.cuda.synchronize()
    host, wall = [], []
    for _ in range(6):
        a = time.perf_counter(); eng.step(x, sp); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
        host.append(b - a); wall.append(c - a)
    print(f"B={B} native={native}: host enqueue per step (empty queue) {1e3 * sorted(host)[len(host) // 2]:.2f} ms (min {1e3 * min(host):.2f}); "
          f"step wall {1e3 * sorted(wall)[len(wall) // 2]:.2f} ms", flush=True)
if os.environ.get("PROFILE"):
    import cProfile, pstats
    ops._BLOCK_NATIVE = True
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); eng.step(x, sp); pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
