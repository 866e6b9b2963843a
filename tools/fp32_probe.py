#!/usr/bin/env python3
"""fp32 (no --amp) step of this engine at a given batch size.  The comparison with plain PyTorch fp32 on the same GPU is `python bench.py --fp32
--batch-size B` (its framework_baseline leg runs the restated reference loop without autocast)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 64))
wl = bench.Workload(dev, 0, B=B, fp32=True)
for _ in range(2): wl.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): wl.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f"dinox fp32 mode, bs {B}: {dt*1e3:.1f} ms/step = {B/dt:.0f} samples/s")
