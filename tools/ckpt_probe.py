#!/usr/bin/env python3
"""Step time of the headline workload with --grad-checkpoint (one more student forward inside backward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda", 0)
for ck in (False, True):
    wl = bench.Workload(dev, 0, B=256)
    for m in (wl.eng.student.backbone, wl.eng.teacher.backbone):
        m.use_grad_checkpoint = ck
    wl.eng.student.train()
    for _ in range(4): wl.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): wl.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"grad_checkpoint {ck}: {dt*1e3:.3f} ms/step  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    wl = None; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
