#!/usr/bin/env python3
"""Step time of the headline workload with the KoLeo regulariser on (every production run of the reference uses --koleo-weight 0.1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
sys.argv = ["bench.py"]
import bench
from dinox.engine import StepHyperParams
dev = torch.device("cuda", 0)
for kw in ((0.1,) if os.environ.get("KOLEO_ONLY") else (0.0, 0.1)):
    wl = bench.Workload(dev, 0, B=256)
    wl.eng.hp.koleo_weight = kw
    for _ in range(4): wl.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): wl.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"koleo_weight {kw}: {dt*1e3:.3f} ms/step, koleo = {float(wl.eng.last['koleo']):.4f}")
    wl = None; torch.cuda.empty_cache()
