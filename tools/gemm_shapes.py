#!/usr/bin/env python3
"""Every dinox_gemm launch of two bench steps, timed one by one (GemmTimer(every=1)) and listed per (kernel, shape, epilogue), slowest
family first: where the products of a step stand (the head's dX at K = 8192 on twelve tiles, the Gram products, ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch, bench
from dinox import ops
dev = torch.device("cuda:0")
kw = dict(B=int(os.environ.get("B", 256)))
if os.environ.get("MODEL"):
    kw["model"] = os.environ["MODEL"]
wl = bench.Workload(dev, 0, **kw)
for _ in range(3): wl.step()
torch.cuda.synchronize()
t = ops.GemmTimer(every=1)
with t:
    for _ in range(2): wl.step()
    torch.cuda.synchronize()
rows = [l.split() for l in t.text.splitlines()]
rows = [r for r in rows if len(r) == 13]
rows.sort(key=lambda r: -float(r[12]))
print("kernel M N K batch epi in out aux sharedb launches timed ms_timed  us/launch")
for r in rows[:40]:
    print(" ".join(r), f" {1e3 * float(r[12]) / max(int(r[11]), 1):.1f}")
