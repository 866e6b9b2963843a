#!/usr/bin/env python3
"""Which host code issues device copies / framework kernels inside a training step?  (torch.profiler with python stacks.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda", 0)
wl = bench.Workload(dev, 0, B=int(os.environ.get("B", 64)))
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    wl.step()
    torch.cuda.synchronize()
ev = prof.events()
from collections import Counter
c = Counter()
for e in ev:
    n = e.name
    if ("Memcpy" in n or "Memset" in n or n.startswith("aten::copy_") or n.startswith("aten::fill_") or n.startswith("aten::add") or n.startswith("aten::zero_")
            or n.startswith("aten::mul") or n.startswith("aten::to") or n.startswith("aten::_to_copy") or n.startswith("aten::contiguous") or n.startswith("aten::clone")):
        st = [s for s in (e.stack or []) if "dino-x_amd" in s or "bench.py" in s]
        c[(n, st[0] if st else "?")] += 1
for (n, s), k in c.most_common(40):
    print(k, n, "|", s)
