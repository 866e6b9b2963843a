#!/usr/bin/env python3
"""Throughput of the device-side view pipeline (dinox_slice_views) at the headline batch: 256 stacks of 3x512x512 u16 ->
512 views of 3x224x224 fp32, reference draw distributions."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox.views import StackBatch, draw_view, make_views

B, H, W, S = 256, 512, 512, 224
g = torch.Generator().manual_seed(0)
raw = torch.randint(22768, 42768, (B, 3, H, W), generator=g, dtype=torch.int32).to(torch.int16).reshape(-1).cuda()
random.seed(0)
views = [[draw_view(H, W) for _ in range(B)] for _ in range(2)]
sb = StackBatch(raw, [i * 3 * H * W for i in range(B)], [(H, W)] * B, views, torch.ones(B, 3))
out = torch.empty(2 * B, 3, S, S, device="cuda")
for _ in range(3): make_views(sb, S, out)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); make_views(sb, S, out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
crop_px = sum(p.h * p.w for vs in views for p in vs)
rd, wr = crop_px * 3 * 2, out.numel() * 4
print(f"slice_views: {2*B} views in med {ts[5]:.3f} ms (min {ts[0]:.3f}) incl. the host-side table upload = {2*B/ts[5]*1e3:,.0f} views/s; "
      f"algorithmic {rd/1e6:.0f} MB read + {wr/1e6:.0f} MB written = {(rd+wr)/ts[5]/1e6:.0f} GB/s")
