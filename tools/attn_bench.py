#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the hot-path shape (ViT-S/16 bs256: 512 views x 6 heads x 201 tokens x 64).
HIP-event timing, interleaved rounds, random data.  DINOX_ATTN_BWD_SPLIT=1 times the two-kernel backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops

V, N, H = int(os.environ.get("V", 512)), int(os.environ.get("N", 201)), int(os.environ.get("H", 6))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(V, N, 3 * H * 64, device=dev, generator=g) * 0.7).bfloat16()
do = (torch.randn(V, N, H * 64, device=dev, generator=g) * 0.1).bfloat16()
o, lse = ops.attention_fwd(qkv, H)
cases = {"fwd": lambda: ops.attention_fwd(qkv, H), "bwd": lambda: ops.attention_bwd(do, qkv, o, lse, H)}
for f in cases.values():
    for _ in range(3): f()
torch.cuda.synchronize()
times = {n: [] for n in cases}
for r in range(int(os.environ.get("ROUNDS", 10))):
    for n, f in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        times[n].append((e0, e1))
torch.cuda.synchronize()
el = V * N * H * 64 * 2
byt = {"fwd": 4 * el, "bwd": 8 * el}          # q,k,v in + o out ; q,k,v,dO,O in + dq,dk,dv out
for n in cases:
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in times[n])
    print(f"attention {n} V={V} N={N} h={H}: med {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us   {byt[n] / ts[len(ts)//2] / 1e6:5.2f} TB/s (every tensor once)")
