#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the hot-path shape (B=512 views, N=201, 6 heads, d=64, bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
B, N, H, D = int(os.environ.get("B", 512)), int(os.environ.get("N", 201)), 6, 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * H * D, device="cuda", generator=g).bfloat16()
do = torch.randn(B, N, H * D, device="cuda", generator=g).bfloat16()
o, lse = ops.attention_fwd(qkv, H)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]
f = 4.0 * B * H * N * N * D
m, lo = t(lambda: ops.attention_fwd(qkv, H)); print(f"attn fwd  med {m:7.1f} us min {lo:7.1f} us  {f / m / 1e6:6.1f} TFLOP/s (alg)")
m, lo = t(lambda: ops.attention_bwd(do, qkv, o, lse, H)); print(f"attn bwd  med {m:7.1f} us min {lo:7.1f} us  {2.5 * f / m / 1e6:6.1f} TFLOP/s (alg, 5 products)")
