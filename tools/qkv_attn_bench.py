#!/usr/bin/env python3
"""The fused qkv-projection + attention forward (dinox_qkv_attention_fwd) against the two launches it replaces (dinox_gemm for qkv, then
dinox_attention_fwd), interleaved in one process, HIP events.  MODEL=S|L, B=<images> (default: the bench's 512 views / 256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops

dev = "cuda"
model = os.environ.get("MODEL", "S")
D, heads, B = {"S": (384, 6, 512), "B": (768, 12, 256), "L": (1024, 16, 256)}[model]
B = int(os.environ.get("B", B))
N = int(os.environ.get("N", 201))
C = heads * 64
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(B, N, D, device=dev, generator=g) * 0.7).bfloat16()
w = (torch.randn(3 * C, D, device=dev, generator=g) * (1.5 / D ** 0.5)).bfloat16()
b = torch.randn(3 * C, device=dev, generator=g) * 0.2
variants = {
    "composed: gemm + attention": lambda: ops.attention_fwd(ops.gemm(x.view(B * N, D), w, bias=b, out_dtype=torch.bfloat16).view(B, N, 3 * C), heads),
    "gemm alone": lambda: ops.gemm(x.view(B * N, D), w, bias=b, out_dtype=torch.bfloat16),
    "fused": lambda: ops.qkv_attention(x, w, b, heads),
    "fused + qkv out + lse": lambda: ops.qkv_attention(x, w, b, heads, want_qkv=True, want_lse=True),
}
def _dbg(flag):
    def f():
        os.environ["DINOX_QA_DBG"] = flag
        try:
            return ops.qkv_attention(x, w, b, heads)
        finally:
            os.environ.pop("DINOX_QA_DBG", None)
    return f
if os.environ.get("SPLIT"):
    variants.update({"fused, no attention arithmetic": _dbg("1"), "fused, one ring round of K only": _dbg("2"), "fused, neither": _dbg("3")})
for f in variants.values():
    for _ in range(3):
        f()
torch.cuda.synchronize()
R = int(os.environ.get("ROUNDS", 10))
ts = {k: [] for k in variants}
for r in range(R):
    for k, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        ts[k].append((e0, e1))
torch.cuda.synchronize()
fl = 2 * B * N * D * 3 * C + 4 * B * heads * N * N * 64
print(f"model {model}: B {B} N {N} heads {heads} D {D}; {fl / 1e9:.1f} GFLOP (projection + attention)")
for k, v in ts.items():
    t = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in v)
    print(f"  {k:30s} median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us")
