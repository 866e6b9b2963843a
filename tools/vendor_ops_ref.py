#!/usr/bin/env python3
"""Framework kernels (torch 2.10 / ROCm 7: SDPA, layer_norm) on the hot-path shapes beside this library's -- a yardstick for DESIGN.md, not
a code path of the product.  ViT-S/16 bs 256: 512 views x 201 tokens x 384 (6 heads of 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
import torch.nn.functional as F
from dinox import ops
V, N, Hh, D = 512, 201, 6, 384
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(V, N, 3 * D, device=dev, generator=g) * 0.7).bfloat16()
do = (torch.randn(V, N, D, device=dev, generator=g) * 0.1).bfloat16()
q, k, v = (t.reshape(V, N, Hh, 64).transpose(1, 2).contiguous().requires_grad_(True) for t in qkv.split(D, dim=-1))
dot = do.reshape(V, N, Hh, 64).transpose(1, 2).contiguous()
o_ours, lse = ops.attention_fwd(qkv, Hh)
x = torch.randn(V * N, D, device=dev, generator=g)
w, b = torch.randn(D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
dy = torch.randn(V * N, D, device=dev, generator=g).bfloat16()
xr = x.clone().requires_grad_(True)
wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
_, mean, rstd = ops.layernorm_fwd(x, w, b, torch.bfloat16)

def sdpa_fwd():
    with torch.no_grad():
        return F.scaled_dot_product_attention(q, k, v)
def sdpa_fwd_bwd():
    for t in (q, k, v): t.grad = None
    F.scaled_dot_product_attention(q, k, v).backward(dot)
def ln_fwd_t():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return F.layer_norm(x, (D,), w, b)
def ln_fwd_bwd_t():
    for t in (xr, wr, br): t.grad = None
    F.layer_norm(xr, (D,), wr, br).backward(dy.float())
cases = [
    ("attention forward", sdpa_fwd, lambda: ops.attention_fwd(qkv, Hh)),
    ("attention forward + backward", sdpa_fwd_bwd, lambda: (ops.attention_fwd(qkv, Hh), ops.attention_bwd(do, qkv, o_ours, lse, Hh))),
    ("LayerNorm forward (fp32 in)", ln_fwd_t, lambda: ops.layernorm_fwd(x, w, b, torch.bfloat16)),
    ("LayerNorm forward + backward", ln_fwd_bwd_t, lambda: (ops.layernorm_fwd(x, w, b, torch.bfloat16), ops.layernorm_bwd(dy, x, w, mean, rstd, want_lowp=True))),
]
R = int(os.environ.get("ROUNDS", 10))
for name, lib, ours in cases:
    out = {}
    for tag, fn in (("framework", lib), ("dinox", ours)):
        try:
            for _ in range(3): fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(R):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            out[tag] = f"{ts[len(ts) // 2]:8.1f} us"
        except Exception as e:
            out[tag] = f"failed ({type(e).__name__}: {str(e)[:80]})"
    print(f"{name:32s} framework {out['framework']}   dinox {out['dinox']}")
