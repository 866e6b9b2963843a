#!/usr/bin/env python3
"""Where a tile of the ping-pong NT kernel spends its cycles: runs one product with the diagnostic stamp buffer (p.ws) and prints, for a few
workgroups, the s_memtime deltas of every K-tile, of the wait in front of the epilogue and of the epilogue.  CASE = qkv | fc1 | fc1t | dact | plain | fc2"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops
from dinox._lib import lib, GemmArgs, check, BF16, F32, EPI_BIAS, EPI_GELU, EPI_DGELU, EPI_AUXGRAD, EPI_RESIDUAL

os.environ["DINOX_NT_PP"] = os.environ.get("PP_MODE", "1")
os.environ.setdefault("DINOX_PP_ORDER", "1")
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M = int(os.environ.get("M", 102912))
case = os.environ.get("CASE", "fc1")
N, K = {"qkv": (1152, 384), "fc1": (1536, 384), "fc1t": (1536, 384), "dact": (1536, 384), "plain": (1536, 384), "fc2": (384, 1536)}[case]
a = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
w = (torch.randn(N, K, device=dev, generator=g) * 0.5).bfloat16()
bias = torch.randn(N, device=dev, generator=g)
odt = torch.float32 if case == "fc2" else torch.bfloat16
out = torch.empty(M, N, dtype=odt, device=dev)
aux = (torch.randn(M, N, device=dev, generator=g)).to(odt) if case in ("fc1", "dact") else None
res = torch.randn(M, N, device=dev, generator=g) if case == "fc2" else None
epi = {"qkv": EPI_BIAS, "fc1": EPI_BIAS | EPI_GELU | EPI_AUXGRAD, "fc1t": EPI_BIAS | EPI_GELU, "dact": EPI_DGELU | EPI_AUXGRAD, "plain": 0,
       "fc2": EPI_BIAS | EPI_RESIDUAL}[case]
dbg = torch.zeros(256 * 2 * 256, dtype=torch.int64, device=dev)
p = lambda t: None if t is None else t.data_ptr()


def args(ws):
    return GemmArgs(A=p(a), B=p(w), C=p(out), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, batch=1, strideA=0, strideB=0, strideC=M * N, transA=0, transB=0,
                    in_dtype=BF16, out_dtype=F32 if odt == torch.float32 else BF16, epilogue=epi, alpha=1.0, bias=p(bias), residual=p(res), ldr=N,
                    aux=p(aux), ldaux=N, colsum=None, ws=ws)


st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ga = args(None)
    check(lib.dinox_gemm(C.byref(ga), st), "gemm")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ga = args(p(dbg))
e0.record()
check(lib.dinox_gemm(C.byref(ga), st), "gemm")
e1.record()
torch.cuda.synchronize()
print(f"{case}: M {M} N {N} K {K}: {e0.elapsed_time(e1) * 1e3:.1f} us with stamps")
d = dbg.cpu().view(256, 2, 256)
nk = K // 64
per = nk + 4
for wg in (0, 1, 8, 77, 200, 255):
    for grp in (0, 1):
        s = d[wg, grp]
        n = int((s != 0).sum())
        ntile = n // per
        if ntile == 0:
            continue
        print(f"wg {wg} grp {grp}: {ntile} tiles; total {int(s[ntile * per - 1] - s[0])} cycles (100 MHz ticks x?)")
        for t in range(ntile):
            b = s[t * per:(t + 1) * per].tolist()
            kts = [b[i + 1] - b[i] for i in range(1, nk)] + [b[nk + 1] - b[nk]]
            print(f"   tile {t}: lead {b[1] - b[0]:6d} | K-tiles {' '.join(f'{x:6d}' for x in kts)} | wait0 {b[nk + 2] - b[nk + 1]:6d} | epilogue {b[nk + 3] - b[nk + 2]:6d}"
                  + (f" | gap {int(s[(t + 1) * per]) - b[nk + 3]:5d}" if t + 1 < ntile else ""))
