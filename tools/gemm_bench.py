#!/usr/bin/env python3
"""Micro-benchmark of dinox_gemm on the hot-path shapes (ViT-S/16 bs256: M = 102912 tokens).
Interleaved rounds in one process, HIP-event timing, random data (cdna_hip_programming.md rules 24/25)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops

M = int(os.environ.get("M", 102912))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def rb(*s):
    # DATA=narrow: |x| in [0.125, 0.5) with a random sign (one exponent value, as tools/tile_probe.hip fills its operands);
    # DATA=gelu: GELU of a unit normal (what fc2 / dW2 see in the step); default: 0.5 * unit normal.  The chip's clocks follow the
    # operands' toggle rate (MI355X_MICROARCH.md, DVFS), so the same kernel reads differently on each.
    kind = os.environ.get("DATA", "")
    if kind == "narrow":
        return ((torch.rand(*s, device=dev, generator=g) * 0.375 + 0.125) * (torch.randint(0, 2, s, device=dev, generator=g) * 2 - 1)).bfloat16()
    if kind == "gelu":
        return torch.nn.functional.gelu(torch.randn(*s, device=dev, generator=g)).bfloat16()
    return (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
def rf(*s): return torch.randn(*s, device=dev, generator=g)

cases = {}
def case(name, fn, flops, bytes_):
    cases[name] = (fn, flops, bytes_)

D, H = 384, 1536
x = rb(M, D); xh = rb(M, H); x3 = rb(M, 3 * D)
wqkv, wproj, w1, w2 = rb(3 * D, D), rb(D, D), rb(H, D), rb(D, H)
wqkvT, w1T, w2T = rb(D, 3 * D), rb(D, H), rb(H, D)
bq, bd, bh = rf(3 * D), rf(D), rf(H)
res = rf(M, D)
pre = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
case("qkv   NT K384 N1152 bias        ", lambda: ops.gemm(x, wqkv, bias=bq), 2 * M * D * 3 * D, M * D * 2 + M * 3 * D * 2)
case("proj  NT K384 N384  bias+res f32", lambda: ops.gemm(x, wproj, bias=bd, residual=res, out_dtype=torch.float32), 2 * M * D * D, M * D * 2 + 2 * M * D * 4)
case("fc1   NT K384 N1536 bias+gelu+aux", lambda: ops.gemm(x, w1, bias=bh, gelu=True, aux=pre, auxgrad=True), 2 * M * D * H, M * D * 2 + 2 * M * H * 2)
case("fc1t  NT K384 N1536 bias+gelu    ", lambda: ops.gemm(x, w1, bias=bh, gelu=True), 2 * M * D * H, M * D * 2 + M * H * 2)
case("fc2   NT K1536 N384 bias+res f32", lambda: ops.gemm(xh, w2, bias=bd, residual=res, out_dtype=torch.float32), 2 * M * D * H, M * H * 2 + 2 * M * D * 4)
case("fc2p  NT K1536 N384 plain f32 out ", lambda: ops.gemm(xh, w2, out_dtype=torch.float32), 2 * M * D * H, M * H * 2 + M * D * 4)
case("fc2b  NT K1536 N384 bias f32 out  ", lambda: ops.gemm(xh, w2, bias=bd, out_dtype=torch.float32), 2 * M * D * H, M * H * 2 + M * D * 4)
case("dact  NT K384 N1536 dgelu        ", lambda: ops.gemm(x, w2T, dgelu=True, aux=pre, auxgrad=True), 2 * M * D * H, M * D * 2 + 2 * M * H * 2)
case("dxn2  NT K1536 N384 plain        ", lambda: ops.gemm(xh, w1T), 2 * M * D * H, M * H * 2 + M * D * 2)
case("dxn1  NT K1152 N384 plain        ", lambda: ops.gemm(x3, wqkvT), 2 * M * D * 3 * D, M * 3 * D * 2 + M * D * 2)
gam, bet = rf(D), rf(D)
case("projLN rowln K384 +res +LN->bf16 ", lambda: ops.linear_residual_ln(x, wproj, bd, res, gam, bet, 1e-5, torch.bfloat16), 2 * M * D * D, M * D * 2 + 2 * M * D * 4 + M * D * 2)
case("fc2LN  rowln K1536 +res +LN->bf16", lambda: ops.linear_residual_ln(xh, w2, bd, res, gam, bet, 1e-5, torch.bfloat16), 2 * M * D * H, M * H * 2 + 2 * M * D * 4 + M * D * 2)
case("ln_fwd 384 fp32 -> bf16          ", lambda: ops.layernorm_fwd(res, gam, bet, torch.bfloat16), 8 * M * D, M * D * 4 + M * D * 2)
_y, _mean, _rstd = ops.layernorm_fwd(res, gam, bet, torch.bfloat16)
case("ln_bwd 384 bf16 dy +dx_add +lowp   ", lambda: ops.layernorm_bwd(x, res, gam, _mean, _rstd, dx_add=res, want_lowp=True), 16 * M * D, M * D * (2 + 4 + 4 + 4 + 2))
db = torch.empty(H, device=dev)
case("dW1   TN M1536 N384  (+db)       ", lambda: ops.gemm(xh, x, transA=True, transB=True, out_dtype=torch.float32, colsum_out=db), 2 * M * D * H, M * H * 2 + M * D * 2)
dbq = torch.empty(3 * D, device=dev)
case("dWqkv TN M1152 N384  (+db)       ", lambda: ops.gemm(x3, x, transA=True, transB=True, out_dtype=torch.float32, colsum_out=dbq), 2 * M * D * 3 * D, M * 3 * D * 2 + M * D * 2)
dbd = torch.empty(D, device=dev)
case("dW2   TN M384 N1536  (+db)       ", lambda: ops.gemm(x, xh, transA=True, transB=True, out_dtype=torch.float32, colsum_out=dbd), 2 * M * D * H, M * H * 2 + M * D * 2)
case("dWp   TN M384 N384   (+db)       ", lambda: ops.gemm(x, x, transA=True, transB=True, out_dtype=torch.float32, colsum_out=dbd), 2 * M * D * D, 2 * M * D * 2)

if os.environ.get("VITB"):
    Db = 768
    xb_, wqb, wpb = rb(M, Db), rb(3 * Db, Db), rb(Db, Db)
    bqb, bdb, resb = rf(3 * Db), rf(Db), rf(M, Db)
    case("vitb qkv  NT K768 N2304 bias   ", lambda: ops.gemm(xb_, wqb, bias=bqb), 2 * M * Db * 3 * Db, M * Db * 2 + M * 3 * Db * 2)
    case("vitb proj NT K768 N768 bias+res", lambda: ops.gemm(xb_, wpb, bias=bdb, residual=resb, out_dtype=torch.float32), 2 * M * Db * Db, M * Db * 2 + 2 * M * Db * 4)
if os.environ.get("SQUARE"):
    S = int(os.environ["SQUARE"])
    sa, sb = rb(S, S), rb(S, S)
    case(f"square NT {S}^3 plain          ", lambda: ops.gemm(sa, sb), 2 * S ** 3, 3 * S * S * 2)
    case(f"square TN {S}^3 plain f32out   ", lambda: ops.gemm(sa, sb, transA=True, transB=True, out_dtype=torch.float32), 2 * S ** 3, 2 * S * S * 2 + S * S * 4)

sel = os.environ.get("CASES")
names = [n for n in cases if not sel or any(s in n for s in sel.split(","))]
for n in names:
    for _ in range(3): cases[n][0]()
torch.cuda.synchronize()
R = int(os.environ.get("ROUNDS", 10))
times = {n: [] for n in names}
for r in range(R):
    for n in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); cases[n][0](); e1.record()
        times[n].append((e0, e1))
torch.cuda.synchronize()
for n in names:
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in times[n])
    med = ts[len(ts) // 2]
    fn, fl, by = cases[n]
    print(f"{n}  med {med:8.1f} us  min {ts[0]:8.1f} us   {fl / med / 1e6:7.1f} TFLOP/s   {by / med / 1e6:6.2f} TB/s alg")
