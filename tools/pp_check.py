#!/usr/bin/env python3
"""Correctness screen + A/B timing of the persistent ping-pong NT kernel (csrc/gemm_bf16_pp.hip) against the shipped NT kernels.
DINOX_NT_PP is read by the dispatcher on every call, so both run interleaved in ONE process (cdna_hip_programming.md rule 24).
  python tools/pp_check.py            correctness (several shapes / epilogues, repeated launches) then timing at M = 102912
  CHECK=0 / TIME=0 skip a part;  ORDERS="0,1" tile orders to time;  M=... token count"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dino-x_amd")]
import torch
from dinox import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def rb(*s):
    return (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()


def rf(*s):
    return torch.randn(*s, device=dev, generator=g)


def run(pp, fn):
    os.environ["DINOX_NT_PP"] = (os.environ.get("PP_MODE_V") or os.environ.get("PP_MODE", "1")) if pp else "0"
    ops.TRACE_KERNELS = []
    out = fn()
    names = ops.TRACE_KERNELS
    ops.TRACE_KERNELS = None
    return out, names


def gelu_ref(x):
    return torch.nn.functional.gelu(x), 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * 3.141592653589793) ** 0.5


def check():
    bad = 0
    shapes = [(256, 256, 192), (512, 512, 384), (1000, 392, 384), (777, 1152, 384), (2048, 1536, 384), (1300, 384, 1536), (4096, 1152, 1152),
              (256 * 9 + 17, 1536, 192), (3000, 1024, 1024), (515, 264, 4096), (5000, 384, 384), (70000, 384, 1152), (300, 128, 448), (2049, 120, 192)]
    if os.environ.get("FORCE"):            # FORCE=2 / 3: every shape on the 128-wide / 256-wide kernel
        os.environ["PP_MODE"] = os.environ["FORCE"]
    for (M, N, K) in shapes:
        a, w = rb(M, K), rb(N, K)
        bias, res = rf(N), rf(M, N)
        ref = a.float() @ w.float().t()
        for name in ("plain_bf16", "bias_bf16", "plain_f32", "bias_res_f32", "gelu_aux", "gelu_noaux", "dgelu"):
            aux = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
            if name == "plain_bf16":
                fn = lambda: ops.gemm(a, w)
                want = ref
            elif name == "bias_bf16":
                fn = lambda: ops.gemm(a, w, bias=bias)
                want = ref + bias
            elif name == "plain_f32":
                fn = lambda: ops.gemm(a, w, out_dtype=torch.float32)
                want = ref
            elif name == "bias_res_f32":
                fn = lambda: ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32)
                want = ref + bias + res
            elif name == "gelu_aux":
                fn = lambda: (ops.gemm(a, w, bias=bias, gelu=True, aux=aux, auxgrad=True), aux)
                want = gelu_ref(ref + bias)
            elif name == "gelu_noaux":
                fn = lambda: ops.gemm(a, w, bias=bias, gelu=True)
                want = gelu_ref(ref + bias)[0]
            else:
                auxin = rb(M, N)
                fn = lambda: ops.gemm(a, w, dgelu=True, aux=auxin, auxgrad=True)
                want = ref * auxin.float()
            for rep in range(3):
                got, names = run(True, fn)
                torch.cuda.synchronize()
                if names not in (["gemm_bf16_nt_pp"], ["gemm_bf16_nt_pp128"]):
                    print(f"  !! {M}x{N}x{K} {name}: dispatched to {names}")
                    bad += 1
                    break
                pairs = list(zip(got, want)) if isinstance(got, tuple) else [(got, want)]
                for gi, (go, wa) in enumerate(pairs):
                    err = (go.float() - wa).abs()
                    tol = 2e-2 * wa.abs() + 2e-2 if go.dtype == torch.bfloat16 else 2e-3 * wa.abs() + 2e-3
                    nbad = int((err > tol).sum())
                    if nbad:
                        idx = torch.nonzero(err > tol)[:5].tolist()
                        print(f"  !! {M}x{N}x{K} {name}[{gi}] rep {rep}: {nbad} wrong elements, max err {float(err.max()):.4g}, first at {idx}")
                        bad += 1
            # bit-repeatability against the first launch
            first, _ = run(True, fn)
            f0 = first[0] if isinstance(first, tuple) else first
            for rep in range(5):
                again, _ = run(True, fn)
                a0 = again[0] if isinstance(again, tuple) else again
                if not torch.equal(f0, a0):
                    print(f"  !! {M}x{N}x{K} {name}: launch {rep} differs from launch 0 in {int((f0 != a0).sum())} elements")
                    bad += 1
                    break
        print(f"shape {M}x{N}x{K}: checked", flush=True)
    print("CHECK", "FAILED" if bad else "ok", flush=True)
    return bad


def timing():
    M = int(os.environ.get("M", 102912))
    D, H = 384, 1536
    x, xh, x3 = rb(M, D), rb(M, H), rb(M, 3 * D)
    wqkv, wproj, w1, w2 = rb(3 * D, D), rb(D, D), rb(H, D), rb(D, H)
    wqkvT, w1T, w2T = rb(D, 3 * D), rb(D, H), rb(H, D)
    bq, bd, bh = rf(3 * D), rf(D), rf(H)
    res = rf(M, D)
    pre = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    cases = {
        "qkv   K384  N1152 bias": (lambda: ops.gemm(x, wqkv, bias=bq), 2 * M * D * 3 * D),
        "fc1   K384  N1536 gelu+aux": (lambda: ops.gemm(x, w1, bias=bh, gelu=True, aux=pre, auxgrad=True), 2 * M * D * H),
        "fc1t  K384  N1536 gelu": (lambda: ops.gemm(x, w1, bias=bh, gelu=True), 2 * M * D * H),
        "dact  K384  N1536 dgelu": (lambda: ops.gemm(x, w2T, dgelu=True, aux=pre, auxgrad=True), 2 * M * D * H),
        "plain K384  N1536": (lambda: ops.gemm(x, w1), 2 * M * D * H),
        "proj  K384  N384 bias+res f32": (lambda: ops.gemm(x, wproj, bias=bd, residual=res, out_dtype=torch.float32), 2 * M * D * D),
        "dxp   K384  N384 plain": (lambda: ops.gemm(x, wproj), 2 * M * D * D),
        "fc2   K1536 N384 bias+res f32": (lambda: ops.gemm(xh, w2, bias=bd, residual=res, out_dtype=torch.float32), 2 * M * D * H),
        "dxn2  K1536 N384 plain": (lambda: ops.gemm(xh, w1T), 2 * M * D * H),
        "dxn1  K1152 N384 plain": (lambda: ops.gemm(x3, wqkvT), 2 * M * D * 3 * D),
    }
    if os.environ.get("VITL"):
        Ml, Dl = 51456, 1024
        xl, xl4 = rb(Ml, Dl), rb(Ml, 4 * Dl)
        wq, w1l, w2l = rb(3 * Dl, Dl), rb(4 * Dl, Dl), rb(Dl, 4 * Dl)
        resl, prel = rf(Ml, Dl), torch.empty(Ml, 4 * Dl, dtype=torch.bfloat16, device=dev)
        cases.update({
            "L qkv  K1024 N3072": (lambda: ops.gemm(xl, wq), 2 * Ml * Dl * 3 * Dl),
            "L fc1  K1024 N4096 gelu+aux": (lambda: ops.gemm(xl, w1l, gelu=True, aux=prel, auxgrad=True), 2 * Ml * Dl * 4 * Dl),
            "L fc2  K4096 N1024 res f32": (lambda: ops.gemm(xl4, w2l, residual=resl, out_dtype=torch.float32), 2 * Ml * Dl * 4 * Dl),
        })
    sel = os.environ.get("CASES")
    names = [n for n in cases if not sel or any(s in n for s in sel.split(","))]
    orders = os.environ.get("ORDERS", "1").split(",")
    variants = [("old", False, "0", None)] + [(f"pp/o{o}", True, o, None) for o in orders] + [("pp/ns", True, "1", "0")]
    if os.environ.get("ABL"):
        variants += [("nostore", True, str(1 + 256), None), ("stagall", True, str(1 + 512), None)]
    R = int(os.environ.get("ROUNDS", 8))
    times = {(n, v[0]): [] for n in names for v in variants}
    def setenv(v):
        os.environ["DINOX_PP_ORDER"] = v[2]
        os.environ["PP_MODE_V"] = v[4] if len(v) > 4 else ""
        if v[3] is None:
            os.environ.pop("DINOX_PP_STAGGER", None)
        else:
            os.environ["DINOX_PP_STAGGER"] = v[3]
    for n in names:
        for v in variants:
            setenv(v)
            for _ in range(2):
                run(v[1], cases[n][0])
    torch.cuda.synchronize()
    for r in range(R):
        for n in names:
            for v in variants:
                setenv(v)
                os.environ["DINOX_NT_PP"] = (os.environ.get("PP_MODE_V") or os.environ.get("PP_MODE", "1")) if v[1] else "0"
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                cases[n][0]()
                e1.record()
                times[(n, v[0])].append((e0, e1))
    torch.cuda.synchronize()
    for n in names:
        line = f"{n:32s}"
        for v in variants:
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in times[(n, v[0])])
            med = ts[len(ts) // 2]
            line += f" | {v[0]:6s} med {med:7.1f} min {ts[0]:7.1f} us {cases[n][1] / med / 1e6:6.0f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    rc = 0
    if os.environ.get("CHECK", "1") != "0":
        rc = check()
    if os.environ.get("TIME", "1") != "0":
        timing()
    sys.exit(1 if rc else 0)
