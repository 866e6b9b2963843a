"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C ABI, against
the golden fixtures captured from the reference and against the CPU oracle on the same seeded inputs.

Tolerances: fp32 "parity mode" must meet the north-star gate of 1e-3 relative (we assert 1e-4 or
tighter where fp32 round-off allows); bf16 "throughput mode" is compared by relative L2 error, since
bf16 has 2^-9 relative precision per element (the reference's --amp path has the same property).
"""
import contextlib
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, sub, t

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def dx():
    from dinox import ops
    import zoo.arch as arch
    import dinox._lib as L
    assert L.lib.dinox_device_ok() == 1, L.last_error()
    return ops, arch


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def close(a, b, rtol, atol=0.0, what=""):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    assert err <= atol + rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rtol {rtol})"


AMP_FACTOR = 1.5


def assert_within_autocast_distance(got: dict, ref_f32: dict, ref_amp: dict, what: str, factor: float = AMP_FACTOR, floor: float = 0.0):
    """The bf16 throughput mode against the reference's own --amp arithmetic (torch.autocast(bfloat16), scripts/phase5_big_run.py:
    1716-1717): for every tensor,  relL2(HIP bf16, reference fp32)  <=  factor x relL2(reference autocast, reference fp32).
    I.e. the HIP step may sit at most 1.5x as far from the fp32 step as the reference's own bf16 step does, per parameter."""
    bad, worst = [], (0.0, "")
    for n, r32 in ref_f32.items():
        r32 = torch.as_tensor(r32)
        if float(r32.abs().max()) <= 1e-6:          # numerically-zero gradients (key bias ...): relative error is meaningless
            continue
        d_ref = rel_l2(ref_amp[n], r32)
        d_hip = rel_l2(got[n], r32)
        ratio = d_hip / max(d_ref, 1e-12)
        if ratio > worst[0]:
            worst = (ratio, n)
        if d_hip > factor * d_ref + floor:
            bad.append((n, round(d_hip, 5), round(d_ref, 5)))
    assert not bad, f"{what}: further from fp32 than {factor} x the reference's autocast step (name, hip, reference): {bad[:8]} ({len(bad)} tensors)"
    return worst


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (77, 130, 45), (201, 96, 384), (5, 3, 7)])
def test_gemm_f32_layouts(dx, tA, tB, M, N, K):
    ops, _ = dx
    g = torch.Generator().manual_seed(M * 1000 + N * 10 + K)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((K, N) if tB else (N, K), generator=g)
    ref = (A.t() if tA else A).double() @ (B if tB else B.t()).double()
    out = ops.gemm(A.to(DEV), B.to(DEV), transA=bool(tA), transB=bool(tB))
    close(out, ref, rtol=2e-6, atol=1e-5, what=f"gemm {tA}{tB}")


def test_gemm_f32_epilogues_and_batch(dx):
    ops, _ = dx
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(3, 50, 40, generator=g), torch.randn(3, 70, 40, generator=g)
    bias, res = torch.randn(70, generator=g), torch.randn(3, 50, 70, generator=g)
    pre_ref = torch.einsum("bmk,bnk->bmn", A.double(), B.double()) * 0.5 + bias.double()
    aux = torch.empty(3, 50, 70, device=DEV)
    out = ops.gemm(A.to(DEV), B.to(DEV), bias=bias.to(DEV), gelu=True, aux=aux, residual=res.to(DEV), alpha=0.5)
    gelu = 0.5 * pre_ref * (1 + torch.erf(pre_ref / math.sqrt(2)))
    close(aux, pre_ref, 1e-5, 1e-5, "aux")
    close(out, gelu + res.double(), 1e-5, 1e-5, "gelu+res")
    # DGELU + ACCUM
    c0 = torch.randn(3, 50, 70, generator=g)
    out2 = c0.to(DEV).clone()
    ops.gemm(A.to(DEV), B.to(DEV), out=out2, dgelu=True, aux=aux, accumulate=True)
    x = pre_ref
    dg = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    close(out2, c0.double() + torch.einsum("bmk,bnk->bmn", A.double(), B.double()) * dg, 1e-5, 1e-5, "dgelu+accum")


@pytest.mark.parametrize("tA,tB,M,N,K", [(0, 0, 256, 384, 384), (0, 0, 201 * 4, 1152, 384), (0, 0, 130, 72, 200), (0, 0, 77, 200, 128),
                                         (1, 1, 384, 1536, 804), (1, 1, 200, 384, 200), (1, 1, 96, 72, 1000),
                                         (0, 1, 64, 64, 64), (0, 0, 512, 8192, 384)])
def test_gemm_bf16_vs_exact(dx, tA, tB, M, N, K):
    """bf16 operands: products are exact in fp32, so the result must match an fp64 product of the
    bf16-rounded inputs to fp32-accumulation accuracy, whatever kernel the dispatcher picks."""
    ops, _ = dx
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), generator=g).bfloat16()
    B = torch.randn((K, N) if tB else (N, K), generator=g).bfloat16()
    ref = (A.t() if tA else A).double() @ (B if tB else B.t()).double()
    ops.TRACE_KERNELS = []
    out = ops.gemm(A.to(DEV), B.to(DEV), transA=bool(tA), transB=bool(tB), out_dtype=torch.float32)
    used, ops.TRACE_KERNELS = ops.TRACE_KERNELS, None
    # the MFMA-bf16 kernels must take every aligned NT / TN shape; only (0,1) falls to the fp32-MFMA kernel
    nt = "gemm_bf16_nt"
    if K % 64 == 0 and N % 8 == 0:
        nt = "gemm_bf16_nt_areg" if K in (384, 576) else "gemm_bf16_nt_glds"      # short K: token operand through the register file
    want = {(0, 0): nt, (1, 1): "gemm_bf16_tn_dma", (0, 1): "gemm_f32"}[(tA, tB)]
    assert used == [want], used
    close(out, ref, rtol=1e-5, atol=1e-5 * math.sqrt(K), what=f"bf16 gemm {tA}{tB} {M}x{N}x{K}")
    out_b = ops.gemm(A.to(DEV), B.to(DEV), transA=bool(tA), transB=bool(tB))
    assert out_b.dtype == torch.bfloat16
    assert rel_l2(out_b.float(), ref) < 4e-3


def test_gemm_bf16_epilogues(dx):
    ops, _ = dx
    g = torch.Generator().manual_seed(9)
    M, N, K = 402, 256, 128
    A, B = torch.randn(M, K, generator=g).bfloat16(), (0.1 * torch.randn(N, K, generator=g)).bfloat16()
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    pre = A.double() @ B.double().t() + bias.double()
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    act = ops.gemm(A.to(DEV), B.to(DEV), bias=bias.to(DEV), gelu=True, aux=aux)
    assert rel_l2(aux.float(), pre) < 4e-3
    assert rel_l2(act.float(), 0.5 * pre * (1 + torch.erf(pre / math.sqrt(2)))) < 5e-3
    y = ops.gemm(A.to(DEV), B.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.float32)
    close(y, pre + res.double(), 1e-5, 1e-4, "bias+residual fp32 out")
    auxf = aux.float().double().cpu()
    dg = 0.5 * (1 + torch.erf(auxf / math.sqrt(2))) + auxf * torch.exp(-0.5 * auxf * auxf) / math.sqrt(2 * math.pi)
    d = ops.gemm(A.to(DEV), B.to(DEV), dgelu=True, aux=aux)
    assert rel_l2(d.float(), (A.double() @ B.double().t()) * dg) < 5e-3
    # AUXGRAD: forward stores gelu'(pre) in aux, backward multiplies by it
    aux2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    act2 = ops.gemm(A.to(DEV), B.to(DEV), bias=bias.to(DEV), gelu=True, aux=aux2, auxgrad=True)
    dgp = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    assert rel_l2(aux2.float(), dgp) < 5e-3 and rel_l2(act2.float(), act.float()) < 1e-3
    d2 = ops.gemm(A.to(DEV), B.to(DEV), dgelu=True, aux=aux2, auxgrad=True)
    assert rel_l2(d2.float(), (A.double() @ B.double().t()) * aux2.float().double().cpu()) < 5e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,nout,kin", [(804, 384, 256), (1000, 136, 72), (77, 64, 64)])
def test_gemm_tn_fused_colsum(dx, dt, rows, nout, kin):
    """dW = dY^T X with the bias gradient colsum(dY) riding along (TN kernel) or via the fallback."""
    ops, _ = dx
    g = torch.Generator().manual_seed(rows)
    dy, x = torch.randn(rows, nout, generator=g).to(dt), torch.randn(rows, kin, generator=g).to(dt)
    db = torch.full((nout,), 7.0, device=DEV)            # must be overwritten, not accumulated
    dw = ops.gemm(dy.to(DEV), x.to(DEV), transA=True, transB=True, out_dtype=torch.float32, colsum_out=db)
    close(dw, dy.double().t() @ x.double(), 1e-5, 1e-4 * math.sqrt(rows), "dW")
    close(db, dy.double().sum(0), 1e-5, 1e-4 * math.sqrt(rows), "db")


def test_colsum(dx):
    ops, _ = dx
    x = torch.randn(1003, 130)
    close(ops.colsum(x.to(DEV)), x.double().sum(0), 1e-5, 1e-4, "colsum f32")
    xb = x.bfloat16()
    close(ops.colsum(xb.to(DEV)), xb.double().sum(0), 1e-5, 1e-4, "colsum bf16")


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,dim", [(7, 64), (804, 384), (33, 1024), (5, 50),
                                      (1, 384), (3, 384), (4099, 384), (33001, 384)])   # width 384 has its own kernels (half a wave per row,
def test_layernorm(dx, rows, dim):                                                        # two rows per half and sweep): ragged row counts
    ops, _ = dx
    from oracle import kernels_np as K
    rng = np.random.default_rng(rows + dim)
    x, w, b, dy = rng.normal(size=(rows, dim)) * 2 + 0.5, rng.normal(size=dim), rng.normal(size=dim), rng.normal(size=(rows, dim))
    y_ref, mu, rstd = K.layernorm_fwd(x, w, b)
    dx_ref, dw_ref, db_ref = K.layernorm_bwd(dy, x, w, mu, rstd)
    X, W, Bb, DY = [torch.tensor(a, dtype=torch.float32, device=DEV) for a in (x, w, b, dy)]
    y, mean, rs = ops.layernorm_fwd(X, W, Bb, torch.float32)
    close(y, y_ref, 1e-5, 1e-5, "ln y")
    yb, _, _ = ops.layernorm_fwd(X, W, Bb, torch.bfloat16)
    assert rel_l2(yb.float(), y_ref) < 4e-3
    base = torch.ones(rows, dim, device=DEV)
    dxo, dw, db, lowp = ops.layernorm_bwd(DY, X, W, mean, rs, dx_add=base, want_lowp=True)        # out of place
    inpl = base.clone()
    ops.layernorm_bwd(DY, X, W, mean, rs, dx=inpl, dx_add=inpl)                                   # in place
    assert torch.equal(inpl, dxo)
    close(dxo, dx_ref + 1.0, 2e-5, 2e-5, "ln dx (accumulate)")
    close(dw, dw_ref, 2e-5, 1e-4, "ln dw")
    close(db, db_ref, 2e-5, 1e-4, "ln db")
    assert rel_l2(lowp.float(), dx_ref + 1.0) < 4e-3
    dxb, dwb, _, _ = ops.layernorm_bwd(DY.bfloat16(), X, W, mean, rs)
    assert rel_l2(dxb, dx_ref) < 6e-3 and rel_l2(dwb, dw_ref) < 6e-3


# ------------------------------------------------------------------------------------------ modules vs golden
def _load(mod, sd):
    mod.load_state_dict({k: v for k, v in sd.items()})
    return mod.to(DEV)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_attention_module_golden(dx, mode):
    ops, arch = dx
    g = load_golden("ops_attention.npz")
    m = _load(arch.Attention(128, num_heads=int(g["heads"])), sub(g, "w"))
    x = t(g["x"]).to(DEV).requires_grad_(True)
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    with ops.compute_dtype(dt):
        y = m(x)
        y.float().backward(t(g["dy"]).to(DEV))
    if mode == "fp32":
        close(y, g["y"], 1e-4, 1e-5, "attn y")
        close(x.grad, g["dx"], 1e-4, 1e-5, "attn dx")
        for k, v in sub(g, "g").items():
            close(dict(m.named_parameters())[k].grad, v, 2e-4, 1e-5, f"attn d{k}")
    else:
        assert rel_l2(y.float(), g["y"]) < 1.5e-2
        assert rel_l2(x.grad, g["dx"]) < 2.5e-2
        for k, v in sub(g, "g").items():
            assert rel_l2(dict(m.named_parameters())[k].grad, v) < 2.5e-2, k


@pytest.mark.parametrize("B,N,h,d", [(2, 201, 6, 64), (1, 19, 2, 32), (3, 64, 1, 64), (1, 261, 2, 64), (2, 7, 2, 16), (40, 250, 6, 64), (50, 37, 6, 64), (3, 530, 2, 64), (60, 201, 6, 64),
                                     # round 3, the tiled kernels (csrc/attention_flash.hip): 448 / 512 px inputs, the vit-giant head size, 128, ragged tails
                                     (8, 789, 6, 64), (4, 1029, 6, 64), (8, 261, 16, 88), (2, 300, 2, 32), (3, 333, 3, 128), (17, 290, 2, 40), (2, 1342, 2, 88)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_attention_core_vs_np(dx, B, N, h, d, mode):
    """softmax(Q K^T / sqrt(d)) V and its backward on the packed qkv tensor against the float64 NumPy statement (oracle/kernels_np.py):
    the whole-strip MFMA kernels (d = 64, up to 288 / 544 tokens), the tiled MFMA kernels (any length, d <= 128: (8, 789, 6, 64),
    (4, 1029, 6, 64) and (8, 261, 16, 88) are the shapes --img-size 448 / 512 and the vit-giant preset reach), the per-lane reference
    kernels (small / fp32) and the fp32 product form (full-size fp32)."""
    ops, _ = dx
    from oracle import kernels_np as K
    rng = np.random.default_rng(N * 7 + d)
    qkv = rng.normal(size=(B, N, 3 * h * d)).astype(np.float32)
    do = rng.normal(size=(B, N, h * d)).astype(np.float32)
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    Q = torch.tensor(qkv, device=DEV).to(dt)
    DO = torch.tensor(do, device=DEV).to(dt)
    qkv_r, do_r = Q.float().cpu().double().numpy(), DO.float().cpu().double().numpy()
    o_ref, lse_ref = K.attention_core_fwd(qkv_r, h)
    o, lse = ops.attention_fwd(Q, h)
    dq_ref = K.attention_core_bwd(do_r, qkv_r, o_ref, lse_ref, h)
    dqkv = ops.attention_bwd(DO, Q, o, lse, h)
    if mode == "fp32":
        close(o, o_ref, 2e-5, 2e-5, "o")
        close(lse, lse_ref, 2e-5, 2e-5, "lse")
        close(dqkv, dq_ref, 1e-4, 1e-4, "dqkv")
    else:
        assert rel_l2(o.float(), o_ref) < 8e-3
        close(lse, lse_ref, 1e-3, 1e-2, "lse")
        assert rel_l2(dqkv.float(), dq_ref) < 2e-2


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_mlp_module_golden(dx, mode):
    ops, arch = dx
    g = load_golden("ops_mlp.npz")
    m = _load(arch.Mlp(64, 4.0), sub(g, "w"))
    x = t(g["x"]).to(DEV).requires_grad_(True)
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    with ops.compute_dtype(dt):
        y = m(x)
        y.float().backward(t(g["dy"]).to(DEV))
    P = dict(m.named_parameters())
    if mode == "fp32":
        close(y, g["y"], 1e-4, 1e-5, "mlp y")
        close(x.grad, g["dx"], 1e-4, 1e-5, "mlp dx")
        for k, v in sub(g, "g").items():
            close(P[k].grad, v, 2e-4, 1e-5, f"mlp d{k}")
    else:
        assert rel_l2(y.float(), g["y"]) < 1.5e-2
        assert rel_l2(x.grad, g["dx"]) < 2e-2
        for k, v in sub(g, "g").items():
            assert rel_l2(P[k].grad, v) < 2e-2, k


def test_scale_embedding_golden(dx):
    ops, arch = dx
    g = load_golden("ops_scale_embed.npz")
    m = _load(arch.ScaleEmbedding(64), sub(g, "w"))
    sp = t(g["spacing"]).to(DEV).requires_grad_(True)
    y = m(sp)
    assert y.shape == (5, 1, 64)
    close(y, g["y"], 1e-4, 1e-5, "scale y")
    y.backward(t(g["dy"]).to(DEV))
    close(sp.grad, g["dspacing"], 5e-4, 1e-5, "dspacing")
    P = dict(m.named_parameters())
    for k, v in sub(g, "g").items():
        close(P[k].grad, v, 5e-4, 1e-5, f"scale d{k}")


def test_scale_embedding_zero_init_is_noop(dx):
    """reference tests/test_scale_embedding.py:51-62: a fresh module outputs ~0."""
    _, arch = dx
    m = arch.ScaleEmbedding(64).to(DEV)
    y = m(torch.tensor([[0.5, 0.5, 1.0], [1.5, 1.5, 5.0]], device=DEV))
    assert y.abs().max().item() < 1e-3


def test_dino_loss_golden(dx):
    ops, _ = dx
    g = load_golden("dino_loss.npz")
    s = t(g["s"]).to(DEV).requires_grad_(True)
    c = t(g["center0"]).to(DEV).clone()
    l1 = ops.DinoCEFn.apply(s, t(g["t"]).to(DEV), c, 0.1, 0.04)
    close(l1, g["loss1"], 1e-5, 1e-6, "dino loss1")
    l1.backward()
    close(s.grad, g["ds1"], 1e-4, 1e-8, "dino ds")
    ops.center_ema_(c.view(-1), ops.colmean(t(g["t"]).to(DEV)), 0.9)
    close(c, g["center1"], 1e-5, 1e-7, "center1")
    l2, _ = ops.dino_ce(s.detach(), t(g["t2"]).to(DEV), c, 0.1, 0.04, False)
    close(l2.reshape(()), g["loss2"], 1e-5, 1e-6, "dino loss2")


def test_dino_loss_survey_known_answer(dx):
    ops, _ = dx
    from oracle.dinox_oracle import det
    S, T = det(8, 128, f=0.37).to(DEV), (2 * det(8, 128, f=0.91, ph=0.5)).to(DEV)
    c = torch.zeros(1, 128, device=DEV)
    l, _ = ops.dino_ce(S, T, c, 0.1, 0.04, False)
    assert float(l) == pytest.approx(12.77463341, rel=1e-5)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gram_loss_golden(dx, mode):
    ops, _ = dx
    g = load_golden("gram_loss.npz")
    sf = t(g["sf"]).to(DEV).requires_grad_(True)
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    with ops.compute_dtype(dt):
        l = ops.GramLossFn.apply(sf, t(g["tf"]).to(DEV))
        l.backward()
    ref = torch.as_tensor(g["dsf"]).clone()
    got = sf.grad.cpu().clone()
    if mode == "fp32":
        close(l, g["loss"], 1e-5, 1e-7, "gram loss")
        # the zero-norm token's gradient is ~1e12-scaled noise in the reference too: compare it loosely
        assert rel_l2(got[1, 5], ref[1, 5]) < 1e-3
        got[1, 5] = 0; ref[1, 5] = 0
        close(got, ref, 2e-4, 1e-8, "gram dsf")
    else:
        assert float(l) == pytest.approx(float(g["loss"]), rel=2e-2)
        got[1, 5] = 0; ref[1, 5] = 0
        assert rel_l2(got, ref) < 3e-2


# ------------------------------------------------------------------------------------------ whole model
def _cfg(arr):
    img, patch, dim, depth, heads, regs, sa, out = [int(v) for v in arr]
    return dict(img_size=img, patch=patch, dim=dim, depth=depth, heads=heads, num_registers=regs, scale_aware=bool(sa)), out


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_vit_tiny_golden(dx, mode):
    ops, arch = dx
    g = load_golden("vit_tiny.npz")
    cfg, out_dim = _cfg(g["cfg"])
    student = _load(arch.DinoStudentTeacher(arch.PatchViT(**cfg), out_dim), sub(g, "student"))
    teacher = _load(arch.DinoStudentTeacher(arch.PatchViT(**cfg), out_dim), sub(g, "teacher"))
    x, sp = t(g["x"]).to(DEV), t(g["spacing"]).to(DEV)
    center = t(g["center"]).to(DEV).clone()
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    with ops.compute_dtype(dt):
        s_feats = student.backbone(x, spacing=sp)
        with torch.no_grad():
            t_feats = teacher.backbone(x, spacing=sp)
            t_out = teacher.head(t_feats[:, 0])
            nosp = student.backbone(x, spacing=None)
        s_out = student.head(s_feats[:, 0])
        l_dino = ops.DinoCEFn.apply(s_out, t_out, center, 0.1, 0.04)
        l_gram = ops.GramLossFn.apply(s_feats, t_feats)
        (l_dino + l_gram).backward()
    assert s_feats.dtype == torch.float32 and s_feats.shape == (4, 21, 64)
    P = dict(student.named_parameters())
    if mode == "fp32":
        close(s_feats, g["s_feats"], 1e-4, 1e-5, "s_feats")
        close(t_feats, g["t_feats"], 1e-4, 1e-5, "t_feats")
        close(nosp, g["feats_nospacing"], 1e-4, 1e-5, "feats_nospacing")
        close(s_out, g["s_out"], 1e-4, 1e-5, "s_out")
        close(l_dino, g["loss_dino"], 1e-4, 0, "loss_dino")
        close(l_gram, g["loss_gram"], 1e-4, 0, "loss_gram")
        for n in [str(s) for s in g["param_order"]]:
            close(P[n].grad, g[f"grad/{n}"], 1e-3, 2e-7, f"grad {n}")      # the north-star 1e-3 gate
    else:
        # pinned to the reference's own autocast arithmetic: vit_tiny_autocast.npz is this same case run by the real reference under
        # torch.autocast(bfloat16) ("f32loss" = the GPU op policy: softmax / log_softmax in fp32)
        a = load_golden("vit_tiny_autocast.npz")
        names = [str(s) for s in g["param_order"]]
        assert_within_autocast_distance({n: P[n].grad for n in names}, {n: g[f"grad/{n}"] for n in names},
                                        {n: a[f"f32loss/grad/{n}"] for n in names}, "vit_tiny parameter gradients")
        assert_within_autocast_distance({"s_feats": s_feats, "s_out": s_out, "t_feats": t_feats},
                                        {k: g[k] for k in ("s_feats", "s_out", "t_feats")},
                                        {k: a[f"f32loss/{k}"] for k in ("s_feats", "s_out", "t_feats")}, "vit_tiny activations")
        for k, got_l in (("loss_dino", l_dino), ("loss_gram", l_gram)):
            d_ref = abs(float(a[f"f32loss/{k}"]) - float(g[k]))
            assert abs(float(got_l) - float(g[k])) <= AMP_FACTOR * d_ref + 2e-3 * abs(float(g[k])), (k, float(got_l), float(g[k]), float(a[f"f32loss/{k}"]))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_block_fused_node_matches_composed_ops(dx, mode):
    """TransformerBlock takes the single-node BlockFn path when its leaves are the stock module types and the
    per-op composition otherwise (e.g. LoRA-wrapped leaves); both must agree (and both are HIP-only)."""
    ops, arch = dx

    class Wrapped(arch.Linear):          # same math, different type -> forces the composed path
        pass

    torch.manual_seed(4)
    blk = arch.TransformerBlock(128, 2).to(DEV)
    with torch.no_grad():
        for p_ in blk.parameters():
            if p_.ndim == 1:
                p_.add_(0.05 * torch.randn_like(p_))
    blk2 = arch.TransformerBlock(128, 2).to(DEV)
    blk2.load_state_dict(blk.state_dict())
    w = Wrapped(128, 512).to(DEV)
    w.load_state_dict(blk.mlp.fc1.state_dict())
    blk2.mlp.fc1 = w
    assert blk._fusable() and not blk2._fusable()
    x = torch.randn(3, 37, 128, device=DEV)
    dy = torch.randn(3, 37, 128, device=DEV)
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    outs = []
    for b in (blk, blk2):
        xi = x.clone().requires_grad_(True)
        with ops.compute_dtype(dt):
            y = b(xi)
            y.backward(dy)
        outs.append((y.detach(), xi.grad, {n: p_.grad for n, p_ in b.named_parameters()}))
    tol = 1e-5 if mode == "fp32" else 2e-2
    assert rel_l2(outs[0][0], outs[1][0]) < tol
    assert rel_l2(outs[0][1], outs[1][1]) < tol
    for n in outs[0][2]:
        assert rel_l2(outs[0][2][n], outs[1][2][n]) < tol, n


@pytest.mark.parametrize("accum,ckpt", [(1, False), (2, False), (1, True)])
def test_native_block_sequencing_changes_no_bit(dx, accum, ckpt):
    """dinox_block_forward / dinox_block_backward (csrc/block.hip: one C-ABI call per transformer block instead of ~13 / ~15 from Python)
    enqueue the same kernels in the same order on the same buffers' worth of data: two optimiser steps of a ViT-S-width model through
    them must end BIT-identical to the same steps with the block node issuing every launch itself (ops._BLOCK_NATIVE = False) --
    weights, Adam moments, teacher, centre and the logged scalars; also under gradient accumulation and --grad-checkpoint."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    kw = dict(img_size=112, patch=16, dim=384, depth=3, heads=6, num_registers=4, scale_aware=True, use_grad_checkpoint=ckpt)
    g = torch.Generator().manual_seed(31)
    B = 12
    batch = torch.randn(2 * B, 3, 112, 112, generator=g).to(DEV)
    sp = (torch.rand(2 * B, 3, generator=g) + 0.5).to(DEV)

    def run(native):
        torch.manual_seed(5)
        s_ = arch.DinoStudentTeacher(arch.PatchViT(**kw), 512)
        torch.nn.init.xavier_uniform_(s_.backbone.scale_embed.mlp[2].weight)
        t_ = arch.DinoStudentTeacher(arch.PatchViT(**kw), 512)
        t_.load_state_dict(s_.state_dict())
        was = ops._BLOCK_NATIVE
        ops._BLOCK_NATIVE = native
        try:
            eng = TrainEngine(s_.to(DEV), t_.to(DEV), 512, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99, koleo_weight=0.1),
                              amp_dtype=torch.bfloat16, accumulation_steps=accum)
            s_.train()
            for _ in range(2 * accum):
                eng.step(batch, sp)
            sc = eng.scalars()
        finally:
            ops._BLOCK_NATIVE = was
        return eng.flat_p.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.flat_t.clone(), eng.center.clone(), sc

    a, b = run(True), run(False)
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    assert a[5] == b[5], (a[5], b[5])
    assert math.isfinite(a[5]["loss"]) and a[5]["grad_norm"] > 0


def test_vit_plain_golden_no_registers(dx):
    ops, arch = dx
    g = load_golden("vit_plain.npz")
    m = arch.PatchViT(img_size=28, patch=14, dim=32, depth=1, heads=2, mlp_ratio=2.0, num_registers=0, scale_aware=False)
    assert not hasattr(m, "scale_embed") and not hasattr(m, "registers")
    m = _load(m, sub(g, "w")).eval()
    with torch.no_grad():
        y = m(t(g["x"]).to(DEV))
    assert y.shape == (3, 5, 32)
    close(y, g["y"], 1e-4, 1e-5, "vit_plain")


def test_backward_populates_every_scale_embed_grad(dx):
    """reference tests/test_scale_embedding.py:331-348."""
    _, arch = dx
    torch.manual_seed(0)
    m = arch.DinoStudentTeacher(arch.PatchViT(56, 14, 64, 2, 2, scale_aware=True), 128).to(DEV)
    torch.nn.init.xavier_uniform_(m.backbone.scale_embed.mlp[2].weight)
    out = m(torch.randn(2, 3, 56, 56, device=DEV), spacing=torch.tensor([[0.5, 0.5, 1.0], [1.5, 1.5, 5.0]], device=DEV))
    assert out.shape == (2, 128)
    out.pow(2).mean().backward()
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if "scale_embed" in n:
            assert p.grad.abs().sum() > 0, n


# ------------------------------------------------------------------------------------------ optimiser tail + step
def test_adamw_ema_vs_np(dx):
    ops, _ = dx
    from oracle import kernels_np as K
    rng = np.random.default_rng(3)
    n = 10007
    p, g_, m, v, pt = rng.normal(size=n), rng.normal(size=n), rng.normal(size=n) * 0.1, np.abs(rng.normal(size=n)) * 0.01, rng.normal(size=n)
    pn, mn, vn, ptn, gsq = K.adamw_ema(p, 0.5 * g_, m, v, pt, 3, 2e-3, 0.04, 0.9, 0.999, 1e-8, 0.996)
    n_pad = (n + 3) // 4 * 4
    def dev(a):
        z = torch.zeros(n_pad, dtype=torch.float32, device=DEV)
        z[:n] = torch.tensor(a, dtype=torch.float32)
        return z
    P, G, M, V, T = dev(p), dev(g_), dev(m), dev(v), dev(pt)
    out = ops.adamw_ema_(P, G, M, V, T, lr=2e-3, weight_decay=0.04, beta1=0.9, beta2=0.999, eps=1e-8, step_t=3, ema=0.996, grad_scale=0.5)
    close(P[:n], pn, 1e-5, 1e-6, "p"); close(M[:n], mn, 1e-5, 1e-6, "m"); close(V[:n], vn, 1e-5, 1e-7, "v"); close(T[:n], ptn, 1e-5, 1e-6, "teacher")
    assert float(out) == pytest.approx(gsq, rel=1e-5)
    assert float(ops.sumsq(G)) == pytest.approx(float((g_ * g_).sum()), rel=1e-5)


def test_three_training_steps_golden(dx):
    """The engine's step() against 3 consecutive steps of the reference loop (fp32 parity mode)."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    g = load_golden("step_tiny.npz")
    cfg, out_dim = _cfg(g["cfg"])
    lr, min_lr, warm, max_steps, wd, ema, ts, tt, cm, gw = [float(v) for v in g["hp"]]
    hp = StepHyperParams(lr=lr, min_lr=min_lr, warmup_steps=int(warm), max_steps=int(max_steps), weight_decay=wd, ema=ema,
                         student_temp=ts, teacher_temp=tt, center_momentum=cm, gram_weight=gw)
    student = _load(arch.DinoStudentTeacher(arch.PatchViT(**cfg), out_dim), sub(g, "init"))
    teacher = _load(arch.DinoStudentTeacher(arch.PatchViT(**cfg), out_dim), sub(g, "init"))
    eng = TrainEngine(student, teacher, out_dim, hp)
    for step in range(3):
        eng.step(t(g[f"batch{step}"]).to(DEV), t(g[f"spacing{step}"]).to(DEV))
        r = eng.scalars()
        assert r["loss"] == pytest.approx(float(g["losses"][step]), rel=1e-3)
        assert r["dino"] == pytest.approx(float(g["dinos"][step]), rel=1e-3)
        assert r["gram"] == pytest.approx(float(g["grams"][step]), rel=1e-3, abs=1e-9)
        assert r["grad_norm"] == pytest.approx(float(g["grad_norms"][step]), rel=1e-3)
        assert r["lr"] == pytest.approx(float(g["lrs"][step]), rel=1e-12)
    # Updated weights.  Adam turns a numerically-zero gradient (|g| < 1e-6: the key bias, to which softmax is invariant, and most of the
    # scale-embed input layer) into a +-lr move whose sign is round-off; THOSE elements -- named by the oracle's own gradients of the
    # three steps, the mask tests/test_oracle_golden.py::test_three_training_steps pins against the reference -- are compared at the
    # scale 3 steps x lr allow.  Every other element must sit within 1e-3 of the reference's value (round 2 allowed any 3 % of the
    # elements to be off by 6.5e-3).
    from oracle import dinox_oracle as O
    ocfg = O.VitCfg(out_dim=out_dim, **cfg)
    ohp = O.HyperParams(lr=lr, min_lr=min_lr, warmup_steps=int(warm), max_steps=int(max_steps), weight_decay=wd, ema=ema, student_temp=ts,
                        teacher_temp=tt, center_momentum=cm, gram_weight=gw)
    ost = O.init_state(ocfg, sub(g, "init"))
    noisy = {}
    for step in range(3):
        r = O.train_step(ost, t(g[f"batch{step}"]), t(g[f"spacing{step}"]), ohp)
        for k, gr in r["grads"].items():
            noisy[k] = noisy.get(k, torch.zeros_like(gr, dtype=torch.bool)) | (gr.abs() < 1e-6)
    lr_sum = float(sum(float(v) for v in g["lrs"][:3]))
    for name, have, want_sd, tol_noisy in (("student", student.state_dict(), sub(g, "student3"), 2.1 * lr_sum),
                                           ("teacher", teacher.state_dict(), sub(g, "teacher3"), 2.1 * lr_sum * (1 - ema) * 3)):
        n_tight = n_bad = 0
        for k, v in want_sd.items():
            d = (have[k].cpu().double() - v.double()).abs()
            m = noisy[k]
            if m.any():
                assert float(d[m].max()) <= tol_noisy, (name, k, float(d[m].max()))
            tight = d[~m]
            n_tight += tight.numel()
            n_bad += int((tight > 1e-3 * v.double().abs()[~m] + 2e-5).sum())
        assert n_bad <= 1e-4 * n_tight, (name, n_bad, n_tight)
    close(eng.center, g["center3"], 1e-3, 1e-6, "center3")


def test_cpu_tensors_fail_loudly(dx):
    _, arch = dx
    m = arch.PatchViT(28, 14, 32, 1, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 3, 28, 28))


# ------------------------------------------------------------------------------------------ CLI end to end
def _cli():
    import importlib.util, os, sys
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("phase5_big_run", os.path.join(PKG, "scripts", "phase5_big_run.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_cli_train_checkpoint_resume(dx, tmp_path, capsys):
    """scripts/phase5_big_run.py drop-in: short synthetic run with --log-json, periodic + final checkpoints, then
    --resume auto continues from the saved step with the saved optimiser state (reference integration_canary style:
    loss continuity; same-seed reruns reproduce the per-step losses)."""
    import json
    cli = _cli()
    common = ["--config", "vit-tiny", "--vit-patch", "16", "--vit-dim", "64", "--vit-depth", "2", "--vit-heads", "2", "--out-dim", "256",
              "--img-size", "32", "--batch-size", "8", "--scale-aware", "--amp", "--synthetic", "32", "--num-workers", "0",
              "--warmup-steps", "2", "--lr", "1e-3", "--ckpt-every", "3", "--koleo-weight", "0.1", "--grad-checkpoint",
              "--run-dir", str(tmp_path / "runs")]
    log1 = tmp_path / "a.jsonl"
    cli.main(common + ["--max-steps", "6", "--log-json", str(log1), "--run-suffix", "a"])
    out = capsys.readouterr().out
    for key in ("model_config=custom", "device=cuda", "run_dir=", "checkpoint_saved=", "final_checkpoint=", "step=     0"):
        assert key in out, key
    lines = [json.loads(l) for l in log1.read_text().splitlines()]
    assert [l["step"] for l in lines] == list(range(6)) and all(set(l) == {"step", "loss", "lr"} for l in lines)
    assert all(np.isfinite(l["loss"]) for l in lines) and lines[0]["lr"] == pytest.approx(1e-3 * 1 / 2)
    run = sorted((tmp_path / "runs").iterdir())[-1]
    names = sorted(p.name for p in run.glob("*.pth"))
    assert names == ["checkpoint_00000003.pth", "checkpoint_00000006.pth", "checkpoint_final_00000006.pth"]
    assert json.loads((run / "config.json").read_text())["model"]["dim"] == 64
    # same seed, fresh run: identical data order and init -> the reference canary allows 0.5 % per-step difference
    log2 = tmp_path / "b.jsonl"
    cli.main(common + ["--max-steps", "6", "--log-json", str(log2), "--run-suffix", "b", "--run-dir", str(tmp_path / "runs2")])
    for a, b in zip(lines, [json.loads(l) for l in log2.read_text().splitlines()]):
        assert a["loss"] == pytest.approx(b["loss"], rel=5e-3)
    # resume: picks the latest checkpoint of the latest run, continues at step 6 with restored Adam state
    log3 = tmp_path / "c.jsonl"
    cli.main(common + ["--max-steps", "9", "--log-json", str(log3), "--resume", "auto"])
    out = capsys.readouterr().out
    assert "resumed_from_step=6" in out
    cont = [json.loads(l) for l in log3.read_text().splitlines()]
    assert [l["step"] for l in cont] == [6, 7, 8]
    assert 0.25 < cont[0]["loss"] / lines[-1]["loss"] < 3.0          # reference canary's continuity band
    # the trained backbone loads through zoo.hub and encodes on the HIP path
    import zoo.hub as hub, zoo.encode as enc
    bb = hub.load_model(str(run / "checkpoint_final_00000006.pth"), device=DEV, config_override={"patch": 16, "num_registers": 4})
    f = enc.encode(bb, np.random.default_rng(0).normal(40, 200, size=(48, 48)).astype(np.float32), pixel_spacing=(0.7, 0.7), slice_thickness=2.0)
    assert f.shape == (1, 1, 64) and torch.isfinite(f).all()
    fb = enc.encode_batch(bb, [np.zeros((40, 40), np.float32), np.ones((3, 40, 40), np.float32)], [(0.5, 0.5, 1.0), (1.0, 1.0, 3.0)])
    assert fb.shape == (2, 1, 64)


# ------------------------------------------------------------------------------------------ data parallel on a real device
@pytest.mark.parametrize("accum,ckpt,crops", [(1, False, 0), (2, False, 0), (1, True, 0), (1, False, 3)])
def test_engine_data_parallel_two_ranks_match_single_process(dx, tmp_path, accum, ckpt, crops):
    """Two ranks (gloo, both on this GPU -- RCCL needs one GPU per rank) run TrainEngine.step on their shard of a
    global batch; the result must equal the single-process step at the global batch: same loss, same
    updated weights, same centre.  Exercises broadcast, bucket hooks, centre all-reduce and the 1/world AdamW scale.
    accum = 2: gradient accumulation x data parallel (gradients are exchanged on the last micro-batch only).
    ckpt: --grad-checkpoint under DP -- the buckets must still be launched DURING backward.
    crops = 3: local crops on a model that is NOT scale-aware (cls / pos / registers then close the last bucket): pos_embed is reached
    twice, directly by the global views and through the interpolated grid of the local ones; its slice of the arena may be exchanged
    only when both gradients have landed (the ranks' position embeddings diverged silently)."""
    import os, socket, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(ROOT, "tests", "_dp_gpu_worker.py")
    outs = [str(tmp_path / f"r{r}.pt") for r in range(2)]
    knobs = dict(DINOX_TEST_ACCUM=str(accum), DINOX_TEST_GRAD_CKPT="1" if ckpt else "", DINOX_TEST_LOCAL_CROPS=str(crops),
                 DINOX_TEST_SCALE_AWARE="0" if crops else "1")
    os.environ.update(knobs)
    env = dict(os.environ, DINOX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, worker, outs[r]], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=240)[0].decode(errors="replace")[-1500:] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    single = subprocess.run([sys.executable, worker, str(tmp_path / "single.pt")], env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert single.returncode == 0, single.stdout.decode(errors="replace")[-1500:]
    for k in knobs:
        os.environ.pop(k, None)
    a, b, ref = torch.load(outs[0]), torch.load(outs[1]), torch.load(tmp_path / "single.pt")
    assert torch.equal(a["flat_p"], b["flat_p"]) and torch.equal(a["center"], b["center"])        # ranks stay in lock-step
    assert a["buckets"] >= 3 and a["fired_in_backward"] >= a["buckets"] - 1, a       # overlapped, not left to finish()
    mean_loss = 0.5 * (a["loss"] + b["loss"])
    assert mean_loss == pytest.approx(ref["loss"], rel=2e-4)
    assert a["grad_norm"] == pytest.approx(ref["grad_norm"], rel=2e-3)
    close(a["center"], ref["center"], 1e-4, 1e-7, "centre")
    d = (a["flat_p"] - ref["flat_p"]).abs()
    assert float((d <= 1e-5 + 1e-4 * ref["flat_p"].abs()).double().mean()) > 0.995     # Adam sign-noise on ~zero grads aside
    assert float(d.max()) <= 2.5e-3


# ------------------------------------------------------------------------------------------ full-size model vs the oracle
def test_full_vit_small_16_step_matches_oracle(dx):
    """BASELINE's model (ViT-S/16, 224 px, 12 blocks, 201 tokens, out 8192, scale-aware) at a batch the CPU oracle finishes
    in seconds (B=2 -> 4 views): one whole optimiser step in fp32 parity mode must meet the north-star 1e-3 gate on loss,
    both loss terms, the global grad-norm and the updated weights; the bf16 throughput mode is reported against the
    same oracle with its own (bf16-sized) band."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    from oracle import dinox_oracle as O
    cfg = O.VitCfg(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True, out_dim=8192)
    sd = O.random_params(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    B = 2
    batch = torch.randn(2 * B, 3, 224, 224, generator=g)
    sp = torch.rand(B, 3, generator=g) * 2 + 0.4
    sp2 = torch.cat([sp, sp], 0)
    hp_o = O.HyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99)
    st = O.init_state(cfg, sd)
    st.teacher = {k: v + 0.01 * torch.randn(v.shape, generator=g) for k, v in st.teacher.items()}   # teacher != student: Gram loss > 0
    teacher_sd = {k: v.clone() for k, v in st.teacher.items()}
    want = O.train_step(st, batch, sp2, hp_o)
    kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)

    def run(amp):
        student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
        teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
        student.load_state_dict(sd)
        teacher.load_state_dict(teacher_sd)
        eng = TrainEngine(student.to(DEV), teacher.to(DEV), 8192, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99), amp_dtype=amp)
        eng.step(batch.to(DEV), sp2.to(DEV))
        return eng.scalars(), student.state_dict(), {n: p.grad.detach().clone() for (n, _), p in zip(student.named_parameters(), eng.params)}

    got, ssd, _ = run(None)
    for k in ("loss", "dino", "gram", "grad_norm"):
        assert got[k] == pytest.approx(want[k], rel=1e-3), (k, got[k], want[k])
    # updated weights after the lr = 1e-3 Adam step.  Adam's first step moves an element by lr * g / (|g| + 1e-8): where the
    # gradient is numerically zero (|g| < 1e-6: key bias, most of the scale-embed input layer) the move is round-off-signed
    # and only bounded by lr; everywhere else the HIP result must sit within 1e-3 of the oracle's.
    worst_noisy, n_tight, n_bad = 0.0, 0, 0
    for k, v in st.student.items():
        d = (ssd[k].cpu().double() - v.double()).abs()
        noisy = want["grads"][k].abs() < 1e-6
        if noisy.any():
            worst_noisy = max(worst_noisy, float(d[noisy].max()))
        tight = d[~noisy]
        n_tight += tight.numel()
        n_bad += int((tight > 1e-3 * v.double().abs()[~noisy] + 2e-5).sum())
    assert worst_noisy <= 2.1e-3, worst_noisy
    assert n_bad <= 1e-4 * n_tight, (n_bad, n_tight)
    # bf16 throughput mode (the mode of every bench number): per parameter within 1.5x of the distance the reference's own --amp
    # arithmetic (oracle.autocast_bf16 == the reference under torch.autocast, tests/test_oracle_golden.py) keeps from fp32
    st_a = O.init_state(cfg, sd)
    st_a.teacher = {k: v.clone() for k, v in teacher_sd.items()}
    amp = O.train_step(st_a, batch, sp2, hp_o, amp=True)
    got16, _, g16 = run(torch.bfloat16)
    worst = assert_within_autocast_distance(g16, want["grads"], amp["grads"], "ViT-S/16 parameter gradients (bf16 mode)")
    print(f"ViT-S/16 bf16: worst (HIP distance / reference-autocast distance) = {worst[0]:.2f} at {worst[1]}")
    for k in ("loss", "dino", "gram", "grad_norm"):
        d_ref = abs(amp[k] - want[k])
        assert abs(got16[k] - want[k]) <= AMP_FACTOR * d_ref + 2e-3 * abs(want[k]), (k, got16[k], want[k], amp[k])


def test_gradient_accumulation_semantics(dx):
    """accumulation_steps=2 (reference semantics, phase5_big_run.py:1769-1796): with a frozen centre (momentum 1) two
    identical micro-batches average to the single-batch gradient, so after 2 micro-steps the weights equal one plain step
    taken with the LR of micro-step 1; no optimiser/EMA/weight change happens after micro-step 0."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    kw = dict(img_size=56, patch=14, dim=64, depth=2, heads=2, num_registers=4, scale_aware=True)
    g = torch.Generator().manual_seed(2)
    batch, sp = torch.randn(8, 3, 56, 56, generator=g).to(DEV), (torch.rand(8, 3, generator=g) + 0.5).to(DEV)

    def make(accum, hp):
        torch.manual_seed(9)
        s_ = arch.DinoStudentTeacher(arch.PatchViT(**kw), 128)
        torch.nn.init.xavier_uniform_(s_.backbone.scale_embed.mlp[2].weight)
        t_ = arch.DinoStudentTeacher(arch.PatchViT(**kw), 128)
        t_.load_state_dict(s_.state_dict())
        return TrainEngine(s_.to(DEV), t_.to(DEV), 128, hp, accumulation_steps=accum)

    hp2 = StepHyperParams(lr=1e-3, warmup_steps=4, max_steps=20, ema=0.9, center_momentum=1.0)
    e2 = make(2, hp2)
    p0 = e2.flat_p.clone()
    e2.step(batch, sp)
    assert torch.equal(e2.flat_p, p0) and e2.opt_steps == 0 and torch.equal(e2.flat_t, p0)
    r = e2.step(batch, sp)
    assert e2.opt_steps == 1 and e2.step_count == 2 and r["lr"] == pytest.approx(1e-3 * 2 / 4)
    hp1 = StepHyperParams(lr=1e-3 * 2 / 4, warmup_steps=0, max_steps=None, ema=0.9, center_momentum=1.0)   # constant LR = micro-step 1's
    e1 = make(1, hp1)
    e1.step(batch, sp)
    assert float(e2.last["loss"]) == pytest.approx(float(e1.last["loss"]), rel=1e-6)
    d = (e2.flat_p - e1.flat_p).abs()
    assert float((d <= 1e-6 + 1e-4 * e1.flat_p.abs()).double().mean()) > 0.995 and float(d.max()) <= 1.1e-3
    assert float((e2.flat_t - e1.flat_t).abs().max()) <= 1.2e-4


def test_arena_shadow_and_multi_transpose(dx):
    """ops.ArenaShadow: one cast launch images a whole parameter arena, one launch transposes every requested matrix; both must
    equal the per-weight casts bit for bit, and a parameter modified after the refresh must not be served."""
    ops, arch = dx
    from dinox.engine import flatten_parameters
    torch.manual_seed(3)
    mod = torch.nn.ModuleList([torch.nn.Linear(40, 72), torch.nn.Linear(72, 33, bias=False), torch.nn.Conv2d(3, 24, 5)]).to(DEV)
    flat, params, offs = flatten_parameters(mod)
    assert all(o % 8 == 0 for o in offs)
    sh = ops.ArenaShadow(flat, params, offs)
    mats = [p for p in params if p.dim() >= 2]
    for w in mats:
        assert sh.get(w, True) is None and sh.get(w, False) is None      # nothing imaged yet; transposes are now on the wanted list
    sh.refresh()
    for w in mats:
        w2 = w.detach().reshape(w.shape[0], -1)
        assert torch.equal(sh.get(w, False), w2.bfloat16())
        assert torch.equal(sh.get(w, True), w2.t().contiguous().bfloat16())
        assert sh.get(w, False).data_ptr() % 16 == 0 and sh.get(w, True).data_ptr() % 16 == 0
    with torch.no_grad():
        mats[0].mul_(2.0)                                                # version bump: the image is stale
    assert sh.get(mats[0], False) is None and sh.get(mats[1], False) is not None
    sh.refresh()
    assert torch.equal(sh.get(mats[0], False), mats[0].detach().bfloat16())


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_grad_sink_accumulates_in_place(dx, mode):
    """ops._GradSink: with a registered arena the dW product (and its bias column sums) lands in p.grad directly and autograd gets
    None; the result must equal the autograd-accumulated gradient, including accumulation over two backward passes."""
    ops, arch = dx
    from dinox.engine import flatten_parameters
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    torch.manual_seed(5)
    lin = torch.nn.Sequential(arch.LayerNorm(64), arch.Linear(64, 136)).to(DEV)      # LayerNorm's affine gradients use the sink too
    with torch.no_grad():
        lin[0].weight.add_(0.1 * torch.randn(64, device=DEV))
    x = torch.randn(300, 64, device=DEV)

    def run(sink):
        for p in lin.parameters():
            p.grad = None
        flat, params, offs = flatten_parameters(lin)
        g = torch.zeros_like(flat)
        for p, o in zip(params, offs):
            p.grad = g[o:o + p.numel()].view(p.shape)
        events = []
        if sink:
            ops.grad_sink.register("test", params, events.append)
        else:
            ops.grad_sink.clear()
        try:
            with ops.compute_dtype(dt):
                for _ in range(2):
                    ops.grad_sink.uses.clear()
                    lin(x).float().square().sum().backward()
        finally:
            ops.grad_sink.clear()
        return g.clone(), events

    ref, _ = run(False)
    got, events = run(True)
    assert sorted(events) == [0, 0, 1, 1, 2, 2, 3, 3]                    # every parameter announced once per backward
    assert rel_l2(got, ref) < (1e-6 if mode == "fp32" else 1e-5)


@pytest.mark.parametrize("tag", ["small", "mm"])
def test_koleo_loss_golden(dx, tag):
    """ops.koleo_loss vs the reference KoLeoLoss fixture (phase5_big_run.py:742-773): loss and input gradient.  Tolerances as
    in tests/test_oracle_golden.py: the 40-row fixture went through cdist's matmul route, which loses digits on its close pair."""
    ops, _ = dx
    g = load_golden("koleo_loss.npz")
    x = t(g[f"{tag}_x"]).to(DEV).requires_grad_(True)
    l = ops.koleo_loss(x)
    l.backward()
    assert float(l) == pytest.approx(float(g[f"{tag}_loss"]), rel=1e-5 if tag == "small" else 5e-4, abs=1e-6)
    assert rel_l2(x.grad, g[f"{tag}_dx"]) < (2e-5 if tag == "small" else 1e-3)


def test_koleo_loss_vs_oracle_and_edges(dx):
    """Head-sized rows (512 x 8192) against the oracle; then the edge cases: an exact duplicate pair (distance 0: loss term
    -log(eps), no gradient through that pair, as cdist's backward gives), an all-zero row (F.normalize's eps clamp), two rows."""
    ops, _ = dx
    from oracle import dinox_oracle as O
    g = torch.Generator().manual_seed(11)
    x = torch.randn(96, 8192, generator=g)
    xo = x.clone().requires_grad_(True)
    lo = O.koleo_loss(xo)
    lo.backward()
    xg = x.to(DEV).requires_grad_(True)
    lg = ops.koleo_loss(xg)
    (3.0 * lg).backward()
    assert float(lg) == pytest.approx(float(lo), rel=1e-5)
    assert rel_l2(xg.grad, 3.0 * xo.grad) < 1e-4
    # duplicate pair + zero row
    y = torch.randn(12, 64, generator=g)
    y[5] = y[2]
    y[9] = 0.0
    yg = y.to(DEV).requires_grad_(True)
    l = ops.koleo_loss(yg)
    l.backward()
    assert float(l) == pytest.approx(float(O.koleo_loss(y)), rel=1e-5)            # includes two -log(1e-8) terms
    assert torch.isfinite(yg.grad).all()
    # rows 2 and 5 only receive gradient as someone else's neighbour, never through their own zero distance
    y2 = torch.randn(2, 16, generator=g)
    y2g = y2.to(DEV).requires_grad_(True)
    l2 = ops.koleo_loss(y2g)
    l2.backward()
    y2o = y2.clone().requires_grad_(True)
    O.koleo_loss(y2o).backward()
    assert float(l2) == pytest.approx(float(O.koleo_loss(y2)), rel=1e-6) and rel_l2(y2g.grad, y2o.grad) < 1e-5


def test_step_with_koleo_matches_oracle(dx):
    """One engine step with --koleo-weight 0.1 (the value of every production run, docs/EXPERIMENTS.md) against the oracle's
    train_step on the tiny scale-aware model."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    from oracle import dinox_oracle as O
    cfg = O.VitCfg(img_size=56, patch=14, dim=64, depth=2, heads=2, num_registers=4, scale_aware=True, out_dim=128)
    st = O.init_state(cfg, O.random_params(cfg, seed=4))
    g = torch.Generator().manual_seed(6)
    batch = torch.randn(12, 3, 56, 56, generator=g)
    sp2 = torch.rand(12, 3, generator=g) + 0.5
    hp_o = O.HyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99, koleo_weight=0.1)
    sd = {k: v.clone() for k, v in st.student.items()}
    want = O.train_step(st, batch, sp2, hp_o)
    kw = dict(img_size=56, patch=14, dim=64, depth=2, heads=2, num_registers=4, scale_aware=True)
    student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 128)
    teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 128)
    student.load_state_dict(sd)
    teacher.load_state_dict(sd)
    eng = TrainEngine(student.to(DEV), teacher.to(DEV), 128, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99, koleo_weight=0.1))
    eng.step(batch.to(DEV), sp2.to(DEV))
    got = eng.scalars()
    assert got["koleo"] != 0.0
    for k in ("loss", "grad_norm"):
        assert got[k] == pytest.approx(want[k], rel=1e-3), (k, got[k], want[k])
    names = [n for n, _ in student.named_parameters()]
    for n, p in zip(names, eng.params):
        ref = want["grads"][n]
        if float(ref.abs().max()) > 1e-6:
            assert rel_l2(p.grad, ref) < 2e-3, n


def test_grad_checkpoint_matches_plain(dx):
    """--grad-checkpoint (zoo/arch.py:232-233): recomputing every block in backward must give the gradients of the plain run."""
    ops, arch = dx
    kw = dict(img_size=56, patch=14, dim=64, depth=3, heads=2, num_registers=4, scale_aware=True)
    g = torch.Generator().manual_seed(8)
    x, sp = torch.randn(4, 3, 56, 56, generator=g).to(DEV), (torch.rand(4, 3, generator=g) + 0.5).to(DEV)
    grads = []
    for ck in (False, True):
        torch.manual_seed(12)
        m = arch.PatchViT(use_grad_checkpoint=ck, **kw).to(DEV).train()
        m(x, spacing=sp).square().mean().backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        assert torch.allclose(grads[0][n], grads[1][n], rtol=1e-5, atol=1e-8), n


# ------------------------------------------------------------------------------------------ multi-crop extension
def test_dino_ce_multi_reduces_to_reference_two_view_loss(dx):
    """The multi-crop cross-entropy with no local crops must BE the reference-pinned 2-view loss (same value, same gradient)."""
    ops, _ = dx
    g = load_golden("dino_loss.npz")
    s = t(g["s"]).to(DEV).requires_grad_(True)
    tt, c = t(g["t"]).to(DEV), t(g["center0"]).to(DEV)
    l = ops.DinoCEMultiFn.apply(s, tt, c, 0.1, 0.04, 2)
    l.backward()
    assert float(l) == pytest.approx(float(g["loss1"]), rel=1e-5)
    close(s.grad, g["ds1"], 1e-4, 1e-8, "ds (multi, L=0)")


def test_dino_ce_multi_vs_oracle(dx):
    ops, _ = dx
    from oracle import dinox_oracle as O
    g = torch.Generator().manual_seed(13)
    B, K, G, L = 5, 1000, 2, 3
    s = (3 * torch.randn((G + L) * B, K, generator=g))
    tt, c = 1.5 * torch.randn(G * B, K, generator=g), 0.1 * torch.randn(1, K, generator=g)
    so = s.clone().requires_grad_(True)
    lo = O.dino_loss_multicrop(so, tt, c, 0.1, 0.04)
    lo.backward()
    sg = s.to(DEV).requires_grad_(True)
    lg = ops.DinoCEMultiFn.apply(sg, tt.to(DEV), c.to(DEV), 0.1, 0.04, 2)
    (2.0 * lg).backward()
    assert float(lg) == pytest.approx(float(lo), rel=1e-5)
    assert rel_l2(sg.grad, 2.0 * so.grad) < 1e-5


@pytest.mark.parametrize("g_in,g_out", [(14, 6), (4, 2), (4, 9)])
def test_pos_interp_matches_torch_bicubic(dx, g_in, g_out):
    """ops.interp_pos == F.interpolate(bicubic, align_corners=False) on the patch grid, CLS entry untouched; backward is the
    transposed map (checked against autograd through F.interpolate)."""
    import torch.nn.functional as F
    ops, _ = dx
    g = torch.Generator().manual_seed(g_in * 10 + g_out)
    D = 24
    pos = torch.randn(1, 1 + g_in * g_in, D, generator=g)
    up = torch.randn(1, 1 + g_out * g_out, D, generator=g)
    pr = pos.clone().requires_grad_(True)
    grid = pr[0, 1:].reshape(g_in, g_in, D).permute(2, 0, 1)[None]
    ref = torch.cat([pr[:, :1], F.interpolate(grid, size=(g_out, g_out), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).reshape(1, -1, D)], 1)
    (ref * up).sum().backward()
    pg = pos.to(DEV).requires_grad_(True)
    out = ops.interp_pos(pg, g_out)
    (out * up.to(DEV)).sum().backward()
    close(out, ref.detach(), 1e-5, 2e-6, "interp fwd")
    close(pg.grad, pr.grad, 1e-5, 2e-6, "interp bwd")


def test_step_multicrop_matches_oracle(dx):
    """One engine step with 2 global (56 px) + 3 local (28 px) views per sample against the oracle's statement of the DINO-paper
    multi-crop step: loss terms, grad-norm, every parameter gradient.  Parameters are used by two forward passes here, which is
    also what exercises the gradient sink's use counting."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    from oracle import dinox_oracle as O
    cfg = O.VitCfg(img_size=56, patch=14, dim=64, depth=2, heads=2, num_registers=4, scale_aware=True, out_dim=128)
    st = O.init_state(cfg, O.random_params(cfg, seed=8))
    g = torch.Generator().manual_seed(9)
    B, L = 4, 3
    batch = torch.randn(2 * B, 3, 56, 56, generator=g)
    locs = torch.randn(L * B, 3, 28, 28, generator=g)
    sp = torch.rand(B, 3, generator=g) + 0.5
    sp2, spl = torch.cat([sp, sp], 0), torch.cat([sp] * L, 0)
    hp_o = O.HyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99)
    sd = {k: v.clone() for k, v in st.student.items()}
    st.teacher = {k: v + 0.01 * torch.randn(v.shape, generator=g) for k, v in st.teacher.items()}
    tsd = {k: v.clone() for k, v in st.teacher.items()}
    want = O.train_step(st, batch, sp2, hp_o, local_batch=locs, local_spacing=spl)
    kw = dict(img_size=56, patch=14, dim=64, depth=2, heads=2, num_registers=4, scale_aware=True)
    student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 128)
    teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 128)
    student.load_state_dict(sd)
    teacher.load_state_dict(tsd)
    eng = TrainEngine(student.to(DEV), teacher.to(DEV), 128, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99))
    eng.step(batch.to(DEV), sp2.to(DEV), local_batch=locs.to(DEV), local_spacing=spl.to(DEV))
    got = eng.scalars()
    for k in ("loss", "dino", "gram", "grad_norm"):
        assert got[k] == pytest.approx(want[k], rel=1e-3), (k, got[k], want[k])
    names = [n for n, _ in student.named_parameters()]
    for n, p in zip(names, eng.params):
        ref = want["grads"][n]
        if float(ref.abs().max()) > 1e-6:
            assert rel_l2(p.grad, ref) < 2e-3, n
    # without local crops the same engine gives the plain reference step (the extension changes nothing by being there)
    eng2 = TrainEngine(student.to(DEV), teacher.to(DEV), 128, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99))
    assert torch.isfinite(eng2.step(batch.to(DEV), sp2.to(DEV))["loss"])


def test_import_order_independent_device_visibility():
    """`import dinox` ahead of `import torch` must still see the GPU (PyTorch-ROCm bundles its own HIP runtime; the library has
    to bind to that one, dinox/_lib.py) -- this is the order __graft_entry__.build() followed by smoke() produces.  (The test does not
    run build() itself: compiling is not what it checks.)"""
    import subprocess, sys
    from conftest import PKG, ROOT
    code = ("import sys; sys.path[:0] = [%r, %r]; import dinox; from dinox import _lib; import torch; "
            "assert torch.cuda.is_available(); assert _lib.lib.dinox_device_ok() == 1, _lib.last_error(); print('ok')") % (ROOT, PKG)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stdout.decode(errors="replace")[-1500:]


def test_rccl_call_surface_world1(tmp_path):
    """Every collective of the step (broadcast of the arenas, bucketed async all-reduce launched from backward, centre all-reduce,
    KoLeo all-gathers, barrier) goes through the real RCCL backend ("nccl") in a world of ONE rank (a one-GPU box cannot hold two
    RCCL ranks): the result must equal the run without a process group -- a sum over one rank is the identity."""
    import os, socket, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(ROOT, "tests", "_dp_gpu_worker.py")
    base = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.pop("DINOX_DIST_BACKEND", None)
    outs = {}
    for tag, extra in (("plain", {}), ("rccl", {"DINOX_DP_FORCE_COLLECTIVES": "1", "DINOX_EXPECT_BACKEND": "nccl"})):
        out = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, worker, out], env=dict(base, **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        outs[tag] = torch.load(out)
    a, b = outs["plain"], outs["rccl"]
    # bit for bit: a sum over one rank is the identity and (since round 2) no kernel of the step leaves a summation order to the
    # scheduler -- round 1 had to allow 2e-4 here because the split-K dW products met in fp32 atomics
    assert a["loss"] == b["loss"] and a["grad_norm"] == b["grad_norm"]
    assert torch.equal(a["center"], b["center"]) and torch.equal(a["flat_p"], b["flat_p"])


@pytest.mark.parametrize("name", ["configs0_vit_tiny_cifar", "configs1_vit_small_not_scale_aware"])
def test_baseline_secondary_configs_step_matches_oracle(dx, name):
    """BASELINE.json's other configurations as parity cases (one optimiser step, fp32 parity mode, against the oracle):
    configs[0] = the CIFAR plumbing model (ViT-Tiny dim 192 / depth 12 / heads 3 / out 4096, 32 px, patch 4 -> 64 patches + CLS
    + 4 registers, no scale embedding, Gram term off as in scripts/baseline_cifar10_pretrain.py);
    configs[1] = ViT-S/16 at 224 px with scale-aware OFF.  Batches are cut to what the CPU oracle steps in seconds."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    from oracle import dinox_oracle as O
    if name.startswith("configs0"):
        kw = dict(img_size=32, patch=4, dim=192, depth=12, heads=3, num_registers=4, scale_aware=False)
        out_dim, B, gram_w = 4096, 4, 0.0
    else:
        kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=False)
        out_dim, B, gram_w = 8192, 2, 1.0
    cfg = O.VitCfg(out_dim=out_dim, **kw)
    st = O.init_state(cfg, O.random_params(cfg, seed=21))
    g = torch.Generator().manual_seed(22)
    batch = torch.randn(2 * B, 3, kw["img_size"], kw["img_size"], generator=g)
    st.teacher = {k: v + 0.01 * torch.randn(v.shape, generator=g) for k, v in st.teacher.items()}
    sd, tsd = {k: v.clone() for k, v in st.student.items()}, {k: v.clone() for k, v in st.teacher.items()}
    want = O.train_step(st, batch, None, O.HyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99, gram_weight=gram_w))
    student = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim)
    teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim)
    student.load_state_dict(sd)
    teacher.load_state_dict(tsd)
    eng = TrainEngine(student.to(DEV), teacher.to(DEV), out_dim,
                      StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99, gram_weight=gram_w))
    eng.step(batch.to(DEV), None)
    got = eng.scalars()
    for k in ("loss", "dino", "grad_norm") + (("gram",) if gram_w else ()):
        assert got[k] == pytest.approx(want[k], rel=1e-3), (k, got[k], want[k])
    names = [n for n, _ in student.named_parameters()]
    worst = 0.0
    for n, p in zip(names, eng.params):
        ref = want["grads"][n]
        if float(ref.abs().max()) > 1e-6:
            worst = max(worst, rel_l2(p.grad, ref))
    assert worst < 2e-3, worst


def test_vit_large_16_step_matches_oracle(dx):
    """BASELINE configs[4]'s model (ViT-L/16: dim 1024, 24 blocks, 16 heads, hidden 4096, 201 tokens, scale-aware, Gram on) for
    one optimiser step at the smallest batch (1 sample = 2 views) against the CPU oracle, fp32 parity mode: loss terms, grad-norm
    and every parameter gradient.  Exercises the K = 1024 / 4096 GEMM shapes, 16-head attention and the 1024-wide LayerNorm."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    from oracle import dinox_oracle as O
    kw = dict(img_size=224, patch=16, dim=1024, depth=24, heads=16, num_registers=4, scale_aware=True)
    cfg = O.VitCfg(out_dim=8192, **kw)
    st = O.init_state(cfg, O.random_params(cfg, seed=31))
    g = torch.Generator().manual_seed(32)
    batch = torch.randn(2, 3, 224, 224, generator=g)
    sp = torch.rand(1, 3, generator=g) * 2 + 0.4
    sp2 = torch.cat([sp, sp], 0)
    st.teacher = {k: v + 0.01 * torch.randn(v.shape, generator=g) for k, v in st.teacher.items()}
    sd, tsd = {k: v.clone() for k, v in st.student.items()}, {k: v.clone() for k, v in st.teacher.items()}
    hp_o = O.HyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99)
    want = O.train_step(st, batch, sp2, hp_o)
    st_a = O.init_state(cfg, sd)
    st_a.teacher = {k: v.clone() for k, v in tsd.items()}
    amp = O.train_step(st_a, batch, sp2, hp_o, amp=True)           # the reference's --amp arithmetic (configs[4] is a bf16 config)
    amp = {k: amp[k] for k in ("loss", "dino", "gram", "grad_norm", "grads")}
    del st, st_a

    def run(mode):
        student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
        teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
        student.load_state_dict(sd)
        teacher.load_state_dict(tsd)
        eng = TrainEngine(student.to(DEV), teacher.to(DEV), 8192, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99),
                          amp_dtype=mode)
        eng.step(batch.to(DEV), sp2.to(DEV))
        return eng.scalars(), {n: p.grad for (n, _), p in zip(student.named_parameters(), eng.params)}

    got, grads = run(None)
    for k in ("loss", "dino", "gram", "grad_norm"):
        assert got[k] == pytest.approx(want[k], rel=1e-3), (k, got[k], want[k])
    worst = 0.0
    for n, gr in grads.items():
        ref = want["grads"][n]
        if float(ref.abs().max()) > 1e-6:
            worst = max(worst, rel_l2(gr, ref))
    assert worst < 2e-3, worst
    del grads
    got16, g16 = run(torch.bfloat16)
    w = assert_within_autocast_distance(g16, want["grads"], amp["grads"], "ViT-L/16 parameter gradients (bf16 mode)")
    print(f"ViT-L/16 bf16: worst (HIP distance / reference-autocast distance) = {w[0]:.2f} at {w[1]}")
    for k in ("loss", "dino", "gram", "grad_norm"):
        assert abs(got16[k] - want[k]) <= AMP_FACTOR * abs(amp[k] - want[k]) + 2e-3 * abs(want[k]), (k, got16[k], want[k], amp[k])


@pytest.mark.parametrize("M,N,K", [(1000, 1152, 384), (77, 40, 384), (128 * 9 + 5, 1536, 384), (4096, 384, 384), (128 * 70 + 9, 1152, 384),
                                   (1000, 384, 1536), (517, 392, 1152), (300, 256, 768), (40, 384, 3072), (128 * 30, 384, 576)])
def test_gemm_nt_areg(dx, M, N, K, monkeypatch):
    """The register-prefetch form of the NT product (csrc/gemm_bf16_areg.hip; K a multiple of 192: two blocks of six K-steps up to
    sixteen): every epilogue it takes over from the LDS-DMA kernel -- plain, bias, GELU with its GELU' side tensor, GELU' from the side
    tensor, fp32 residual -- bf16 and fp32 outputs, ragged M and N (including a wave whose rows are all past M), against fp64 on the
    same bf16 operands.  (The dispatcher sends only K <= 576 to this kernel by default: DINOX_NT_AREG_MAXK lifts that here.)"""
    ops, _ = dx
    monkeypatch.setenv("DINOX_NT_AREG_MAXK", str(1 << 30))
    g = torch.Generator().manual_seed(M + N + K)
    A, B = (torch.randn(M, K, generator=g) * 0.5).bfloat16(), (torch.randn(N, K, generator=g) * (6.0 / math.sqrt(K))).bfloat16()
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd = A.to(DEV), B.to(DEV)
    ref = A.double() @ B.double().t()
    ops.TRACE_KERNELS = []
    try:
        c = ops.gemm(Ad, Bd, out_dtype=torch.float32)
        cb = ops.gemm(Ad, Bd, bias=bias.to(DEV))
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        act = ops.gemm(Ad, Bd, bias=bias.to(DEV), gelu=True, aux=aux, auxgrad=True)
        pre_aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        act_t = ops.gemm(Ad, Bd, bias=bias.to(DEV), gelu=True, aux=pre_aux)
        y = ops.gemm(Ad, Bd, bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.float32)
        yb = ops.gemm(Ad, Bd, residual=res.to(DEV))
        d = ops.gemm(Ad, Bd, dgelu=True, aux=aux, auxgrad=True)
        d2 = ops.gemm(Ad, Bd, dgelu=True, aux=pre_aux)
        auxf = torch.randn(M, N, generator=g).to(DEV)
        d3 = ops.gemm(Ad, Bd, dgelu=True, aux=auxf, out_dtype=torch.float32)
        assert ops.TRACE_KERNELS == ["gemm_bf16_nt_areg"] * 9, ops.TRACE_KERNELS
    finally:
        ops.TRACE_KERNELS = None
    erf = lambda t: torch.erf(t / math.sqrt(2))
    gelu_grad = lambda t: 0.5 * (1 + erf(t)) + t * torch.exp(-0.5 * t * t) / math.sqrt(2 * math.pi)
    close(c, ref, 1e-5, 1e-4, "plain fp32 out")
    assert rel_l2(cb.float(), ref + bias.double()) < 3e-3
    pre = ref + bias.double()
    assert rel_l2(act.float(), 0.5 * pre * (1 + erf(pre))) < 3e-3 and rel_l2(aux.float(), gelu_grad(pre)) < 3e-3
    assert rel_l2(act_t.float(), 0.5 * pre * (1 + erf(pre))) < 3e-3 and rel_l2(pre_aux.float(), pre) < 3e-3
    close(y, pre + res.double(), 1e-5, 1e-4, "bias + residual, fp32 out")
    assert rel_l2(yb.float(), ref + res.double()) < 3e-3
    assert rel_l2(d.float(), ref * aux.float().double().cpu()) < 3e-3
    assert rel_l2(d2.float(), ref * gelu_grad(pre_aux.float().double().cpu())) < 3e-3
    close(d3, ref * gelu_grad(auxf.double().cpu()), 1e-4, 1e-3, "GELU' from an fp32 side tensor, fp32 out")


@pytest.mark.parametrize("M,K,res,bias", [(1000, 384, True, True), (128 * 5 + 17, 1536, True, True), (77, 384, True, False), (128 * 3 + 70, 384, False, True),
                                          (4096, 1152, True, True), (128 * 700 + 9, 384, True, True), (300, 608, True, True)])
@pytest.mark.parametrize("ydt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("pp", ["0", "1"])
def test_linear_residual_ln(dx, M, K, res, bias, ydt, pp, monkeypatch):
    """dinox_linear_residual_ln (csrc/gemm_bf16_rowln.hip: x = residual + a W^T + bias and y = LayerNorm(x) with the row statistics in
    one launch, width 384) against fp64 on the same bf16 operands: x to fp32 accumulation accuracy, mean / rstd / y to LayerNorm
    accuracy; ragged M (a last tile whose second row-wave is entirely past M), no residual, no bias, a row offset that makes
    E[x^2] - mean^2 cancel (|mean| >> std), both output dtypes; and bit-repeatable.  pp = DINOX_ROWLN_PP: 0 = the 128 x 384 kernel,
    1 = the full-row 208 x 384 kernel with the LayerNorm epilogue (csrc/gemm_bf16_pp384.hip: bf16 y; fp32 y stays on the former)."""
    ops, _ = dx
    monkeypatch.setenv("DINOX_ROWLN_PP", pp)
    if pp == "1" and ydt != torch.bfloat16:
        pytest.skip("fp32 y: one kernel only")
    N = 384
    g = torch.Generator().manual_seed(M + K)
    A, W = (torch.randn(M, K, generator=g) * 0.5).bfloat16(), (torch.randn(N, K, generator=g) * (3.0 / math.sqrt(K))).bfloat16()
    b = torch.randn(N, generator=g) if bias else None
    R = (torch.randn(M, N, generator=g) * 2 + 30.0 * torch.randn(M, 1, generator=g)) if res else None       # rows with a large common offset
    gamma, beta = 1 + 0.2 * torch.randn(N, generator=g), 0.3 * torch.randn(N, generator=g)
    import dinox._lib as L_
    assert L_.lib.dinox_linear_residual_ln_ok(M, N, K) == 1
    dev = lambda t_: None if t_ is None else t_.to(DEV)
    x, y, mean, rstd = ops.linear_residual_ln(dev(A), dev(W), dev(b), dev(R), dev(gamma), dev(beta), 1e-5, ydt)
    x2, y2, mean2, rstd2 = ops.linear_residual_ln(dev(A), dev(W), dev(b), dev(R), dev(gamma), dev(beta), 1e-5, ydt)
    assert torch.equal(x, x2) and torch.equal(y, y2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    xr = A.double() @ W.double().t() + (b.double() if bias else 0) + (R.double() if res else 0)
    mu = xr.mean(-1, keepdim=True)
    var = ((xr - mu) ** 2).mean(-1, keepdim=True)
    yr = (xr - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    assert x.shape == (M, N) and x.dtype == torch.float32 and y.dtype == ydt
    close(x, xr, 2e-6, 1e-5, "x")
    close(mean, mu.reshape(-1), 2e-6, 1e-5, "mean")
    assert rel_l2(rstd, 1 / torch.sqrt(var + 1e-5).reshape(-1)) < 1e-5
    assert rel_l2(y.float(), yr) < (4e-3 if ydt == torch.bfloat16 else 2e-5)
    # and against the two launches it replaces, on the device
    xg = ops.gemm(dev(A), dev(W), bias=dev(b), residual=dev(R), out_dtype=torch.float32)
    yg, mg, rg = ops.layernorm_fwd(xg, dev(gamma), dev(beta), ydt, 1e-5)
    close(x, xg, 1e-6, 1e-5, "x vs gemm")
    assert rel_l2(y.float(), yg.float()) < (4e-3 if ydt == torch.bfloat16 else 2e-5) and rel_l2(rstd, rg) < 1e-5


@pytest.mark.parametrize("M,K,add", [(4096 + 17, 1536, "other"), (208 * 9 + 1, 1152, "alias"), (5000, 384, None), (90, 128, "other"), (208 * 40, 1152, "alias")])
def test_linear_ln_bwd_equals_two_launches(dx, M, K, add, monkeypatch):
    """dinox_linear_ln_bwd (csrc/gemm_bf16_pp384.hip's LayerNorm-backward epilogue: the input-gradient product into width 384 and the
    LayerNorm backward behind it in one launch) against dinox_gemm + dinox_layernorm_bwd: dx equal to the last bit (dy is rounded to
    bf16 in both and the row arithmetic is the same source, compiled twice), d gamma / d beta equal to the summation order of their row
    sums, every launch bit-repeatable; ragged last tiles, dx aliasing dx_add (the in-place LN1 form), no dx_add; and against fp64."""
    ops, _ = dx
    monkeypatch.setenv("DINOX_NT_PP384", "1")     # the stand-alone product on the same K loop (as in the step: M >= 8192, K >= 768)
    N = 384
    g = torch.Generator().manual_seed(M + K)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(DEV)
    Wt = (torch.randn(N, K, generator=g) * (2.0 / math.sqrt(K))).bfloat16().to(DEV)
    x = (torch.randn(M, N, generator=g) * 2 + 3.0 * torch.randn(M, 1, generator=g)).to(DEV)
    gamma = (1 + 0.2 * torch.randn(N, generator=g)).to(DEV)
    mean = x.mean(-1)
    rstd = 1 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)
    base = torch.randn(M, N, generator=g).to(DEV) if add else None

    def run(fused):
        dx_add = None if add is None else base.clone()
        dx_buf = dx_add if add == "alias" else None
        if fused:
            return ops.linear_ln_bwd(A, Wt, x, gamma, mean, rstd, dx=dx_buf, dx_add=dx_add, want_lowp=True)
        dy = ops.gemm(A, Wt)
        return ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx=dx_buf, dx_add=dx_add, want_lowp=True)

    import dinox._lib as L_
    assert L_.lib.dinox_linear_ln_bwd_ok(M, N, K) == 1
    d0, w0, b0, l0 = run(False)
    d1, w1, b1, l1 = run(True)
    d2, w2, b2, l2 = run(True)
    assert torch.equal(d1, d2) and torch.equal(w1, w2) and torch.equal(b1, b2) and torch.equal(l1, l2)
    # the same dy (bf16), the same formulas -- compiled twice: one element in a hundred differs in the last bit
    assert float((d1 - d0).abs().max()) <= 2e-6 * float(d0.abs().max()) and int((d1 != d0).sum()) < d1.numel() // 20
    assert int((l1 != l0).sum()) < l1.numel() // 500
    close(w1, w0.double().cpu(), 1e-5, 1e-4, "d gamma")
    close(b1, b0.double().cpu(), 1e-5, 1e-4, "d beta")
    # fp64 on the bf16-rounded dy
    dy = (A.double().cpu() @ Wt.double().cpu().t()).to(torch.bfloat16).double()
    xh = (x.double().cpu() - mean.double().cpu()[:, None]) * rstd.double().cpu()[:, None]
    gy = dy * gamma.double().cpu()
    ref = rstd.double().cpu()[:, None] * (gy - gy.mean(-1, keepdim=True) - xh * (gy * xh).mean(-1, keepdim=True)) + (base.double().cpu() if add else 0)
    assert rel_l2(d1, ref) < 2e-3                 # (a bf16 rounding of dy on either side of a tie moves single elements)
    assert rel_l2(w1, (dy * xh).sum(0)) < 2e-3 and rel_l2(b1, dy.sum(0)) < 2e-3


def test_gemm_nt_store_policy_is_only_a_hint(dx, monkeypatch):
    """The register-prefetch kernel writes its bf16 outputs with non-temporal stores (DINOX_NT_STORES, csrc/gemm_bf16_areg.hip): a cache
    hint, so results must be bit-identical with it off, for the plain, GELU (+ side tensor) and GELU' forms."""
    ops, _ = dx
    g = torch.Generator(device=DEV).manual_seed(3)
    M, N, K = 128 * 37 + 11, 1536, 384
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.3).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DINOX_NT_STORES", mode)
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        c = ops.gemm(A, B, bias=bias)
        act = ops.gemm(A, B, bias=bias, gelu=True, aux=aux, auxgrad=True)
        d = ops.gemm(A, B, dgelu=True, aux=aux, auxgrad=True)
        outs[mode] = (c, act, aux, d)
    for x, y in zip(outs["0"], outs["1"]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("N,K,res", [(1152, 384, False), (384, 1536, True)])
def test_gemm_nt_areg_full_size_repeatable(dx, N, K, res, monkeypatch):
    """BASELINE size (M = 512 views x 201 tokens; the qkv product and the fc2 product with its fp32 residual) through the
    register-prefetch NT kernel, 120 launches back to back: the kernel has no atomics, so every launch must reproduce the first bit
    for bit (a missed wait on a staged slice shows up as a sporadic difference), and sampled rows must match fp64.  This test is
    what caught LDS reads left in flight across the barrier in front of the accumulator parking (one tile in ~1e5 multiplied by
    parked fp32 words)."""
    ops, _ = dx
    monkeypatch.setenv("DINOX_NT_AREG_MAXK", str(1 << 30))
    monkeypatch.setenv("DINOX_NT_PP", "0")          # (these shapes go to the ping-pong kernels by default since round 3: test_gemm_nt_pp_*)
    g = torch.Generator(device=DEV).manual_seed(0)
    M = 512 * 201
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    r = torch.randn(M, N, device=DEV, generator=g) if res else None
    run = lambda: ops.gemm(A, B, bias=bias, residual=r, out_dtype=torch.float32 if res else None)
    ops.TRACE_KERNELS = []
    try:
        first = run()
        assert ops.TRACE_KERNELS == ["gemm_bf16_nt_areg"], ops.TRACE_KERNELS
    finally:
        ops.TRACE_KERNELS = None
    for _ in range(120):
        assert torch.equal(run(), first)
    assert bool(torch.isfinite(first.float()).all())
    rows = torch.randint(0, M, (256,), device=DEV, generator=g)
    ref = A[rows].double() @ B.double().t() + bias.double() + (r[rows].double() if res else 0)
    assert rel_l2(first[rows].float(), ref) < 3e-3


@pytest.mark.parametrize("K,N", [(1024, 1024), (1600, 384), (4096, 1024)])
def test_gemm_nt_glds_repeatable(dx, K, N, monkeypatch):
    """The LDS-DMA ring NT kernel at long K, 60 launches back to back, bit for bit.  K = 1024 and 4096 (ViT-L widths) end the K loop
    on ring slot 1, which is also where two waves park their accumulators: the last step's LDS reads must have returned before the
    barrier in front of the parking (csrc/gemm_bf16_glds.hip)."""
    ops, _ = dx
    monkeypatch.setenv("DINOX_NT_PP", "0")
    g = torch.Generator(device=DEV).manual_seed(K + N)
    M = 128 * 400
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
    res = torch.randn(M, N, device=DEV, generator=g)
    ops.TRACE_KERNELS = []
    try:
        first = ops.gemm(A, B, residual=res, out_dtype=torch.float32)
        assert ops.TRACE_KERNELS == ["gemm_bf16_nt_glds"], ops.TRACE_KERNELS
    finally:
        ops.TRACE_KERNELS = None
    for _ in range(60):
        assert torch.equal(ops.gemm(A, B, residual=res, out_dtype=torch.float32), first)
    rows = torch.randint(0, M, (128,), device=DEV, generator=g)
    ref = A[rows].double() @ B.double().t() + res[rows].double()
    assert rel_l2(first[rows], ref) < 3e-3


# ------------------------------------------------------------------------------------------ round 3: persistent ping-pong NT kernels
PP_SHAPES = [(256, 256, 192), (512, 512, 384), (1000, 392, 384), (777, 1152, 384), (2048, 1536, 384), (1300, 384, 1536), (256 * 9 + 17, 1536, 192),
             (3000, 1024, 1024), (515, 264, 4096), (5000, 384, 384), (300, 128, 448), (2049, 120, 192)]


@pytest.mark.parametrize("mode", ["1", "2", "3"])
@pytest.mark.parametrize("M,N,K", PP_SHAPES)
def test_gemm_nt_pp(dx, M, N, K, mode, monkeypatch):
    """The persistent ping-pong NT kernels (csrc/gemm_bf16_pp.hip: 256 x 256 tiles, csrc/gemm_bf16_pp128.hip: 256 x 128), forced onto
    small and ragged shapes (DINOX_NT_PP = 1: tile width by shape, 2 / 3: every shape on the 128- / 256-wide form): every epilogue they
    take -- plain, bias, GELU with and without its GELU' side tensor, x GELU', fp32 residual; bf16 and fp32 outputs -- against fp64 on
    the same bf16 operands, with edge tiles in M and N (a wave whose rows / whose column strip lie entirely past the edge), tiles
    of two to 64 K-tiles, a workgroup that owns several tiles (the request stream crosses tile boundaries) and fewer tiles than CUs."""
    ops, _ = dx
    monkeypatch.setenv("DINOX_NT_PP", mode)
    if mode == "2" and K < 192:
        pytest.skip("the 128-wide kernel needs three K-tiles")
    g = torch.Generator().manual_seed(M + N + K)
    A, B = (torch.randn(M, K, generator=g) * 0.5).bfloat16(), (torch.randn(N, K, generator=g) * (6.0 / math.sqrt(K))).bfloat16()
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd = A.to(DEV), B.to(DEV)
    ref = A.double() @ B.double().t()
    ops.TRACE_KERNELS = []
    try:
        c = ops.gemm(Ad, Bd, out_dtype=torch.float32)
        cb = ops.gemm(Ad, Bd, bias=bias.to(DEV))
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        act = ops.gemm(Ad, Bd, bias=bias.to(DEV), gelu=True, aux=aux, auxgrad=True)
        act_t = ops.gemm(Ad, Bd, bias=bias.to(DEV), gelu=True)
        y = ops.gemm(Ad, Bd, bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.float32)
        yb = ops.gemm(Ad, Bd, residual=res.to(DEV))
        d = ops.gemm(Ad, Bd, dgelu=True, aux=aux, auxgrad=True)
        auxf = torch.randn(M, N, generator=g).to(DEV)
        d3 = ops.gemm(Ad, Bd, dgelu=True, aux=auxf, auxgrad=True, out_dtype=torch.float32)
        names = set(ops.TRACE_KERNELS)
        assert len(ops.TRACE_KERNELS) == 8 and names <= {"gemm_bf16_nt_pp", "gemm_bf16_nt_pp128"}, ops.TRACE_KERNELS
        if mode != "1":
            assert names == {"gemm_bf16_nt_pp128" if mode == "2" else "gemm_bf16_nt_pp"}
    finally:
        ops.TRACE_KERNELS = None
    erf = lambda t: torch.erf(t / math.sqrt(2))
    gelu_grad = lambda t: 0.5 * (1 + erf(t)) + t * torch.exp(-0.5 * t * t) / math.sqrt(2 * math.pi)
    close(c, ref, 1e-5, 1e-4, "plain fp32 out")
    assert rel_l2(cb.float(), ref + bias.double()) < 3e-3
    pre = ref + bias.double()
    assert rel_l2(act.float(), 0.5 * pre * (1 + erf(pre))) < 3e-3 and rel_l2(aux.float(), gelu_grad(pre)) < 3e-3
    assert rel_l2(act_t.float(), 0.5 * pre * (1 + erf(pre))) < 3e-3
    close(y, pre + res.double(), 1e-5, 1e-4, "bias + residual, fp32 out")
    assert rel_l2(yb.float(), ref + res.double()) < 3e-3
    assert rel_l2(d.float(), ref * aux.float().double().cpu()) < 3e-3
    close(d3, ref * auxf.double().cpu(), 1e-5, 1e-4, "x GELU' from an fp32 side tensor, fp32 out")
    # and the kernels they replace give the same numbers up to the summation order of the K loop
    monkeypatch.setenv("DINOX_NT_PP", "0")
    close(ops.gemm(Ad, Bd, bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.float32), y, 2e-6, 2e-5, "pp vs the 128 x 128 kernels")


@pytest.mark.parametrize("M,K", [(4096, 128), (4096 + 17, 384), (5000, 1152), (208 * 20 + 207, 1536), (208 * 21 + 1, 160), (208 * 30, 384), (90, 256), (70000, 768)])
def test_gemm_nt_pp384(dx, M, K, monkeypatch):
    """The full-row kernel for N = 384 (csrc/gemm_bf16_pp384.hip: 208 x 384 tiles, one per workgroup, a wave owns all 208 rows of 48
    columns), forced onto small and ragged shapes (DINOX_NT_PP384=1): plain and bias with bf16 and fp32 out, the fp32 residual
    epilogue (three slabs through LDS), against fp64 on the same bf16 operands -- last tiles of 1 .. 207 rows (rows past M read as zeros
    through the buffer descriptor and are never stored), a single partial tile, four to 48 K-tiles (the three-K-tile tail on its own
    at K = 128) -- every launch repeated bit for bit, and equal to the kernels it replaces up to the summation order of the K loop."""
    ops, _ = dx
    monkeypatch.setenv("DINOX_NT_PP384", "1")
    g = torch.Generator().manual_seed(M + K)
    A, B = (torch.randn(M, K, generator=g) * 0.5).bfloat16(), (torch.randn(384, K, generator=g) * (6.0 / math.sqrt(K))).bfloat16()
    bias, res = torch.randn(384, generator=g), torch.randn(M, 384, generator=g)
    Ad, Bd, bd, rd = A.to(DEV), B.to(DEV), bias.to(DEV), res.to(DEV)
    ref = A.double() @ B.double().t()
    runs = {"plain": lambda: ops.gemm(Ad, Bd), "bias": lambda: ops.gemm(Ad, Bd, bias=bd), "f32": lambda: ops.gemm(Ad, Bd, out_dtype=torch.float32),
            "bias_res": lambda: ops.gemm(Ad, Bd, bias=bd, residual=rd, out_dtype=torch.float32), "res": lambda: ops.gemm(Ad, Bd, residual=rd, out_dtype=torch.float32)}
    out = {}
    ops.TRACE_KERNELS = []
    try:
        for k, fn in runs.items():
            out[k] = fn()
            assert torch.equal(fn(), out[k]), k
        assert set(ops.TRACE_KERNELS) == {"gemm_bf16_nt_pp384"} and len(ops.TRACE_KERNELS) == 10, ops.TRACE_KERNELS
    finally:
        ops.TRACE_KERNELS = None
    assert rel_l2(out["plain"].float(), ref) < 3e-3 and rel_l2(out["bias"].float(), ref + bias.double()) < 3e-3
    close(out["f32"], ref, 1e-5, 1e-4, "plain fp32 out")
    close(out["bias_res"], ref + bias.double() + res.double(), 1e-5, 1e-4, "bias + residual, fp32 out")
    close(out["res"], ref + res.double(), 1e-5, 1e-4, "residual, fp32 out")
    monkeypatch.setenv("DINOX_NT_PP384", "0")
    close(ops.gemm(Ad, Bd, bias=bd, residual=rd, out_dtype=torch.float32), out["bias_res"], 2e-6, 2e-5, "pp384 vs the kernels it replaces")


@pytest.mark.parametrize("case", ["qkv", "fc2", "dact", "dx", "dx384"])
def test_gemm_nt_pp_full_size_repeatable(dx, case, monkeypatch):
    """BASELINE size (M = 512 views x 201 tokens) through the ping-pong kernels as the DEFAULT policy dispatches them (qkv and the GELU'
    product on 256 x 256 tiles, fc2 with its fp32 residual and the K = 384 dX product on 256 x 128 tiles, the K = 1152 dX product on the
    full-row 208 x 384 tiles of gemm_bf16_pp384.hip), 60 launches back to back:
    no atomics, so every launch must reproduce the first bit for bit (a missed wait on a landed K-tile, a request overtaking a read
    or a staging tile overwritten too early shows up as a sporadic difference), and sampled rows must match fp64."""
    ops, _ = dx
    monkeypatch.delenv("DINOX_NT_PP", raising=False)
    g = torch.Generator(device=DEV).manual_seed(1)
    M = 512 * 201
    N, K = {"qkv": (1152, 384), "fc2": (384, 1536), "dact": (1536, 384), "dx": (384, 1152), "dx384": (384, 384)}[case]
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=DEV, generator=g) * 0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    r = torch.randn(M, N, device=DEV, generator=g) if case == "fc2" else None
    aux = torch.randn(M, N, device=DEV, generator=g).bfloat16() if case == "dact" else None
    if case == "qkv":
        run = lambda: ops.gemm(A, B, bias=bias)
    elif case == "fc2":
        run = lambda: ops.gemm(A, B, bias=bias, residual=r, out_dtype=torch.float32)
    elif case == "dact":
        run = lambda: ops.gemm(A, B, dgelu=True, aux=aux, auxgrad=True)
    else:
        run = lambda: ops.gemm(A, B)
    ops.TRACE_KERNELS = []
    try:
        first = run()
        want = {"qkv": "gemm_bf16_nt_pp", "dact": "gemm_bf16_nt_pp", "fc2": "gemm_bf16_nt_pp128", "dx": "gemm_bf16_nt_pp384", "dx384": "gemm_bf16_nt_pp128"}[case]
        assert ops.TRACE_KERNELS == [want], ops.TRACE_KERNELS
    finally:
        ops.TRACE_KERNELS = None
    for _ in range(60):
        assert torch.equal(run(), first)
    assert bool(torch.isfinite(first.float()).all())
    rows = torch.cat([torch.randint(0, M, (256,), device=DEV, generator=g), torch.arange(M - 64, M, device=DEV), torch.arange(0, 64, device=DEV)])
    ref = A[rows].double() @ B.double().t()
    if case in ("qkv", "fc2"):
        ref = ref + bias.double()
    if case == "fc2":
        ref = ref + r[rows].double()
    if case == "dact":
        ref = ref * aux[rows].double()
    assert rel_l2(first[rows].float(), ref) < 3e-3


# ------------------------------------------------------------------------------------------ round 2: cross-implementation checkpoints, cache staleness
def _ckpt_tiny_engine(cli, g, hp):
    from dinox.engine import StepHyperParams, TrainEngine
    import zoo.arch as arch
    kw = dict(img_size=28, patch=14, dim=32, depth=2, heads=2, mlp_ratio=4.0, use_grad_checkpoint=False, scale_aware=True)
    student = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim=64).to(DEV)
    teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim=64).to(DEV)
    lr, min_lr, warm, max_steps, wd, ema, ts, tt, cm, gw = [float(v) for v in hp]
    eng = TrainEngine(student, teacher, 64, StepHyperParams(lr=lr, min_lr=min_lr, warmup_steps=int(warm), max_steps=int(max_steps), weight_decay=wd,
                                                            ema=ema, teacher_temp=tt, student_temp=ts, center_momentum=cm, gram_weight=gw))
    return student, teacher, eng


def test_reference_written_checkpoint_resumes_in_engine(dx):
    """SURVEY 8f-1, direction reference -> engine.  tests/golden/ref_checkpoint_00000003.pth was written by the REFERENCE's
    save_checkpoint (scripts/phase5_big_run.py:1104-1125) after three steps of its loop; the CLI's load_checkpoint must restore
    student, teacher, AdamW moments + step count, DINO centre and the micro-batch counter from it, and the engine's step 4 must
    equal the step 4 the reference takes after its own load_checkpoint (ckpt_tiny.npz), at the north-star 1e-3."""
    import os
    from conftest import GOLDEN
    cli = _cli()
    g = load_golden("ckpt_tiny.npz")
    student, teacher, eng = _ckpt_tiny_engine(cli, g, g["hp"])
    step, cfg = cli.load_checkpoint(os.path.join(GOLDEN, "ref_checkpoint_00000003.pth"), student, teacher, eng, torch.device(DEV), scale_aware=True)
    assert step == 3 and eng.step_count == 3 and eng.opt_steps == 3
    assert cfg.model.dim == 32 and cfg.scale_aware is True and cfg.lr == pytest.approx(1e-3)
    for k, v in sub(g, "student3").items():
        close(student.state_dict()[k], v, 0, 0, f"restored student {k}")
    for k, v in sub(g, "teacher3").items():
        close(teacher.state_dict()[k], v, 0, 0, f"restored teacher {k}")
    close(eng.center, g["center3"], 0, 0, "restored centre")
    eng.step(t(g["batch3"]).to(DEV), t(g["spacing3"]).to(DEV))
    got = eng.scalars()
    assert got["loss"] == pytest.approx(float(g["losses"][3]), rel=1e-3)
    assert got["grad_norm"] == pytest.approx(float(g["grad_norms"][3]), rel=1e-3)
    assert got["lr"] == pytest.approx(float(g["lrs"][3]), rel=1e-12)
    assert eng.opt_steps == int(g["adam_step4"])
    def close_params(model, want, what):
        for k, v in want.items():
            got_k = model.state_dict()[k]
            if k.endswith("attn.qkv.bias"):      # key part: numerically-zero gradient, Adam moves it by up to lr with a round-off sign
                D = v.numel() // 3               # (same rule as the payload test below) -> one step x lr
                close(got_k[D:2 * D], v[D:2 * D], 0, 1.1e-3, f"{what}: {k} (key part)")
                got_k, v = torch.cat([got_k[:D], got_k[2 * D:]]), torch.cat([v[:D], v[2 * D:]])
            close(got_k, v, 1e-3, 2e-5, f"{what}: {k}")
    close_params(student, sub(g, "student4"), "student after step 4")
    close_params(teacher, sub(g, "teacher4"), "teacher after step 4")
    close(eng.center, g["center4"], 1e-4, 1e-7, "centre after step 4")


def test_engine_written_checkpoint_payload(dx, tmp_path):
    """Direction engine -> reference: three engine steps from ckpt_tiny's initial state, the CLI's save_checkpoint, then step 4.
    Here the payload is checked to be what the reference's load_checkpoint consumes (keys, AdamW state_dict format accepted by a
    stock torch.optim.AdamW, centre under dino_loss.center) and to equal the reference's state after the same three steps; the
    file + the engine's step 4 are left in gpurun_out/engine_ckpt/ so that tests/test_checkpoint_xref.py (build container, real
    reference) can resume it with the reference's own load_checkpoint and compare step 4."""
    import json, os
    from conftest import ROOT
    cli = _cli()
    g = load_golden("ckpt_tiny.npz")
    student, teacher, eng = _ckpt_tiny_engine(cli, g, g["hp"])
    student.load_state_dict({k: v.to(DEV) for k, v in sub(g, "init").items()})
    teacher.load_state_dict(student.state_dict())
    losses = []
    for i in range(3):
        eng.step(t(g[f"batch{i}"]).to(DEV), t(g[f"spacing{i}"]).to(DEV))
        losses.append(eng.scalars()["loss"])
    assert losses == pytest.approx([float(v) for v in g["losses"][:3]], rel=1e-3)
    mc = cli.ModelConfig(name="custom", patch=14, dim=32, depth=2, heads=2, mlp_ratio=4.0, out_dim=64)
    tc = cli.TrainingConfig(model=mc, img_size=28, batch_size=3, lr=1e-3, min_lr=1e-5, warmup_steps=2, weight_decay=0.04, max_steps=10, ema=0.9,
                            center_momentum=0.9, scale_aware=True)
    out_dir = os.path.join(ROOT, "gpurun_out", "engine_ckpt")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "engine_checkpoint_00000003.pth")
    cli.save_checkpoint(path, 3, student, teacher, eng, tc)
    payload = torch.load(path, map_location="cpu", weights_only=False)         # our own file, written one line above
    assert set(payload) == {"step", "student", "teacher", "opt", "scaler", "dino_loss", "rng", "config"}       # reference :1115-1124
    assert payload["scaler"] is None and set(payload["dino_loss"]) == {"center"} and payload["config"]["model"]["dim"] == 32
    import torch.nn as nn
    probe = [nn.Parameter(torch.zeros_like(p, device="cpu")) for p in eng.params]
    stock = torch.optim.AdamW(probe, lr=1e-3, weight_decay=0.04)
    stock.load_state_dict(payload["opt"])                                      # what the reference does at :1171
    assert float(stock.state_dict()["state"][0]["step"]) == 3.0
    for k, v in sub(g, "student3").items():
        got_k = payload["student"][k]
        if k.endswith("attn.qkv.bias"):          # the key bias has a numerically-zero gradient (softmax is invariant to it): Adam turns
            D = v.numel() // 3                   # round-off into +-lr moves whose sign is noise -> bounded by 3 steps x lr
            close(got_k[D:2 * D], v[D:2 * D], 0, 3.1e-3, f"payload student {k} (key part)")
            got_k, v = torch.cat([got_k[:D], got_k[2 * D:]]), torch.cat([v[:D], v[2 * D:]])
        close(got_k, v, 1e-3, 2e-5, f"payload student {k}")
    close(payload["dino_loss"]["center"], g["center3"], 1e-4, 1e-7, "payload centre")
    eng.step(t(g["batch3"]).to(DEV), t(g["spacing3"]).to(DEV))
    got = eng.scalars()
    torch.save({k: v.detach().cpu() for k, v in student.state_dict().items()}, os.path.join(out_dir, "engine_student_after_step4.pth"))
    json.dump({"loss4": got["loss"], "grad_norm4": got["grad_norm"], "lr4": got["lr"], "losses": losses},
              open(os.path.join(out_dir, "engine_step4.json"), "w"))
    assert got["loss"] == pytest.approx(float(g["losses"][3]), rel=1e-3)


def test_encode_twice_and_two_models_do_not_share_cached_tensors(dx):
    """encode(model, A) then encode(model, B) -- the second input lands on the freed first one's address -- must
    give B's features, not A's; and loading a second checkpoint onto the first model's addresses must not be served the first
    model's cached bf16 weights."""
    ops, arch = dx
    import gc
    import zoo.encode as enc
    kw = dict(img_size=32, patch=16, dim=64, depth=2, heads=2, num_registers=4, scale_aware=True)
    rng = np.random.default_rng(1)
    imgs = [rng.normal(40, 300, size=(32, 32)).astype(np.float32) for _ in range(3)]
    with ops.compute_dtype(torch.bfloat16):
        torch.manual_seed(1)
        m = arch.PatchViT(**kw).to(DEV).eval()
        feats = [enc.encode(m, im, pixel_spacing=(0.7, 0.7), slice_thickness=2.0).clone() for im in imgs]      # same shapes: addresses recycle
        assert rel_l2(feats[1], feats[0]) > 1e-3 and rel_l2(feats[2], feats[1]) > 1e-3
        again = enc.encode(m, imgs[1], pixel_spacing=(0.7, 0.7), slice_thickness=2.0)
        close(again, feats[1], 0, 0, "encode is a function of its input")
        sd1 = {k: v.clone() for k, v in m.state_dict().items()}
        x = torch.randn(2, 3, 32, 32, device=DEV)
        sp = torch.tensor([[0.7, 0.7, 2.0]] * 2, device=DEV)
        with torch.no_grad():
            y1 = m(x, spacing=sp).clone()
        del m
        gc.collect()
        torch.manual_seed(2)
        m2 = arch.PatchViT(**kw).to(DEV).eval()          # lands on the first model's freed blocks, every parameter at version 0
        with torch.no_grad():
            y2 = m2(x, spacing=sp).clone()
        assert rel_l2(y2, y1) > 1e-2, "second model answered with the first model's cached weights"
        m2.load_state_dict(sd1)
        with torch.no_grad():
            close(m2(x, spacing=sp), y1, 0, 0, "same weights, same answer")


def test_configs0_fifty_step_trajectory_matches_oracle(dx):
    """BASELINE configs[0] as a TRAJECTORY: the plumbing loop of scripts/baseline_cifar10_pretrain.py:347-413 (ViT-Tiny dim 192 / depth
    12 / heads 3 / out 4096, 32 px, patch 4 -> 69 tokens, bs 32 = 64 views, fresh synthetic N(0,1) batch every step, Gram off, no
    scale embedding, lr 2e-4 / warm-up 500 / wd 0.04 / ema 0.996 and DINOLoss's default centre momentum 0.999) for its 50 steps:
    engine (fp32 parity mode) against the CPU oracle stepping the same batches.  Per-step loss within 1e-3 relative for the first
    ten steps and within the reference canary's 0.5 % (integration_canary.py:161-178) all the way to step 50; the weights after 50
    optimiser steps (Adam moments, EMA teacher and centre carried the whole way) within 1e-3."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    from oracle import dinox_oracle as O
    kw = dict(img_size=32, patch=4, dim=192, depth=12, heads=3, num_registers=4, scale_aware=False)
    out_dim, B, steps = 4096, 32, 50
    cfg = O.VitCfg(out_dim=out_dim, **kw)
    st = O.init_state(cfg, O.random_params(cfg, seed=41))
    hyp = dict(lr=2e-4, min_lr=1e-6, warmup_steps=500, max_steps=20000, weight_decay=0.04, ema=0.996, center_momentum=0.999, gram_weight=0.0)
    student = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim)
    teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), out_dim)
    student.load_state_dict(st.student)
    teacher.load_state_dict(st.teacher)
    eng = TrainEngine(student.to(DEV), teacher.to(DEV), out_dim, StepHyperParams(**hyp))
    g = torch.Generator().manual_seed(42)
    hp_o = O.HyperParams(**hyp)
    want, got = [], []
    for i in range(steps):
        batch = torch.randn(2 * B, 3, 32, 32, generator=g)
        r = O.train_step(st, batch, None, hp_o)
        want.append((r["loss"], r["grad_norm"]))
        eng.step(batch.to(DEV), None)
        sc = eng.scalars()
        got.append((sc["loss"], sc["grad_norm"]))
    for i, ((lw, gw), (lg, gg)) in enumerate(zip(want, got)):
        tol = 1e-3 if i < 10 else 5e-3
        assert lg == pytest.approx(lw, rel=tol), (i, lg, lw)
        assert gg == pytest.approx(gw, rel=10 * tol), (i, gg, gw)
    assert want[-1][0] < want[0][0]                           # (it trains: the loss moved)
    close(eng.center, st.center, 1e-4, 1e-7, "centre after 50 steps")
    ssd, tsd = student.state_dict(), teacher.state_dict()
    n_bad = n_all = 0
    for k, v in st.student.items():
        d = (ssd[k].cpu().double() - v.double()).abs()
        n_bad += int((d > 1e-3 * v.double().abs() + 2e-5).sum())
        n_all += d.numel()
    assert n_bad <= 2e-3 * n_all, (n_bad, n_all)              # (Adam's sign noise on numerically-zero gradients)
    for k, v in st.teacher.items():
        close(tsd[k], v, 1e-3, 2e-5, f"teacher {k} after 50 EMA updates")


def test_graph_replay_matches_eager(dx):
    """TrainEngine(use_graph=True): two eager steps, then the whole optimiser step is captured into ONE hipGraph and replayed with a
    new batch, a new learning rate and new Adam bias corrections every step (lr / corrections travel through device memory,
    dinox_adamw_ema_dev).  Six steps must equal six eager steps (fp32 mode; the split-K atomics of the dW products make both
    runs agree to round-off, not bit-wise), and bf16 mode must replay too."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    kw = dict(img_size=32, patch=16, dim=64, depth=2, heads=2, num_registers=4, scale_aware=True)
    torch.manual_seed(5)
    ref = arch.DinoStudentTeacher(arch.PatchViT(**kw), 256)
    torch.nn.init.xavier_uniform_(ref.backbone.scale_embed.mlp[2].weight)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    batches = [(torch.randn(8, 3, 32, 32, generator=g).to(DEV), (torch.rand(4, 3, generator=g) + 0.5).repeat(2, 1).to(DEV)) for _ in range(6)]

    def run(graph, amp):
        student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 256)
        teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 256)
        student.load_state_dict(sd)
        teacher.load_state_dict(sd)
        eng = TrainEngine(student.to(DEV), teacher.to(DEV), 256, StepHyperParams(lr=1e-3, warmup_steps=4, max_steps=8, ema=0.9, koleo_weight=0.1),
                          amp_dtype=amp, use_graph=graph)
        out = []
        for b, s in batches:
            eng.step(b, s)
            sc = eng.scalars()
            out.append((sc["loss"], sc["grad_norm"], sc["lr"]))
        assert (eng._graph is not None) == graph and eng.step_count == 6 and eng.opt_steps == 6
        return out, eng.flat_p.clone(), eng.flat_t.clone(), eng.center.clone()

    for amp, tol in ((None, 1e-4), (torch.bfloat16, 2e-2)):
        e, pe, te, ce = run(False, amp)
        gr, pg, tg, cg = run(True, amp)
        for (le, ge, lre), (lg, gg, lrg) in zip(e, gr):
            assert lg == pytest.approx(le, rel=tol) and gg == pytest.approx(ge, rel=10 * tol) and lrg == lre
        assert len({round(l, 6) for l, _, _ in gr}) == 6                 # every replay saw its own batch
        close(cg, ce, tol, 1e-6, "centre")
        close(tg, te, 10 * tol, 1e-4, "teacher arena")
        assert float(((pg - pe).abs() <= 10 * tol * pe.abs() + 2.1e-3).double().mean()) == 1.0


@pytest.mark.parametrize("K,M,N,cs", [(102912, 1536, 384, True), (25728, 384, 384, True), (5000, 1152, 384, False), (804 * 32 + 17, 136, 264, True),
                                      (102912, 384, 1536, True), (30000, 1152, 384, True), (100352, 384, 768, False), (9000, 64, 72, True),
                                      (12864, 4096, 1024, True), (12864, 1024, 1024, True), (9000, 3072, 1024, True)])      # ViT-L: the 256 x 256 / 384 x 128 forms
def test_gemm_tn_split_k_is_bit_reproducible(dx, K, M, N, cs):
    """The dW products (C[M,N] = A[K,M]^T B[K,N], K = every token of the batch) split K over the chip.  With the workspace the host
    side hands them (ops._tn_workspace) the splits meet in a fixed-order two-stage reduction instead of fp32 atomics: five launches
    give bit-identical results -- also when accumulating into a gradient that is already there, and for the bias gradient riding
    along -- and equal fp64 on the same bf16 operands."""
    ops, _ = dx
    g = torch.Generator().manual_seed(K + M)
    A = (torch.randn(K, M, generator=g) * 0.5).bfloat16().to(DEV)
    B = (torch.randn(K, N, generator=g) * 0.5).bfloat16().to(DEV)
    C0 = torch.randn(M, N, generator=g).to(DEV)
    c0 = torch.randn(M, generator=g).to(DEV)
    outs = []
    ops.TRACE_KERNELS = []
    try:
        for _ in range(5):
            C, cvec = C0.clone(), c0.clone()
            ops.gemm(A, B, transA=True, transB=True, out=C, accumulate=True, colsum_out=cvec if cs else None)
            P = ops.gemm(A, B, transA=True, transB=True, out_dtype=torch.float32)
            outs.append((C, cvec, P))
        assert set(ops.TRACE_KERNELS) == {"gemm_bf16_tn_big" if K >= 8192 else "gemm_bf16_tn_dma"}       # long K: the big-tile form
    finally:
        ops.TRACE_KERNELS = None
    for C, cvec, P in outs[1:]:
        assert torch.equal(C, outs[0][0]) and torch.equal(cvec, outs[0][1]) and torch.equal(P, outs[0][2])
    ref = A.double().t() @ B.double()
    assert rel_l2(outs[0][2], ref) < 2e-6
    assert rel_l2(outs[0][0] - C0, ref) < 1e-5
    if cs:
        assert rel_l2(outs[0][1] - c0, A.double().sum(0)) < 1e-5


@pytest.mark.parametrize("B,N,heads,D,bias", [(3, 201, 6, 384, True), (70, 201, 6, 384, True), (2, 224, 2, 96, False), (5, 197, 16, 1024, True),
                                                 (1, 201, 6, 384, True)])
def test_qkv_attention_fused_vs_composed_and_fp64(dx, B, N, heads, D, bias):
    """dinox_qkv_attention_fwd (projection + attention in one launch: the packed qkv never reaches HBM) against (a) fp64 on the same
    bf16 operands -- qkv rounded to bf16 as the kernel hands it over -- and (b) the two launches it replaces (dinox_gemm +
    dinox_attention_fwd): the same arithmetic in another summation order, so equal to a bf16 rounding step.  Padded tokens (N < 224),
    both pair->workgroup mappings (B < 64: pair by pair; B >= 64: one image's heads on one XCD), optional qkv / lse outputs."""
    ops, _ = dx
    C = heads * 64
    g = torch.Generator().manual_seed(B * 1000 + N + D)
    x = (torch.randn(B, N, D, generator=g) * 0.7).bfloat16().to(DEV)
    w = (torch.randn(3 * C, D, generator=g) * (1.5 / D ** 0.5)).bfloat16().to(DEV)
    b = (torch.randn(3 * C, generator=g) * 0.2).to(DEV) if bias else None
    assert ops.qkv_attention_ok(B, N, heads, D, C) and not ops.qkv_attention_ok(B, 261, heads, D, C) and not ops.qkv_attention_ok(B, N, heads, D, heads * 88)
    o, qkv, lse = ops.qkv_attention(x, w, b, heads, want_qkv=True, want_lse=True)
    o2, none_qkv, none_lse = ops.qkv_attention(x, w, b, heads)
    assert none_qkv is None and none_lse is None and torch.equal(o, o2)                      # the optional outputs change nothing
    torch.cuda.synchronize()
    # (a) fp64
    qkv64 = x.double().reshape(B * N, D) @ w.double().t() + (b.double() if bias else 0.0)
    assert float((qkv.double().reshape(B * N, 3 * C) - qkv64).abs().max()) <= 2.0 ** -8 * float(qkv64.abs().max()) + 1e-6
    qr = qkv.double().reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)                    # what the attention saw: the bf16 hand-over
    s = qr[0] @ qr[1].transpose(-1, -2) / 8.0
    ref = (torch.softmax(s, -1) @ qr[2]).permute(0, 2, 1, 3).reshape(B, N, C)
    assert rel_l2(o.double(), ref) < 4e-3 and float((o.double() - ref).abs().max()) < 2.5e-2 * float(ref.abs().max())
    assert float((lse.double() - torch.logsumexp(s, -1)).abs().max()) < 2e-3
    # (b) the composed launches
    qkv_c = ops.gemm(x.reshape(B * N, D), w, bias=b, out_dtype=torch.bfloat16).reshape(B, N, 3 * C)
    o_c, lse_c = ops.attention_fwd(qkv_c, heads)
    assert float((qkv.float() - qkv_c.float()).abs().max()) <= 2.0 ** -7 * float(qkv_c.float().abs().max())
    assert rel_l2(o.float(), o_c.float()) < 4e-3 and float((lse - lse_c).abs().max()) < 2e-2


@pytest.mark.parametrize("native", [True, False])
def test_no_grad_forward_with_fused_qkv_attention(dx, native, monkeypatch):
    """DINOX_QKV_FUSED=1: the no-grad forward (the teacher of a step, encode()) runs qkv projection + attention as one launch per block,
    through dinox_block_forward (qkv == NULL) and through the Python-sequenced block alike; features equal the default path's to bf16
    rounding, and a training forward (grad enabled) never takes it."""
    ops, arch = dx
    torch.manual_seed(5)
    net = arch.PatchViT(img_size=224, patch=16, dim=128, depth=2, heads=2, scale_aware=False).to(DEV)
    x = torch.randn(3, 3, 224, 224, device=DEV)
    monkeypatch.setattr(ops, "_BLOCK_NATIVE", native)

    def fwd(fused, grad):
        """-> (features, dinox_gemm launches with N = 3 D: the qkv products that ran as launches of their own)"""
        monkeypatch.setattr(ops, "_QKV_FUSED", fused)
        t = ops.GemmTimer(every=1 << 20)
        with t, ops.compute_dtype(torch.bfloat16), (torch.enable_grad() if grad else torch.no_grad()):
            y = net(x)
        n_qkv = sum(int(f[10]) for f in (l.split() for l in t.text.splitlines()) if len(f) == 13 and int(f[2]) == 3 * 128 and int(f[3]) == 128)
        return y, n_qkv

    y0, q0 = fwd(False, False)
    y1, q1 = fwd(True, False)
    assert (q0, q1) == (2, 0)                                                         # both blocks: no qkv launch, no qkv tensor
    assert rel_l2(y1.float(), y0.float()) < 6e-3                                      # (the same sums in another order; often bit-equal at K = 128)
    y2, q2 = fwd(True, True)                                                          # a forward that will be differentiated keeps qkv
    assert q2 == 2
    y2.float().square().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.requires_grad)


@pytest.mark.parametrize("form", ["1", "2", "3"])
def test_gemm_tn_big_every_tile_form(dx, form, monkeypatch):
    """gemm_bf16_tn_big has three tile shapes (256 x 192, 384 x 128, 256 x 256; the plan picks by kps x TM x TN): each one forced
    (DINOX_TN_FORM) on a ragged product -- partial tiles in both directions, a K that is no multiple of the 32-row step -- against fp64,
    with the bias gradient, twice for bit-reproducibility."""
    ops, _ = dx
    monkeypatch.setenv("DINOX_TN_FORM", form)
    K, M, N = 9000 + 24, 1096, 520
    g = torch.Generator().manual_seed(int(form))
    A = (torch.randn(K, M, generator=g) * 0.5).bfloat16().to(DEV)
    B = (torch.randn(K, N, generator=g) * 0.5).bfloat16().to(DEV)
    outs = []
    ops.TRACE_KERNELS = []
    try:
        for _ in range(2):
            cvec = torch.zeros(M, device=DEV)
            outs.append((ops.gemm(A, B, transA=True, transB=True, out_dtype=torch.float32, colsum_out=cvec), cvec))
        assert set(ops.TRACE_KERNELS) == {"gemm_bf16_tn_big"}
    finally:
        ops.TRACE_KERNELS = None
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel_l2(outs[0][0], A.double().t() @ B.double()) < 2e-6 and rel_l2(outs[0][1], A.double().sum(0)) < 1e-5


@pytest.mark.parametrize("K,M,N", [(9000 + 24, 1096, 520), (8192, 384, 384), (8192 + 40, 1536, 384), (33000, 384, 1152), (12864, 1024, 4096)])
def test_gemm_tn_big_anti_phase_loop_equals_in_step_loop(dx, K, M, N, monkeypatch):
    """gemm_bf16_tn_big's K loop runs two wave groups half a step apart (one reads a step's fragments while the other issues its MFMAs);
    DINOX_TN_PP=0 keeps every wave in step, =1 is the anti-phase loop on the same MFMA shape: the same products added in the same order,
    bit-equal results, bias gradient included.  The default (=2, v_mfma_f32_16x16x32_bf16: 32 products per instruction instead of 16)
    differs from them by fp32 rounding only and is bit-reproducible.  Ragged tiles, a K that leaves the last split a handful of steps
    (or less than one), every tile shape the plan picks."""
    ops, _ = dx
    g = torch.Generator().manual_seed(K % 997)
    A = (torch.randn(K, M, generator=g) * 0.5).bfloat16().to(DEV)
    B = (torch.randn(K, N, generator=g) * 0.5).bfloat16().to(DEV)
    outs = {}
    for pp in ("0", "1", "2", "2 again", None):
        if pp is None:
            monkeypatch.delenv("DINOX_TN_PP")
        else:
            monkeypatch.setenv("DINOX_TN_PP", pp[0])
        ops.TRACE_KERNELS = []
        try:
            cvec = torch.zeros(M, device=DEV)
            outs[pp] = (ops.gemm(A, B, transA=True, transB=True, out_dtype=torch.float32, colsum_out=cvec), cvec)
            assert set(ops.TRACE_KERNELS) == {"gemm_bf16_tn_big"}
        finally:
            ops.TRACE_KERNELS = None
    assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])
    assert torch.equal(outs["2"][0], outs["2 again"][0]) and torch.equal(outs["2"][1], outs["2 again"][1])
    assert torch.equal(outs["2"][0], outs[None][0]) and torch.equal(outs["2"][1], outs[None][1])          # (the default)
    want, wantb = A.double().t() @ B.double(), A.double().sum(0)
    for pp in ("1", "2"):
        assert rel_l2(outs[pp][0], want) < 2e-6 and rel_l2(outs[pp][1], wantb) < 1e-5


def test_training_steps_are_bit_reproducible(dx):
    """Two runs of three bf16 optimiser steps from the same state end in bit-identical student, teacher, Adam moments and centre:
    no kernel of the step leaves the order of a floating-point sum to the scheduler (round 1: the split-K atomics of the dW products did);
    the same again with DINOX_SIDE_STREAM / DINOX_DW_STREAM (concurrent launch chains)."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    kw = dict(img_size=64, patch=16, dim=384, depth=2, heads=6, num_registers=4, scale_aware=True)
    torch.manual_seed(3)
    ref = arch.DinoStudentTeacher(arch.PatchViT(**kw), 1024)
    torch.nn.init.xavier_uniform_(ref.backbone.scale_embed.mlp[2].weight)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    batches = [(torch.randn(64, 3, 64, 64, generator=g).to(DEV), (torch.rand(32, 3, generator=g) + 0.5).repeat(2, 1).to(DEV)) for _ in range(3)]

    def run():
        student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 1024)
        teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 1024)
        student.load_state_dict(sd)
        teacher.load_state_dict(sd)
        eng = TrainEngine(student.to(DEV), teacher.to(DEV), 1024, StepHyperParams(lr=1e-3, warmup_steps=2, max_steps=8, ema=0.9, koleo_weight=0.1),
                          amp_dtype=torch.bfloat16)
        for b, s in batches:
            eng.step(b, s)
        return [t.clone() for t in (eng.flat_p, eng.flat_t, eng.adam_m, eng.adam_v, eng.center, eng.flat_g)], eng.scalars()["loss"]

    a, la = run()
    b, lb = run()
    assert la == lb
    for name, x, y in zip(("student", "teacher", "adam_m", "adam_v", "centre", "last gradient"), a, b):
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} of {x.numel()} elements differ between two identical runs"
    # ... and with the opt-in second streams (the teacher's forward beside the student's, the weight-gradient products beside backward):
    # another order of launches in time, the same sums
    import os
    was = ops.dw_stream.enabled
    os.environ["DINOX_SIDE_STREAM"] = "1"
    ops.dw_stream.enabled = True
    try:
        c, lc = run()
    finally:
        os.environ.pop("DINOX_SIDE_STREAM", None)
        ops.dw_stream.enabled = was
    assert lc == la
    for name, x, y in zip(("student", "teacher", "adam_m", "adam_v", "centre", "last gradient"), a, c):
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} of {x.numel()} elements differ with the side streams on"


@pytest.mark.parametrize("amp", [False, True])
def test_dw_stream_changes_no_bit(dx, amp):
    """DINOX_DW_STREAM=1 (ops.dw_stream: weight-gradient products enqueued on a second HIP stream, joined before the optimiser / before a
    bucket is exchanged) must change the ORDER of launches only: three optimiser steps with it end bit-identical to three without, in the
    fp32 parity mode (where the proj product reads the very buffer LayerNorm's backward then overwrites in place) and in the bf16 mode."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    kw = dict(img_size=64, patch=16, dim=384, depth=3, heads=6, num_registers=4, scale_aware=True)
    torch.manual_seed(5)
    ref = arch.DinoStudentTeacher(arch.PatchViT(**kw), 1024)
    torch.nn.init.xavier_uniform_(ref.backbone.scale_embed.mlp[2].weight)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    batches = [(torch.randn(96, 3, 64, 64, generator=g).to(DEV), (torch.rand(48, 3, generator=g) + 0.5).repeat(2, 1).to(DEV)) for _ in range(3)]

    def run(side: bool):
        was = ops.dw_stream.enabled
        ops.dw_stream.enabled = side
        try:
            student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 1024)
            teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 1024)
            student.load_state_dict(sd)
            teacher.load_state_dict(sd)
            eng = TrainEngine(student.to(DEV), teacher.to(DEV), 1024, StepHyperParams(lr=1e-3, warmup_steps=2, max_steps=8, ema=0.9, koleo_weight=0.1),
                              amp_dtype=torch.bfloat16 if amp else None, accumulation_steps=1)
            for b, s in batches:
                eng.step(b, s)
            torch.cuda.synchronize()
            return [t.clone() for t in (eng.flat_p, eng.flat_t, eng.adam_m, eng.adam_v, eng.center, eng.flat_g)]
        finally:
            ops.dw_stream.enabled = was

    a, b = run(False), run(True)
    assert ops.dw_stream.streams, "the side stream was never used"
    for name, x, y in zip(("student", "teacher", "adam_m", "adam_v", "centre", "last gradient"), a, b):
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} of {x.numel()} elements differ with the dW stream"


def test_empty_batch_is_a_no_op_like_the_reference(dx):
    """A batch of zero images (the reference's torch modules take it: conv2d / LayerNorm / SDPA of nothing) returns (0, 1 + P + R, D) in
    fp32 without a launch -- the C ABI takes no null operands -- and a backward pass through it leaves ZERO gradients on every parameter,
    as autograd does for the reference; a CPU tensor still raises (no CPU fallback)."""
    ops, arch = dx
    m = arch.PatchViT(img_size=64, patch=16, dim=384, depth=2, heads=6, num_registers=4, scale_aware=True).to(DEV)
    for ctx in (contextlib.nullcontext(), torch.autocast("cuda", dtype=torch.bfloat16)):
        with ctx:
            y = m(torch.empty(0, 3, 64, 64, device=DEV), spacing=torch.empty(0, 3, device=DEV))
        assert y.shape == (0, 1 + 16 + 4, 384) and y.dtype == torch.float32 and y.requires_grad
        m.zero_grad(set_to_none=True)
        y.sum().backward()
        for n, q in m.named_parameters():
            assert q.grad is not None and not q.grad.any(), n
    with pytest.raises(RuntimeError, match="CPU"):
        m(torch.empty(0, 3, 64, 64))


def test_fp32_mode_fast_paths_equal_the_reference_kernels(dx):
    """The fp32 parity mode at full size runs (a) attention as batched exact-fp32 products around softmax row kernels instead of the
    per-lane reference kernels, (b) dW products with the token axis cut into chunks (one batched launch + the fixed-order column sum),
    (c) Linear products on 128 x 128 tiles.  Each must agree with the small-size form it replaces: (a) to fp32 round-off of a different
    summation order, (b) likewise, (c) bit for bit (same k-ordered chain per output)."""
    ops, _ = dx
    g = torch.Generator(device="cuda").manual_seed(11)
    B, N, H = 24, 201, 6                                            # 24 x 201 = 4824 rows: above the size where the product form takes over
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g) * 0.5
    do = torch.randn(B, N, H * 64, device="cuda", generator=g) * 0.1
    assert ops._use_f32_products(qkv, N, 64)
    o1, l1 = ops.attention_fwd(qkv, H)
    d1 = ops.attention_bwd(do, qkv, o1, l1, H)
    ops._ATTN_F32_REF = True
    try:
        assert not ops._use_f32_products(qkv, N, 64)
        o0, l0 = ops.attention_fwd(qkv, H)
        d0 = ops.attention_bwd(do, qkv, o0, l0, H)
    finally:
        ops._ATTN_F32_REF = False
    close(o1, o0, 1e-5, 1e-6, "fp32 attention o")
    close(l1, l0, 1e-6, 1e-6, "fp32 attention lse")
    close(d1, d0, 2e-5, 1e-6, "fp32 attention dqkv")
    # (a') bf16 mode, shapes outside the whole-strip MFMA kernels: since round 3 the tiled MFMA kernels (csrc/attention_flash.hip) take
    # every head size that is a multiple of 8 up to 128 and any length; they must agree with the per-lane reference kernels
    # (DINOX_ATTN_NO_FLASH=1 at the C ABI) to bf16 round-off -- ViT-g's head size 88, and head size 64 at 600 tokens
    import os
    for (Bg, Ng, Hg, dg) in ((48, 100, 4, 88), (8, 600, 2, 64)):
        qb = (torch.randn(Bg, Ng, 3 * Hg * dg, device="cuda", generator=g) * 0.5).bfloat16()
        dob = (torch.randn(Bg, Ng, Hg * dg, device="cuda", generator=g) * 0.1).bfloat16()
        assert not ops._bf16_attention_needs_products(qb, Ng, dg, fwd=True) and not ops._bf16_attention_needs_products(qb, Ng, dg, fwd=False)
        ob, lb = ops.attention_fwd(qb, Hg)
        db_ = ops.attention_bwd(dob, qb, ob, lb, Hg)
        os.environ["DINOX_ATTN_NO_FLASH"] = "1"
        try:
            orf, lrf = ops.attention_fwd(qb, Hg)
            drf = ops.attention_bwd(dob, qb, orf, lrf, Hg)
        finally:
            del os.environ["DINOX_ATTN_NO_FLASH"]
        assert ob.dtype == torch.bfloat16 and db_.dtype == torch.bfloat16
        assert rel_l2(ob.float(), orf.float()) < 6e-3 and rel_l2(db_.float(), drf.float()) < 1e-2
        close(lb, lrf, 1e-4, 1e-4, f"lse (head size {dg}, {Ng} tokens)")
    # (a'') bf16 mode, a head size that is NOT a multiple of 8 (no preset has one): the exact-fp32 product form on a float copy, in batch
    # chunks that keep the score buffer within ops._PRODUCTS_BUDGET (shrunk here so that the chunk loop really runs)
    Bg, Ng, Hg, dg = 48, 100, 4, 84
    qb = (torch.randn(Bg, Ng, 3 * Hg * dg, device="cuda", generator=g) * 0.5).bfloat16()
    dob = (torch.randn(Bg, Ng, Hg * dg, device="cuda", generator=g) * 0.1).bfloat16()
    assert ops._bf16_attention_needs_products(qb, Ng, dg, fwd=True)
    budget = ops._PRODUCTS_BUDGET
    ops._PRODUCTS_BUDGET = 10 * 4 * Ng * Ng                         # 10 images per chunk: 48 = 4 x 10 + 8
    try:
        assert len(ops._product_chunks(Bg, Ng)) == 5
        ob, lb = ops.attention_fwd(qb, Hg)
        db_ = ops.attention_bwd(dob, qb, ob, lb, Hg)
    finally:
        ops._PRODUCTS_BUDGET = budget
    ops._ATTN_F32_REF = True
    try:
        orf, lrf = ops.attention_fwd(qb, Hg)
        drf = ops.attention_bwd(dob, qb, orf, lrf, Hg)
    finally:
        ops._ATTN_F32_REF = False
    assert rel_l2(ob.float(), orf.float()) < 6e-3 and rel_l2(db_.float(), drf.float()) < 1e-2
    close(lb, lrf, 1e-4, 1e-4, "lse (head size 84, product form in chunks)")
    # (b) dW = dy^T x with K = 16384 tokens, accumulated into an existing gradient, with the bias gradient
    K, M, Nn = 16384, 384, 256
    dy, x = torch.randn(K, M, device="cuda", generator=g), torch.randn(K, Nn, device="cuda", generator=g)
    base, bb = torch.randn(M, Nn, device="cuda", generator=g), torch.randn(M, device="cuda", generator=g)
    got, gb = base.clone(), bb.clone()
    ops.TRACE_KERNELS = []
    try:
        ops.gemm(dy, x, transA=True, transB=True, out=got, accumulate=True, colsum_out=gb)
        assert len(ops.TRACE_KERNELS) == 1                          # one batched launch
    finally:
        ops.TRACE_KERNELS = None
    want = base.double() + dy.double().t() @ x.double()
    close(got, want, 2e-6, 1e-4, "fp32 split-K dW")
    close(gb, bb.double() + dy.double().sum(0), 2e-6, 1e-4, "fp32 split-K db")
    # (c) 128 x 128 tiles vs 64 x 64 tiles: M = 4096 rows takes the big tiles, the same rows in two halves of 64... compare with the product of a
    # row subset small enough for the 64 x 64 kernel (N = 64 < 128)
    a = torch.randn(4096, 512, device="cuda", generator=g)
    w = torch.randn(1024, 512, device="cuda", generator=g)
    big = ops.gemm(a, w)                                            # 32 x 8 = 256 tiles of 128 x 128
    small = torch.cat([ops.gemm(a, w[i:i + 64].contiguous()) for i in range(0, 1024, 64)], 1)     # N = 64 per product: 64 x 64 tiles
    assert torch.equal(big, small)


def test_bench_launches_its_own_ranks(dx):
    """`python bench.py --gpus 2` started plainly (no torchrun) spawns its two ranks as child processes itself -- before the parent
    has touched the GPU -- and relays rank 0's line.  Here both ranks share this box's one GPU over gloo (RCCL wants a GPU per
    rank); on an 8-GPU node the same command runs one rank per GPU over RCCL."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, DINOX_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch-size", "16",
                        "--no-cpu-baseline", "--no-secondary"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 32 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(32 / (d["ms_per_step"] * 1e-3), rel=1e-3) and "cpu_baseline" not in d
    fired, total = (int(x) for x in d["config"]["grad_buckets_launched_during_backward"].split("/"))
    # a few warm-up steps timed the exchange from backward against the exchange after it; the timed region ran the faster
    probe = d["config"]["grad_exchange_probe"]
    assert probe["chosen"] in ("overlapped", "after_backward") and probe["ms_per_step_overlapped"] > 0 and probe["ms_per_step_after_backward"] > 0
    assert (probe["chosen"] == "overlapped") == (probe["ms_per_step_overlapped"] <= probe["ms_per_step_after_backward"])
    assert total >= 3 and (fired >= total - 1 if probe["chosen"] == "overlapped" else fired == 0)
    assert set(d["step_ms_split"]) == {"fwd_student", "fwd_teacher", "loss", "bwd", "comm_exposed", "optimiser_tail"}
    # the line says what the collectives ran on (the driver's 8-GPU record can be checked against DESIGN.md section 5's prediction)
    c = d["config"]
    assert c["dist_backend"] == "gloo" and c["rccl_ranks_seen"] == 2 and c["comm_exposed_ms_max_over_ranks"] >= 0.0
    assert len(c["grad_bucket_bytes"]) == total and sum(c["grad_bucket_bytes"]) == c["grad_bytes_per_step"] and c["centre_allreduce_bytes"] == 4 * 8192


def test_gemm_timer_samples_launches_and_prices_families(dx):
    """bench.py's kernel timer (ops.GemmTimer) counts EVERY launch with its algorithmic flops / bytes and brackets one in `every` with HIP
    events; a family's time is estimated per shape (mean timed duration x launches).  every=1 and every=4 must agree on the counts and, on
    the same launches, on the time to within launch-to-launch noise; results of the products are untouched by the timer."""
    ops, _ = dx
    g = torch.Generator(device="cuda").manual_seed(3)
    a = (torch.randn(4096, 384, device="cuda", generator=g) * 0.5).bfloat16()
    w1 = (torch.randn(1536, 384, device="cuda", generator=g) * 0.5).bfloat16()
    w2 = (torch.randn(384, 1536, device="cuda", generator=g) * 0.5).bfloat16()
    ref1, ref2 = ops.gemm(a, w1), None
    ref2 = ops.gemm(ref1, w2)
    out = {}
    for every in (1, 4):
        t = ops.GemmTimer(every=every)
        with t:                                                      # dinox_gemm_timer_start .. _stop: the library brackets its own launches
            for _ in range(40):
                h = ops.gemm(a, w1)
                y = ops.gemm(h, w2)
        torch.cuda.synchronize()
        assert torch.equal(h, ref1) and torch.equal(y, ref2)
        out[every] = t.summary()
    for every, s in out.items():
        assert sum(d["launches"] for d in s.values()) == 80
        assert sum(d["flops"] for d in s.values()) == pytest.approx(40 * 2 * (2.0 * 4096 * 384 * 1536))
        assert all(d["ms"] > 0 and d["bytes"] > 0 for d in s.values())
    assert sum(d["timed"] for d in out[1].values()) == 80
    n4 = sum(d["timed"] for d in out[4].values())
    assert 8 <= n4 <= 36, n4                                     # one in four, by a fixed hash of the launch counter
    ms1, ms4 = (sum(d["ms"] for d in out[e].values()) for e in (1, 4))
    assert ms4 == pytest.approx(ms1, rel=0.5)


def test_full_size_step_properties(dx):
    """The headline workload itself (ViT-S/16 224, scale-aware, 256 samples = 512 views per step, bf16) -- too big for the CPU oracle, so it
    is held to size-independent properties:
      * determinism: two runs of two optimiser steps from the same state end bit-identical (weights, Adam moments, centre);
      * batch additivity: with the centre frozen (momentum 1) and accumulation_steps = 2, two half batches of 128 samples give the
        optimiser the gradient of the mean loss over all 256 -- so ONE optimiser step taken that way must land where the plain
        bs-256 step lands (to bf16 / summation-order round-off), and the logged quantities must be finite and sane
        (loss below the entropy wall ln 8192 + Gram term, every weight moved by at most lr)."""
    ops, arch = dx
    from dinox.engine import StepHyperParams, TrainEngine
    kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
    torch.manual_seed(0)
    ref = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
    torch.nn.init.xavier_uniform_(ref.backbone.scale_embed.mlp[2].weight)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    del ref
    g = torch.Generator().manual_seed(1)
    B = 256
    v1, v2 = torch.randn(B, 3, 224, 224, generator=g), torch.randn(B, 3, 224, 224, generator=g)
    sp = torch.rand(B, 3, generator=g) * 0.5 + 0.5
    full = (torch.cat([v1, v2], 0).to(DEV), torch.cat([sp, sp], 0).to(DEV))
    halves = [(torch.cat([v1[i:i + 128], v2[i:i + 128]], 0).to(DEV), torch.cat([sp[i:i + 128], sp[i:i + 128]], 0).to(DEV)) for i in (0, 128)]
    del v1, v2

    def run(batches, accum, steps):
        student = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
        teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 8192)
        student.load_state_dict(sd)
        teacher.load_state_dict(sd)
        eng = TrainEngine(student.to(DEV), teacher.to(DEV), 8192, StepHyperParams(lr=1e-3, warmup_steps=1, max_steps=10, ema=0.99, center_momentum=1.0),
                          amp_dtype=torch.bfloat16, accumulation_steps=accum)
        for _ in range(steps):
            for b, s in batches:
                eng.step(b, s)
        sc = eng.scalars()
        return eng.flat_p.clone(), eng.adam_m.clone(), eng.center.clone(), sc

    p1, m1, c1, s1 = run([full], 1, 2)
    p2, m2, c2, s2 = run([full], 1, 2)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(c1, c2) and s1 == s2
    assert math.isfinite(s1["loss"]) and 0 < s1["loss"] < math.log(8192) + 2.0 and math.isfinite(s1["grad_norm"]) and s1["grad_norm"] > 0
    pa, ma, _, _ = run([full], 1, 1)
    pb, mb, _, sb = run(halves, 2, 1)
    p0 = torch.cat([v.reshape(-1) for v in sd.values()])          # (not arena order -- only used for the size of the update below)
    assert float((pa - pb).abs().max()) <= 2.1e-3                  # an Adam step moves a weight by at most lr (sign noise on ~zero gradients)
    moved = (pa - pb).abs()
    assert float((moved <= 2e-5).double().mean()) > 0.97           # ... and almost all of them land on the same value
    assert rel_l2(mb, ma) < 2e-2                                   # first moments = 0.1 x gradient: the two gradients agree to bf16 round-off
    assert p0.numel() <= pa.numel()


def test_full_size_forward_rows_close_the_size_gap(dx):
    """The CPU oracle pins the headline MODEL at B = 2 (test_full_vit_small_16_step_matches_oracle); the headline SIZE (512 views per
    network) is beyond it.  This closes the gap on the device, through the engine's own oracle-pinned fp32 parity mode:
      (1) fp32 mode at full size == fp32 mode on a 6-view batch of the same views, per view, to fp32 round-off (the exact-fp32 MFMA
          sums every product in the same order whatever M is): a grid overflow, a tile past 2^31 bytes or a size-dependent dispatch bug in
          any forward kernel would show here, on sampled views from both ends and the middle of the batch;
      (2) bf16 mode at full size (the kernels the dispatcher picks for M = 102 912: the ping-pong GEMMs, the persistent attention) is no
          further from that fp32 result than bf16 mode is on the small batch (the kernels the B = 2 oracle test pins), times 1.5."""
    ops, arch = dx
    kw = dict(img_size=224, patch=16, dim=384, depth=12, heads=6, num_registers=4, scale_aware=True)
    torch.manual_seed(0)
    net = arch.PatchViT(**kw)
    torch.nn.init.xavier_uniform_(net.scale_embed.mlp[2].weight)
    net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(11)
    V = 512
    x = torch.randn(V, 3, 224, 224, generator=g).to(DEV)
    sp = (torch.rand(V, 3, generator=g) * 0.5 + 0.5).to(DEV)
    idx = torch.tensor([0, 1, 200, 255, 256, 511], device=DEV)
    with torch.no_grad():
        with ops.compute_dtype(torch.float32):
            f32_full = net(x, spacing=sp)[idx].clone()
            f32_small = net(x[idx], spacing=sp[idx])
        with ops.compute_dtype(torch.bfloat16):
            ops.TRACE_KERNELS = []
            try:
                b16_full = net(x, spacing=sp)[idx].float()
                kernels_full = set(ops.TRACE_KERNELS)
            finally:
                ops.TRACE_KERNELS = None
            b16_small = net(x[idx], spacing=sp[idx]).float()
    assert f32_full.shape == (6, 201, 384) and bool(torch.isfinite(f32_full).all())
    close(f32_full, f32_small, 2e-6, 1e-5, "fp32 mode: full batch vs the same views in a small batch")
    assert {"gemm_bf16_nt_pp", "gemm_bf16_nt_pp128"} <= kernels_full, kernels_full         # the full size really runs on the round-3 kernels
    for i in range(6):
        d_full, d_small = rel_l2(b16_full[i], f32_full[i]), rel_l2(b16_small[i], f32_small[i])
        assert d_full <= 1.5 * d_small + 1e-4, (int(idx[i]), d_full, d_small)
