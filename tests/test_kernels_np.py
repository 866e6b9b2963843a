"""Check the hand-derived backward formulas (oracle/kernels_np.py, the math the HIP kernels
implement) against the oracle's autograd (oracle/dinox_oracle.py) on CPU."""
import numpy as np
import pytest
import torch

from oracle import dinox_oracle as O
from oracle import kernels_np as K

RNG = np.random.default_rng(7)


def T(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


def ok(a, b, rtol=1e-9, atol=1e-11):
    np.testing.assert_allclose(np.asarray(a), b.detach().numpy() if torch.is_tensor(b) else b, rtol=rtol, atol=atol)


def test_gelu():
    x = RNG.normal(size=(5, 33)) * 3
    xt = T(x, True)
    y = O.gelu_erf(xt)
    ok(K.gelu(x), y)
    y.sum().backward()
    ok(K.gelu_grad(x), xt.grad)


def test_layernorm():
    x, w, b, dy = RNG.normal(size=(4, 7, 48)), RNG.normal(size=48), RNG.normal(size=48), RNG.normal(size=(4, 7, 48))
    xt, wt, bt = T(x, True), T(w, True), T(b, True)
    y = O.layer_norm(xt, wt, bt)
    yk, mu, rstd = K.layernorm_fwd(x, w, b)
    ok(yk, y)
    y.backward(T(dy))
    dx, dw, db = K.layernorm_bwd(dy, x, w, mu, rstd)
    ok(dx, xt.grad); ok(dw, wt.grad); ok(db, bt.grad)


def test_linear():
    x, w, b, dy = RNG.normal(size=(3, 5, 16)), RNG.normal(size=(24, 16)), RNG.normal(size=24), RNG.normal(size=(3, 5, 24))
    xt, wt, bt = T(x, True), T(w, True), T(b, True)
    y = O.linear(xt, wt, bt)
    ok(K.linear_fwd(x, w, b), y)
    y.backward(T(dy))
    dx, dw, db = K.linear_bwd(dy, x, w)
    ok(dx, xt.grad); ok(dw, wt.grad); ok(db, bt.grad)


@pytest.mark.parametrize("B,N,h,d", [(2, 9, 2, 8), (1, 201, 2, 16)])
def test_attention_core(B, N, h, d):
    C = h * d
    qkv = RNG.normal(size=(B, N, 3 * C))
    do = RNG.normal(size=(B, N, C))
    # drive the oracle's attention() with identity projections so its output is the core
    p = {"qkv.weight": T(np.eye(3 * C)), "qkv.bias": T(np.zeros(3 * C)), "proj.weight": T(np.eye(C)), "proj.bias": T(np.zeros(C))}
    # attention() applies qkv to x of width C; emulate by feeding qkv through a (3C x 3C) identity: use x := qkv
    qt = T(qkv, True)
    B_, N_, _ = qt.shape
    q3 = qt.reshape(B_, N_, 3, h, d).permute(2, 0, 3, 1, 4)
    s = (q3[0] @ q3[1].transpose(-1, -2)) / (d ** 0.5)
    o = (O.softmax_lastdim(s) @ q3[2]).transpose(1, 2).reshape(B_, N_, C)
    ok_o, lse = K.attention_core_fwd(qkv, h)
    ok(ok_o, o)
    o.backward(T(do))
    ok(K.attention_core_bwd(do, qkv, ok_o, lse, h), qt.grad, rtol=1e-8, atol=1e-10)


def test_tokens():
    B, p, g, D, R = 3, 4, 3, 16, 2
    x = RNG.normal(size=(B, 3, p * g, p * g))
    w, b = RNG.normal(size=(D, 3, p, p)), RNG.normal(size=D)
    cls, pos, regs = RNG.normal(size=(1, 1, D)), RNG.normal(size=(1, 1 + g * g, D)), RNG.normal(size=(1, R, D))
    scale = RNG.normal(size=(B, 1, D))
    dt = RNG.normal(size=(B, 1 + g * g + R, D))
    wt, bt, ct, pt, rt, st = [T(a, True) for a in (w, b, cls, pos, regs, scale)]
    t = O.patch_embed(T(x), wt, bt, p)
    t = torch.cat([ct.expand(B, -1, -1), t], 1) + pt + st
    t = torch.cat([t, rt.expand(B, -1, -1)], 1)
    ok(K.tokens_fwd(x, w, b, cls, pos, regs, scale, p), t)
    t.backward(T(dt))
    dw, db, dcls, dpos, dregs, dscale = K.tokens_bwd(dt, x, w, p, R, True)
    ok(dw, wt.grad); ok(db, bt.grad); ok(dcls, ct.grad); ok(dpos, pt.grad); ok(dregs, rt.grad); ok(dscale, st.grad)


def test_dino_ce():
    s, t, c = RNG.normal(size=(8, 96)) * 3, RNG.normal(size=(8, 96)) * 2, RNG.normal(size=(1, 96)) * 0.1
    st_ = T(s, True)
    l = O.dino_loss(st_, T(t), T(c), 0.1, 0.04)
    assert K.dino_ce_fwd(s, t, c, 0.1, 0.04) == pytest.approx(float(l), rel=1e-10)
    l.backward()
    ok(K.dino_ce_bwd(s, t, c, 0.1, 0.04), st_.grad, rtol=1e-8, atol=1e-12)
    ok(K.center_update(c, t, 0.9), O.center_update(T(c), T(t), 0.9))


def test_gram():
    sf, tf = RNG.normal(size=(3, 12, 10)), RNG.normal(size=(3, 12, 10))
    sf[1, 4] = 0.0
    st_ = T(sf, True)
    l = O.gram_loss(st_, T(tf))
    assert K.gram_loss_fwd(sf, tf) == pytest.approx(float(l), rel=1e-10)
    l.backward()
    ok(K.gram_loss_bwd(sf, tf), st_.grad, rtol=1e-8, atol=1e-10)


def test_adamw_ema():
    hp = O.HyperParams(weight_decay=0.04)
    p, g = RNG.normal(size=50), RNG.normal(size=50)
    m, v, pt = RNG.normal(size=50) * 0.1, np.abs(RNG.normal(size=50)) * 0.01, RNG.normal(size=50)
    P_, G_, M_, V_ = T(p), T(g), T(m), T(v)
    O.adamw_update(P_, G_, M_, V_, 3, 2e-3, hp)
    teach = {"a": T(pt)}
    O.ema_update(teach, {"a": P_}, 0.996)
    pn, mn, vn, ptn, gsq = K.adamw_ema(p, g, m, v, pt, 3, 2e-3, 0.04, 0.9, 0.999, 1e-8, 0.996)
    ok(pn, P_); ok(mn, M_); ok(vn, V_); ok(ptn, teach["a"])
    assert gsq == pytest.approx(float((G_ * G_).sum()))
