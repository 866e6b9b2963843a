"""NumPy float64 restatement of each hot-path kernel's math, forward AND hand-derived backward.

TEST INFRASTRUCTURE (same rules as dinox_oracle.py).  Purpose: the HIP kernels in
``dino-x_amd/csrc`` implement hand-derived backward formulas; this file states the same formulas
in float64 NumPy so that ``tests/test_kernels_np.py`` can check the *derivations* against the
oracle's autograd on CPU before any GPU time is spent, and so GPU kernel tests have a per-kernel
reference that does not depend on torch.

Reference lines: see dinox_oracle.py (same functions, per kernel).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf

F = np.float64
SQRT1_2 = 1.0 / math.sqrt(2.0)
INV_SQRT_2PI = 1.0 / math.sqrt(2.0 * math.pi)


# ---- GELU (exact erf) ---------------------------------------------------------------------
def gelu(x):
    return 0.5 * x * (1.0 + erf(x * SQRT1_2))


def gelu_grad(x):
    return 0.5 * (1.0 + erf(x * SQRT1_2)) + x * np.exp(-0.5 * x * x) * INV_SQRT_2PI


# ---- LayerNorm ----------------------------------------------------------------------------
def layernorm_fwd(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    rstd = 1.0 / np.sqrt((xc * xc).mean(-1, keepdims=True) + eps)
    return xc * rstd * w + b, mu, rstd


def layernorm_bwd(dy, x, w, mu, rstd):
    xh = (x - mu) * rstd
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdims=True) - xh * (g * xh).mean(-1, keepdims=True))
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xh).sum(red), dy.sum(red)


# ---- Linear -------------------------------------------------------------------------------
def linear_fwd(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def linear_bwd(dy, x, w):
    dy2 = dy.reshape(-1, dy.shape[-1])
    x2 = x.reshape(-1, x.shape[-1])
    return dy @ w, dy2.T @ x2, dy2.sum(0)


# ---- multi-head attention core (qkv packed as the reference packs it) -----------------------
def _split_qkv(qkv, heads):
    B, N, C3 = qkv.shape
    d = C3 // 3 // heads
    t = qkv.reshape(B, N, 3, heads, d).transpose(2, 0, 3, 1, 4)
    return t[0], t[1], t[2], d


def attention_core_fwd(qkv, heads):
    """qkv (B,N,3C) -> o (B,N,C), plus per-row log-sum-exp of the scaled scores (B,h,N)."""
    q, k, v, d = _split_qkv(qkv, heads)
    s = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(d)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    l = e.sum(-1, keepdims=True)
    p = e / l
    o = p @ v
    B, h, N, _ = o.shape
    return o.transpose(0, 2, 1, 3).reshape(B, N, h * d), (m + np.log(l))[..., 0]


def attention_core_bwd(do, qkv, o, lse, heads):
    """Flash-style: P recomputed from lse, delta = rowsum(dO*O)."""
    q, k, v, d = _split_qkv(qkv, heads)
    B, h, N, _ = q.shape
    sc = 1.0 / math.sqrt(d)
    doh = do.reshape(B, N, h, d).transpose(0, 2, 1, 3)
    oh = o.reshape(B, N, h, d).transpose(0, 2, 1, 3)
    p = np.exp((q @ k.transpose(0, 1, 3, 2)) * sc - lse[..., None])
    dv = p.transpose(0, 1, 3, 2) @ doh
    dp = doh @ v.transpose(0, 1, 3, 2)
    delta = (doh * oh).sum(-1, keepdims=True)
    ds = p * (dp - delta)
    dq = (ds @ k) * sc
    dk = (ds.transpose(0, 1, 3, 2) @ q) * sc
    dqkv = np.stack([dq, dk, dv], 0).transpose(1, 3, 0, 2, 4).reshape(B, N, 3 * h * d)
    return dqkv


# ---- patch embedding + token assembly --------------------------------------------------------
def unfold_patches(x, patch):
    B, C, H, W = x.shape
    g = H // patch
    return x.reshape(B, C, g, patch, g, patch).transpose(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * patch * patch)


def tokens_fwd(x, w, b, cls, pos, regs, scale, patch):
    """patch-embed GEMM + [CLS | patches] + pos (+ scale (B,1,D)) then registers appended."""
    t = unfold_patches(x, patch) @ w.reshape(w.shape[0], -1).T + b
    B = x.shape[0]
    t = np.concatenate([np.broadcast_to(cls, (B, 1, cls.shape[-1])), t], 1) + pos
    if scale is not None:
        t = t + scale
    if regs is not None:
        t = np.concatenate([t, np.broadcast_to(regs, (B,) + regs.shape[1:])], 1)
    return t


def tokens_bwd(dt, x, w, patch, n_regs, has_scale):
    B = dt.shape[0]
    P = (x.shape[2] // patch) ** 2
    d_body = dt[:, :1 + P]
    dpatch = d_body[:, 1:]
    u = unfold_patches(x, patch).reshape(B * P, -1)
    dw = (dpatch.reshape(B * P, -1).T @ u).reshape(w.shape)
    db = dpatch.sum((0, 1))
    dcls = d_body[:, :1].sum(0, keepdims=True)
    dpos = d_body.sum(0, keepdims=True)
    dregs = dt[:, 1 + P:].sum(0, keepdims=True) if n_regs else None
    dscale = d_body.sum(1, keepdims=True) if has_scale else None
    return dw, db, dcls, dpos, dregs, dscale


# ---- DINO centring/sharpening cross-entropy ---------------------------------------------------
def dino_ce_fwd(s, t, center, ts, tt):
    B2 = s.shape[0]
    B = B2 // 2
    zt = (t - center) / tt
    zt = zt - zt.max(-1, keepdims=True)
    tp = np.exp(zt)
    tp /= tp.sum(-1, keepdims=True)
    zs = s / ts
    zs = zs - zs.max(-1, keepdims=True)
    ls = zs - np.log(np.exp(zs).sum(-1, keepdims=True))
    pair = (np.arange(B2) + B) % B2          # student row i is scored against teacher row i+-B
    return -(tp[pair] * ls).sum() / B2


def dino_ce_bwd(s, t, center, ts, tt):
    B2 = s.shape[0]
    B = B2 // 2
    zt = (t - center) / tt
    zt = zt - zt.max(-1, keepdims=True)
    tp = np.exp(zt)
    tp /= tp.sum(-1, keepdims=True)
    zs = s / ts
    zs = zs - zs.max(-1, keepdims=True)
    sp = np.exp(zs)
    sp /= sp.sum(-1, keepdims=True)
    pair = (np.arange(B2) + B) % B2
    return (sp - tp[pair]) / (ts * B2)


def center_update(center, t, momentum):
    return center * momentum + t.mean(0, keepdims=True) * (1 - momentum)


# ---- Gram anchoring ---------------------------------------------------------------------------
def _normalize(x, eps=1e-12):
    nrm = np.maximum(np.sqrt((x * x).sum(-1, keepdims=True)), eps)
    return x / nrm, nrm


def gram_loss_fwd(sf, tf):
    xs, _ = _normalize(sf[:, 1:])
    xt, _ = _normalize(tf[:, 1:])
    gs = xs @ xs.transpose(0, 2, 1)
    gt = xt @ xt.transpose(0, 2, 1)
    return ((gs - gt) ** 2).mean()


def gram_loss_bwd(sf, tf, eps=1e-12):
    """d loss / d sf (CLS row gets zero)."""
    x = sf[:, 1:]
    xs, nrm = _normalize(x, eps)
    xt, _ = _normalize(tf[:, 1:])
    V, T, _ = xs.shape
    diff = xs @ xs.transpose(0, 2, 1) - xt @ xt.transpose(0, 2, 1)
    dxh = (4.0 / (V * T * T)) * (diff @ xs)                       # dG symmetric: (dG+dG^T) Xh
    raw = np.sqrt((x * x).sum(-1, keepdims=True))
    proj = (xs * dxh).sum(-1, keepdims=True)
    dx = np.where(raw > eps, (dxh - xs * proj) / nrm, dxh / eps)  # clamp passes no grad to the norm
    out = np.zeros_like(sf)
    out[:, 1:] = dx
    return out


# ---- optimiser tail ---------------------------------------------------------------------------
def adamw_ema(p, g, m, v, pt, t, lr, wd, b1, b2, eps, ema):
    """Returns new (p, m, v, teacher) and sum(g^2); t is the 1-based optimiser step."""
    p = p * (1.0 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = np.sqrt(v) / math.sqrt(1 - b2 ** t) + eps
    p = p - (lr / (1 - b1 ** t)) * m / denom
    pt = ema * pt + (1 - ema) * p
    return p, m, v, pt, float((g * g).sum())
