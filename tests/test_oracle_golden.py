"""Pin the CPU oracle (oracle/dinox_oracle.py) to fixtures captured from the real reference
(tests/golden/make_golden.py) and to the known-answer values of SURVEY.md section 8c."""

import numpy as np
import pytest
import torch

from conftest import sub, t
from oracle import dinox_oracle as O


def close(a, b, rtol=2e-5, atol=2e-6):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    assert err <= atol + rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def test_attention_fwd_bwd(golden):
    g = golden("ops_attention.npz")
    w = {k: v.clone().requires_grad_(True) for k, v in sub(g, "w").items()}
    x = t(g["x"]).clone().requires_grad_(True)
    y = O.attention(x, w, "", int(g["heads"]))
    close(y, g["y"])
    y.backward(t(g["dy"]))
    close(x.grad, g["dx"])
    for k, v in sub(g, "g").items():
        close(w[k].grad, v)


def test_mlp_fwd_bwd(golden):
    g = golden("ops_mlp.npz")
    w = {k: v.clone().requires_grad_(True) for k, v in sub(g, "w").items()}
    x = t(g["x"]).clone().requires_grad_(True)
    y = O.mlp(x, w, "")
    close(y, g["y"])
    y.backward(t(g["dy"]))
    close(x.grad, g["dx"])
    for k, v in sub(g, "g").items():
        close(w[k].grad, v)


def test_scale_embed_fwd_bwd(golden):
    g = golden("ops_scale_embed.npz")
    w = {k: v.clone().requires_grad_(True) for k, v in sub(g, "w").items()}
    sp = t(g["spacing"]).clone().requires_grad_(True)
    y = O.scale_embedding(sp, w, "")
    close(y, g["y"])
    y.backward(t(g["dy"]))
    close(sp.grad, g["dspacing"], rtol=1e-4)
    for k, v in sub(g, "g").items():
        close(w[k].grad, v, rtol=1e-4)


def test_dino_loss(golden):
    g = golden("dino_loss.npz")
    s = t(g["s"]).clone().requires_grad_(True)
    c0 = t(g["center0"])
    l1 = O.dino_loss(s, t(g["t"]), c0, float(g["student_temp"]), float(g["teacher_temp"]))
    close(l1, g["loss1"])
    l1.backward()
    close(s.grad, g["ds1"])
    c1 = O.center_update(c0, t(g["t"]), float(g["momentum"]))
    close(c1, g["center1"])
    l2 = O.dino_loss(s.detach(), t(g["t2"]), c1, 0.1, 0.04)
    close(l2, g["loss2"])
    close(O.center_update(c1, t(g["t2"]), 0.9), g["center2"])


def test_gram_loss(golden):
    g = golden("gram_loss.npz")
    sf = t(g["sf"]).clone().requires_grad_(True)
    l = O.gram_loss(sf, t(g["tf"]))
    close(l, g["loss"])
    l.backward()
    close(sf.grad, g["dsf"])
    close(O.gram_matrix(sf.detach()[:, 1:]), g["gram_s"])


def _cfg(arr):
    img, patch, dim, depth, heads, regs, sa, out = [int(v) for v in arr]
    return O.VitCfg(img_size=img, patch=patch, dim=dim, depth=depth, heads=heads, num_registers=regs,
                    scale_aware=bool(sa), out_dim=out)


def test_vit_tiny_forward_loss_grads(golden):
    g = golden("vit_tiny.npz")
    cfg = _cfg(g["cfg"])
    student = {k: v.clone().requires_grad_(True) for k, v in sub(g, "student").items()}
    teacher = sub(g, "teacher")
    x, sp = t(g["x"]), t(g["spacing"])
    taps = {}
    s_feats = O.vit_forward(student, x, sp, cfg, pre="backbone.", taps=taps)
    close(s_feats, g["s_feats"], rtol=5e-5)
    for i in range(cfg.depth):
        close(taps[f"block{i}"], g[f"tap/block{i}"], rtol=5e-5)
    with torch.no_grad():
        t_feats = O.vit_forward(teacher, x, sp, cfg, pre="backbone.")
        t_out = O.head_forward(teacher, t_feats[:, 0])
        close(O.vit_forward({k: v.detach() for k, v in student.items()}, x, None, cfg, pre="backbone."),
              g["feats_nospacing"], rtol=5e-5)
    close(t_feats, g["t_feats"], rtol=5e-5)
    s_out = O.head_forward(student, s_feats[:, 0])
    close(s_out, g["s_out"], rtol=5e-5)
    close(t_out, g["t_out"], rtol=5e-5)
    l_dino = O.dino_loss(s_out, t_out, t(g["center"]), 0.1, 0.04)
    l_gram = O.gram_loss(s_feats, t_feats)
    close(l_dino, g["loss_dino"], rtol=5e-5)
    close(l_gram, g["loss_gram"], rtol=5e-5)
    (l_dino + l_gram).backward()
    close(O.center_update(t(g["center"]), t_out, 0.9), g["center_after"])
    order = [str(n) for n in g["param_order"]]
    assert order == list(O.param_shapes(cfg).keys())
    for n in order:
        close(student[n].grad, g[f"grad/{n}"], rtol=2e-4, atol=1e-7)


def test_vit_plain_forward(golden):
    g = golden("vit_plain.npz")
    cfg = _cfg(g["cfg"])
    cfg.mlp_ratio = 2.0
    y = O.vit_forward(sub(g, "w"), t(g["x"]), None, cfg)
    close(y, g["y"], rtol=5e-5)
    assert y.shape == (3, 1 + 4, 32)


def _close_masked(a, b, noisy, rtol, atol, noisy_atol):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs()
    scale = b.abs().max().item()
    ok = err[~noisy]
    assert ok.numel() == 0 or ok.max().item() <= atol + rtol * scale, f"{ok.max().item():.3e} vs {scale:.3e}"
    assert noisy.sum() == 0 or err[noisy].max().item() <= noisy_atol


def test_three_training_steps(golden):
    g = golden("step_tiny.npz")
    cfg = _cfg(g["cfg"])
    lr, min_lr, warm, max_steps, wd, ema, ts, tt, cm, gw = [float(v) for v in g["hp"]]
    hp = O.HyperParams(lr=lr, min_lr=min_lr, warmup_steps=int(warm), max_steps=int(max_steps), weight_decay=wd,
                       ema=ema, student_temp=ts, teacher_temp=tt, center_momentum=cm, gram_weight=gw)
    st = O.init_state(cfg, sub(g, "init"))
    noisy = {}
    for step in range(3):
        r = O.train_step(st, t(g[f"batch{step}"]), t(g[f"spacing{step}"]), hp)
        assert r["loss"] == pytest.approx(float(g["losses"][step]), rel=2e-4)
        assert r["dino"] == pytest.approx(float(g["dinos"][step]), rel=2e-4)
        assert r["gram"] == pytest.approx(float(g["grams"][step]), rel=2e-4)
        assert r["grad_norm"] == pytest.approx(float(g["grad_norms"][step]), rel=5e-4)
        assert r["lr"] == pytest.approx(float(g["lrs"][step]), rel=1e-12)
        # Adam turns a numerically-zero gradient (e.g. the key bias, to which softmax is invariant)
        # into a +-lr update whose sign is round-off: compare those elements at the lr scale only.
        for k, gr in r["grads"].items():
            noisy[k] = noisy.get(k, torch.zeros_like(gr, dtype=torch.bool)) | (gr.abs() < 1e-6)
        if step == 0:
            for k, v in sub(g, "student1").items():
                _close_masked(st.student[k], v, noisy[k], rtol=1e-4, atol=1e-6, noisy_atol=2.1 * r["lr"])
    for k, v in sub(g, "student3").items():
        _close_masked(st.student[k], v, noisy[k], rtol=5e-4, atol=5e-6, noisy_atol=2.1 * 3e-3)
    for k, v in sub(g, "teacher3").items():
        _close_masked(st.teacher[k], v, noisy[k], rtol=5e-4, atol=5e-6, noisy_atol=2.1 * 3e-3)
    close(st.center, g["center3"], rtol=1e-4)


def test_get_lr_grid(golden):
    rows = golden("get_lr.npz")["rows"]
    for step, total, warm, want in rows:
        got = O.get_lr(int(step), None if total < 0 else int(total), int(warm), 1e-4, 1e-6)
        assert got == pytest.approx(want, rel=1e-12, abs=0)


# ---- known answers recorded in SURVEY.md section 8c (reference run on torch 2.10 CPU fp32) ----
def test_survey_known_answers_losses():
    S = O.det(8, 128, f=0.37)
    T = 2 * O.det(8, 128, f=0.91, ph=0.5)
    c = torch.zeros(1, 128)
    assert float(O.dino_loss(S, T, c, 0.1, 0.04)) == pytest.approx(12.77463341, rel=1e-6)
    c = O.center_update(c, T, 0.9)
    assert float(c.sum()) == pytest.approx(0.03956433, rel=1e-4)
    np.testing.assert_allclose(c[0, :3].numpy(), [-0.00465757, 0.01305797, 0.02068612], rtol=1e-5)
    assert float(O.dino_loss(S, T, c, 0.1, 0.04)) == pytest.approx(12.77092743, rel=1e-6)
    sf = O.det(4, 21, 16, f=0.13)
    tf = O.det(4, 21, 16, f=0.29, ph=1.0)
    assert float(O.gram_loss(sf, tf)) == pytest.approx(0.98794204, rel=1e-6)


def test_survey_known_answers_lr():
    f = lambda s, tot: O.get_lr(s, tot, 2500, 1e-4, 1e-6)
    assert f(0, None) == pytest.approx(4e-08)
    assert f(2499, 5000) == pytest.approx(1e-4)
    assert f(2500, 5000) == pytest.approx(1e-4)
    assert f(3750, 5000) == pytest.approx(5.05e-05)
    assert f(5000, 5000) == pytest.approx(1e-06)
    assert f(9999, None) == pytest.approx(1e-4)


def test_survey_known_answer_model():
    cfg = O.VitCfg(img_size=56, patch=14, dim=64, depth=2, heads=2, mlp_ratio=2.0, num_registers=2,
                   scale_aware=True, out_dim=128)
    p = {}
    for i, (n, s) in enumerate(O.param_shapes(cfg).items()):
        v = 0.05 * O.det(*s, f=0.011 * (i + 1), ph=0.3 * i)
        if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("mlp.3.weight") or n == "backbone.norm.weight":
            v = v + 1.0
        p[n] = v.requires_grad_(True)
    x = O.det(2, 3, 56, 56, f=0.0173)
    sp = torch.tensor([[0.5, 0.5, 1.0], [1.5, 1.5, 5.0]])
    feats = O.vit_forward(p, x, sp, cfg, pre="backbone.")
    assert feats.shape == (2, 19, 64)
    assert float(feats.sum()) == pytest.approx(-3.676014, abs=2e-3)
    assert float(feats.abs().mean()) == pytest.approx(0.80310512, rel=1e-5)
    np.testing.assert_allclose(feats[0, 0, :3].detach().numpy(), [1.28862894, 0.45710021, -0.07357411], rtol=2e-5, atol=2e-6)
    out = O.head_forward(p, feats[:, 0])
    assert float(out.abs().mean()) == pytest.approx(0.03626285, rel=1e-4)
    out.pow(2).mean().backward()
    assert float(p["backbone.blocks.0.attn.qkv.weight"].grad.abs().mean()) == pytest.approx(6.03e-08, rel=2e-2)
    assert float(p["backbone.scale_embed.mlp.0.weight"].grad.abs().mean()) == pytest.approx(1.24058e-05, rel=1e-3)


@pytest.mark.parametrize("tag", ["small", "mm"])
def test_koleo_loss_golden(golden, tag):
    """oracle.koleo_loss vs the reference KoLeoLoss (phase5_big_run.py:742-773): loss and input gradient, on torch.cdist's
    direct route (10 rows) and its matmul route (40 rows)."""
    g = golden("koleo_loss.npz")
    x = t(g[f"{tag}_x"]).requires_grad_(True)
    l = O.koleo_loss(x)
    l.backward()
    # the 40-row case holds a close pair (d ~ 0.03): cdist's matmul route (|a|^2+|b|^2-2ab) loses ~3 digits of that distance to
    # cancellation, the direct distance here does not; 5e-4 covers it (the path's fp32 gate is 1e-3)
    assert float(l) == pytest.approx(float(g[f"{tag}_loss"]), rel=1e-5 if tag == "small" else 5e-4, abs=1e-6)
    ref = t(g[f"{tag}_dx"])
    assert float((x.grad - ref).norm() / ref.norm()) < (2e-5 if tag == "small" else 1e-3)


def test_multicrop_extension_anchors():
    """The multi-crop extension is not in the reference; its oracle is anchored three ways: the pair-loop loss with
    no local crops IS the reference-pinned 2-view loss; the bicubic matrix equals torch's own bicubic resize; a model fed
    native-size input never touches the interpolation."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    B, K = 6, 96
    s, tt_, c = 3 * torch.randn(2 * B, K, generator=g), 1.5 * torch.randn(2 * B, K, generator=g), 0.1 * torch.randn(1, K, generator=g)
    assert float(O.dino_loss_multicrop(s, tt_, c, 0.1, 0.04)) == pytest.approx(float(O.dino_loss(s, tt_, c, 0.1, 0.04)), rel=1e-6)
    # 2 global + 3 local views: every (teacher view, other student view) pair, averaged
    sl = 3 * torch.randn(5 * B, K, generator=g)
    tp = torch.softmax((tt_ - c) / 0.04, -1).reshape(2, B, K)
    ls = torch.log_softmax(sl / 0.1, -1).reshape(5, B, K)
    want = sum(-(tp[q] * ls[v]).sum(-1).mean() for q in range(2) for v in range(5) if v != q) / 8
    assert float(O.dino_loss_multicrop(sl, tt_, c, 0.1, 0.04)) == pytest.approx(float(want), rel=1e-6)
    for g_in, g_out in [(14, 6), (4, 2), (7, 7), (4, 9)]:
        pos = torch.randn(1, 1 + g_in * g_in, 8, generator=g)
        grid = pos[0, 1:].reshape(g_in, g_in, 8).permute(2, 0, 1)[None]
        ref = F.interpolate(grid, size=(g_out, g_out), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).reshape(g_out * g_out, 8)
        got = O.interpolate_pos(pos, g_out)
        assert torch.equal(got[:, :1], pos[:, :1]) and torch.allclose(got[0, 1:], ref, atol=2e-6), (g_in, g_out)


def _rl(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("policy", ["f32loss", "cpu"])
def test_oracle_autocast_mode_is_the_reference_autocast_arithmetic(golden, policy):
    """The oracle's ``autocast_bf16()`` mode (used on the GPU box to price the reference's --amp rounding at model sizes no
    fixture covers) against the real reference run under torch.autocast(bfloat16) (vit_tiny_autocast.npz): the two must
    agree far more closely than either agrees with fp32, for the feats, both losses and every parameter gradient."""
    g, a = golden("vit_tiny.npz"), golden("vit_tiny_autocast.npz")
    cfg = _cfg(g["cfg"])
    student = {k: v.clone().requires_grad_(True) for k, v in sub(g, "student").items()}
    teacher = sub(g, "teacher")
    x, sp = t(g["x"]), t(g["spacing"])
    with O.autocast_bf16(loss_fp32=(policy == "f32loss")):
        s_feats = O.vit_forward(student, x, sp, cfg, pre="backbone.")
        with torch.no_grad():
            t_feats = O.vit_forward(teacher, x, sp, cfg, pre="backbone.")
            t_out = O.head_forward(teacher, t_feats[:, 0])
        s_out = O.head_forward(student, s_feats[:, 0])
        l_dino = O.dino_loss(s_out, t_out, t(g["center"]), 0.1, 0.04)
        l_gram = O.gram_loss(s_feats, t_feats)
        loss = l_dino + l_gram
    loss.backward()
    assert s_out.dtype == torch.bfloat16 and s_feats.dtype == torch.float32        # the dtype flow the fixture recorded
    p = policy
    assert _rl(s_feats, a[f"{p}/s_feats"]) < 1e-6 and _rl(s_out.float(), a[f"{p}/s_out"]) < 1e-6
    assert float(l_dino) == pytest.approx(float(a[f"{p}/loss_dino"]), rel=1e-6)
    assert float(l_gram) == pytest.approx(float(a[f"{p}/loss_gram"]), rel=1e-6)
    for n in [str(s) for s in g["param_order"]]:
        ref_amp, ref_f32 = a[f"{p}/grad/{n}"], g[f"grad/{n}"]
        assert _rl(student[n].grad, ref_amp) <= 0.02 * _rl(ref_amp, ref_f32) + 1e-7, n
    # and the size of the thing being priced: the reference's own --amp step sits 2.5-5 % (rel. L2) from its fp32 step per parameter
    d = [_rl(a[f"{p}/grad/{n}"], g[f"grad/{n}"]) for n in [str(s) for s in g["param_order"]]]
    assert 0.02 < min(d) and max(d) < 0.06
