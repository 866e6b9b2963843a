"""CPU checks of the drop-in surface (zoo.arch / zoo.hub / zoo.encode / zoo.models / dinox.schedule):
constructor signatures, attribute and sub-module names, state_dict keys, checkpoint migration, hub
round trip, error behaviour.  These mirror the assertions of the reference's own boundary tests
(reference tests/test_scale_embedding.py, tests/test_zoo_hub_peft.py); forward passes need the GPU and
live in test_gpu_parity.py -- on CPU tensors forward must fail loudly, which is asserted here."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden, sub
from oracle import dinox_oracle as O

import zoo.arch as arch
import zoo.encode as enc
import zoo.hub as hub


def tiny(**kw):
    cfg = dict(img_size=56, patch=14, dim=64, depth=2, heads=2)
    cfg.update(kw)
    return arch.PatchViT(**cfg)


def test_constructor_defaults_and_attributes():
    m = arch.PatchViT()
    assert (m.img_size, m.patch, m.dim, len(m.blocks), m.num_registers, m.scale_aware, m.use_grad_checkpoint) == (224, 16, 384, 6, 4, False, False)
    assert m.blocks[0].attn.num_heads == 6
    assert m.pos_embed.shape == (1, 197, 384) and m.cls_token.shape == (1, 1, 384) and m.registers.shape == (1, 4, 384)
    assert m.patch_embed.weight.shape == (384, 3, 16, 16)


def test_scale_embed_attribute_iff_scale_aware():
    assert hasattr(tiny(scale_aware=True), "scale_embed")
    assert not hasattr(tiny(scale_aware=False), "scale_embed")
    assert not hasattr(tiny(num_registers=0), "registers")
    extra = sum(p.numel() for p in tiny(scale_aware=True).parameters()) - sum(p.numel() for p in tiny().parameters())
    assert 0 < extra < 5000          # reference tests/test_scale_embedding.py:174-184


def test_submodule_names_are_linear_for_peft():
    m = tiny()
    names = dict(m.named_modules())
    for leaf in ("blocks.0.attn.qkv", "blocks.0.attn.proj", "blocks.1.mlp.fc1", "blocks.1.mlp.fc2"):
        assert isinstance(names[leaf], nn.Linear), leaf
    assert isinstance(names["blocks.0.norm1"], nn.LayerNorm) and isinstance(m.norm, nn.LayerNorm)
    assert isinstance(m.patch_embed, nn.Conv2d)
    se = arch.ScaleEmbedding(64)
    assert isinstance(se.mlp, nn.Sequential) and len(se.mlp) == 4
    assert float(se.mlp[2].weight.abs().sum()) == 0.0 and float(se.mlp[2].bias.abs().sum()) == 0.0


@pytest.mark.parametrize("sa,regs", [(True, 4), (False, 0), (True, 2)])
def test_state_dict_keys_and_order_match_reference(sa, regs):
    cfg = O.VitCfg(img_size=56, patch=14, dim=64, depth=2, heads=2, num_registers=regs, scale_aware=sa, out_dim=128)
    m = arch.DinoStudentTeacher(tiny(num_registers=regs, scale_aware=sa), 128)
    got = {n: tuple(p.shape) for n, p in m.named_parameters()}
    want = O.param_shapes(cfg)           # oracle table, itself pinned to the reference's named_parameters()
    assert list(got) == list(want)
    assert got == want
    assert list(m.state_dict().keys()) == list(want)      # no buffers


def test_init_is_bit_identical_to_reference_seed0(golden):
    ref = sub(golden("init_seed0.npz"), "sd")
    torch.manual_seed(0)
    m = arch.DinoStudentTeacher(arch.PatchViT(img_size=28, patch=14, dim=32, depth=2, heads=2, num_registers=2, scale_aware=True), out_dim=64)
    sd = m.state_dict()
    assert list(sd) == list(ref)
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


def test_teacher_loads_student_state_dict():
    s = arch.DinoStudentTeacher(tiny(scale_aware=True), 128)
    tt = arch.DinoStudentTeacher(tiny(scale_aware=True), 128)
    tt.load_state_dict(s.state_dict())
    for a, b in zip(s.parameters(), tt.parameters()):
        assert torch.equal(a, b)


def test_forward_on_cpu_fails_loudly():
    m = tiny(scale_aware=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 3, 56, 56), spacing=torch.ones(1, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        arch.ScaleEmbedding(64)(torch.ones(2, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc.encode(m, np.zeros((64, 64), dtype=np.float32))


# ---- checkpoint key migration (reference tests/test_zoo_hub_peft.py:107-188) -----------------
def test_migrate_old_attention_and_mlp_keys():
    old = {
        "backbone.blocks.0.attn.in_proj_weight": torch.zeros(1), "backbone.blocks.0.attn.in_proj_bias": torch.zeros(1),
        "backbone.blocks.0.attn.out_proj.weight": torch.zeros(1), "backbone.blocks.0.attn.out_proj.bias": torch.zeros(1),
        "backbone.blocks.0.mlp.0.weight": torch.zeros(1), "backbone.blocks.0.mlp.0.bias": torch.zeros(1),
        "blocks.3.mlp.2.weight": torch.zeros(1), "student.backbone.blocks.1.mlp.2.bias": torch.zeros(1),
        "backbone.scale_embed.mlp.0.weight": torch.zeros(1), "backbone.scale_embed.mlp.2.bias": torch.zeros(1),
        "backbone.norm.weight": torch.zeros(1), "head.0.weight": torch.zeros(1),
    }
    assert arch.needs_migration(old)
    new = arch.migrate_state_dict(old)
    assert list(new) == [
        "backbone.blocks.0.attn.qkv.weight", "backbone.blocks.0.attn.qkv.bias", "backbone.blocks.0.attn.proj.weight",
        "backbone.blocks.0.attn.proj.bias", "backbone.blocks.0.mlp.fc1.weight", "backbone.blocks.0.mlp.fc1.bias",
        "blocks.3.mlp.fc2.weight", "student.backbone.blocks.1.mlp.fc2.bias",
        "backbone.scale_embed.mlp.0.weight", "backbone.scale_embed.mlp.2.bias", "backbone.norm.weight", "head.0.weight"]
    assert not arch.needs_migration(new)
    assert "backbone.blocks.0.attn.in_proj_weight" in old          # input untouched
    assert not arch.needs_migration(tiny().state_dict())


# ---- hub (reference tests/test_zoo_hub_peft.py:197-264) ---------------------------------------
@pytest.mark.parametrize("safetensors", [False, True])
def test_hub_export_load_roundtrip(tmp_path, safetensors):
    m = tiny(scale_aware=True)
    out = hub.export_hub_checkpoint(m, tmp_path / "hubdir", use_safetensors=safetensors)
    cfg = json.loads((out / "config.json").read_text())
    assert cfg["dim"] == 64 and cfg["depth"] == 2 and cfg["heads"] == 2 and cfg["scale_aware"] is True
    m2 = hub.load_model(str(out))
    assert not m2.training and m2.scale_aware
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.allclose(a, b, atol=1e-6), k


def test_load_from_training_checkpoint(tmp_path):
    student = arch.DinoStudentTeacher(tiny(scale_aware=True), 128)
    payload = {"step": 7, "student": student.state_dict(), "teacher": student.state_dict(),
               "config": {"model": {"name": "custom", "patch": 14, "dim": 64, "depth": 2, "heads": 2, "mlp_ratio": 4.0, "out_dim": 128},
                          "img_size": 56, "scale_aware": True}}
    p = tmp_path / "checkpoint_00000007.pth"
    torch.save(payload, p)
    bb = hub.load_model(str(p))
    assert isinstance(bb, arch.PatchViT) and bb.dim == 64 and bb.img_size == 56 and bb.scale_aware and not bb.training
    for k, v in bb.state_dict().items():
        assert torch.equal(v, student.state_dict()["backbone." + k]), k
    # scale-aware weights are dropped when the config says the model is not scale-aware
    bb2 = hub.load_model(str(p), config_override={"scale_aware": False})
    assert not hasattr(bb2, "scale_embed")


def test_hub_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        hub.load_from_training_checkpoint(tmp_path / "missing.pth")
    with pytest.raises(FileNotFoundError):
        hub.load_from_hub_dir(tmp_path)
    (tmp_path / "config.json").write_text(json.dumps({"dim": 64, "depth": 1, "heads": 2, "patch": 14, "img_size": 28}))
    with pytest.raises(FileNotFoundError, match="No weights"):
        hub.load_from_hub_dir(tmp_path)
    assert hub.DEFAULT_CONFIG["depth"] == 6 and hub.DEFAULT_CONFIG["patch"] == 16


# ---- encode preprocessing + errors (reference tests/test_zoo_hub_peft.py:272-348) ---------------
def test_encode_preprocess_shapes_and_values():
    for shape in [(64, 64), (64, 64, 3), (3, 64, 64)]:
        x = enc.preprocess(np.full(shape, 40.0, dtype=np.float32), 56, "hu_float", 40.0, 400.0)
        assert x.shape == (3, 56, 56) and x.dtype == torch.float32
        np.testing.assert_allclose(x[:, 0, 0].numpy(), (0.5 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225]), rtol=1e-5)
    u16 = np.full((32, 32), 32768 + 400, dtype=np.uint16)          # HU = 40 -> window centre
    x = enc.preprocess(u16, 28, "hu16_png", 40.0, 400.0)
    assert abs(float(x[0, 0, 0]) - (0.5 - 0.485) / 0.229) < 1e-5
    x = enc.preprocess(np.full((32, 32), 0.25, dtype=np.float32), 28, "windowed_float", 0, 0)
    assert abs(float(x[2, 3, 3]) - (0.25 - 0.406) / 0.225) < 1e-5


def test_encode_errors():
    m = tiny()
    with pytest.raises(ValueError, match="Unknown input_format"):
        enc.encode(m, np.zeros((8, 8)), input_format="nope")
    with pytest.raises(ValueError, match="Unsupported image shape"):
        enc.encode(m, np.zeros((8, 8, 4)))
    with pytest.raises(ValueError, match="same length"):
        enc.encode_batch(m, [np.zeros((8, 8))], [])


def test_models_importable():
    import zoo.models as zm
    e = zm.DatasetEntry(name="lidc", modality="ct", organs=["lung"])
    assert e.preprocessing.hu_shift == 32768 and e.hu_range == (-1024, 3071)
    lin = zm.TrainingLineage(model_name="m", datasets=[zm.DatasetUsage(name="a", slices_used=1, weight=0.25),
                                                      zm.DatasetUsage(name="b", slices_used=2, weight=0.75)])
    assert lin.total_weight() == pytest.approx(1.0) and lin.timestamp
    zm.SliceMetadata(dataset="d", series_id="s", slice_idx=0, pixel_spacing_x=0.5, pixel_spacing_y=0.5, slice_thickness=1.0, image_path="x.png")


def test_get_lr_matches_golden(golden):
    from dinox.schedule import get_lr
    for step, total, warm, want in golden("get_lr.npz")["rows"]:
        assert get_lr(int(step), None if total < 0 else int(total), int(warm), 1e-4, 1e-6) == pytest.approx(want, rel=1e-12, abs=0)


def test_flatten_parameters_views_and_alignment():
    from dinox.engine import flatten_parameters
    m = arch.DinoStudentTeacher(tiny(scale_aware=True), 128)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat, params, offs = flatten_parameters(m)
    assert all(o % 4 == 0 for o in offs) and flat.numel() % 4 == 0
    for p, o in zip(params, offs):
        assert p.data_ptr() == flat.data_ptr() + 4 * o
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in m.parameters())
    m.load_state_dict(before)                       # in-place copy keeps the views
    assert float(flat.abs().sum()) > 0


def test_weight_cache_entries_die_with_their_tensor_and_unfold_sharing_is_scoped():
    """caches keyed by a raw address served stale data when a new tensor reused a freed one's memory.  The bf16
    weight cache now holds a weak reference to the very tensor object (entry gone when it is collected, `is`-checked on
    lookup); the unfold cache exists only inside ``ops.unfold_share()`` and holds its input alive."""
    import gc
    from dinox import ops
    wc = ops._WeightCache()
    w = torch.nn.Parameter(torch.randn(4, 8))
    sentinel = object()
    key = (id(w), False)
    import weakref
    wc.d[key] = (weakref.ref(w, lambda _r, key=key, d=wc.d: d.pop(key, None)), (w.data_ptr(), w._version, tuple(w.shape)), sentinel)
    assert wc.d[key][0]() is w
    del w
    gc.collect()
    assert key not in wc.d                     # collected with its tensor: a successor at the same id/address cannot hit it
    assert ops._unfold_share.depth == 0 and ops._unfold_share.entries == []
    x = torch.randn(2, 3, 8, 8)
    with ops.unfold_share():
        ops._unfold_share.put(x, 4, torch.float32, sentinel)
        assert ops._unfold_share.find(x, 4, torch.float32) is sentinel
        assert ops._unfold_share.find(x.clone(), 4, torch.float32) is None          # identity, not address or shape
        x.add_(1)
        assert ops._unfold_share.find(x, 4, torch.float32) is None                  # version moved
    assert ops._unfold_share.entries == []
