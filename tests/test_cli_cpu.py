"""CPU checks of the scripts/phase5_big_run.py drop-in: flag surface, config dataclasses, importable names, host data
pipeline, checkpoint payload (reference keys; optimiser state in torch.optim.AdamW format).  The flag list below is the
reference's argparse surface (scripts/phase5_big_run.py:1238-1331) written out as data."""
import json
from dataclasses import asdict, fields

import numpy as np
import pytest
import torch


REFERENCE_FLAGS = [
    "--config", "--vit-patch", "--vit-dim", "--vit-depth", "--vit-heads", "--out-dim", "--device", "--num-workers", "--pin-memory",
    "--img-size", "--batch-size", "--accumulation-steps", "--lr", "--min-lr", "--warmup-steps", "--weight-decay", "--max-steps",
    "--grad-checkpoint", "--ema", "--teacher-temp", "--student-temp", "--center-momentum", "--gram-weight", "--koleo-weight",
    "--loss-type", "--scale-aware", "--ckpt-every", "--ckpt-keep-last", "--resume", "--monitor-every", "--index-csv",
    "--split-manifest", "--crop-scale-min", "--crop-scale-max", "--z-stride", "--diverse-batches", "--train-seed", "--sdp-backend",
    "--run-dir", "--run-suffix", "--amp", "--amp-dtype", "--log-json"]
REFERENCE_DEFAULTS = {"config": "vit-large", "device": "auto", "img_size": 224, "batch_size": 64, "accumulation_steps": 1, "lr": 1e-4,
                      "min_lr": 1e-6, "warmup_steps": 2500, "weight_decay": 0.04, "ema": 0.996, "teacher_temp": 0.04,
                      "student_temp": 0.1, "center_momentum": 0.9, "gram_weight": 1.0, "koleo_weight": 0.0, "loss_type": "dino",
                      "ckpt_every": 100, "ckpt_keep_last": 5, "monitor_every": 1000, "crop_scale_min": 0.3, "crop_scale_max": 1.0,
                      "z_stride": 1, "train_seed": 0, "sdp_backend": "auto", "amp_dtype": "bfloat16", "max_steps": None, "resume": None}
TRAINING_CONFIG_FIELDS = [
    "model", "img_size", "hardware", "rw_level_min", "rw_level_max", "rw_width_min", "rw_width_max", "batch_size", "accumulation_steps",
    "lr", "min_lr", "warmup_steps", "weight_decay", "max_steps", "ema", "teacher_temp", "student_temp", "center_momentum", "loss_type",
    "gram_enabled", "gram_weight", "koleo_weight", "scale_aware", "crop_scale_min", "crop_scale_max", "z_stride", "diverse_batches",
    "ckpt_every", "ckpt_keep_last", "monitor_every", "train_seed", "sdp_backend", "amp_dtype", "index_csv", "split_manifest",
    "git_commit", "data_manifest_hash", "created_at"]


def test_flag_surface_matches_reference(cli):
    ap = cli.build_parser()
    have = {s for a in ap._actions for s in a.option_strings if s.startswith("--")} - {"--help"}
    assert set(REFERENCE_FLAGS) <= have
    assert have - set(REFERENCE_FLAGS) == {"--synthetic", "--gpu-views", "--local-crops", "--local-size", "--hip-graph", "--stack-cache",
                                               "--stack-cache-prefill", "--streams"}                                                  # the documented extensions
    d = vars(ap.parse_args([]))
    for k, v in REFERENCE_DEFAULTS.items():
        assert d[k] == v, k


def test_importable_names(cli):
    for n in ("DINOLoss", "DinoStudentTeacher", "PatchViT", "get_lr", "IndexRow", "PngDataset", "_load_index_rows", "dino_collate",
              "ModelConfig", "MODEL_CONFIGS", "TrainingConfig", "HardwareConfig", "save_checkpoint", "load_checkpoint",
              "find_latest_checkpoint", "rotate_checkpoints", "detect_anomaly", "DiverseBatchSampler", "compute_gram_anchoring_loss"):
        assert hasattr(cli, n), n
    assert cli.get_lr(0, None, 2500, 1e-4, 1e-6) == pytest.approx(4e-8)
    assert cli.DINOLoss(64).center.shape == (1, 64) and cli.DINOLoss(64).center_momentum == 0.999


def test_model_presets_and_overrides(cli):
    p = cli.MODEL_CONFIGS
    assert (p["vit-small"].patch, p["vit-small"].dim, p["vit-small"].depth, p["vit-small"].heads, p["vit-small"].out_dim) == (14, 384, 12, 6, 8192)
    assert (p["vit-tiny"].dim, p["vit-tiny"].heads, p["vit-tiny"].out_dim) == (192, 3, 4096)
    assert (p["vit-large"].dim, p["vit-large"].depth, p["vit-large"].heads) == (1024, 24, 16)
    with pytest.raises(ValueError):
        cli.ModelConfig("x", 16, 100, 2, 3)
    ap = cli.build_parser()
    cfg = cli.resolve_model_config(ap.parse_args(["--config", "vit-small", "--vit-patch", "16"]))
    assert (cfg.name, cfg.patch, cfg.dim) == ("custom", 16, 384)                       # BASELINE's ViT-S/16
    cfg = cli.resolve_model_config(ap.parse_args(["--config", "vit-small", "--vit-dim", "1024", "--vit-depth", "24", "--vit-heads", "16"]))
    assert cfg.name == "vit-large"
    with pytest.raises(ValueError):
        cli.resolve_model_config(ap.parse_args(["--config", "custom", "--vit-patch", "16"]))
    assert cli.resolve_model_config(ap.parse_args(["--config", "vit-small", "--out-dim", "1024"])).out_dim == 1024


def test_training_config_fields_match_reference_payload(cli):
    assert [f.name for f in fields(cli.TrainingConfig)] == TRAINING_CONFIG_FIELDS
    cfg = cli.TrainingConfig(model=cli.MODEL_CONFIGS["vit-tiny"], batch_size=8, accumulation_steps=4)
    assert cfg.effective_batch_size == 32 and cfg.gram_enabled is True
    d = asdict(cfg)
    assert d["model"]["name"] == "vit-tiny" and json.dumps(d)


def _write_pngs(tmp_path, n_series=3, n_slices=4, size=64):
    from PIL import Image
    rows = []
    for s in range(n_series):
        for z in range(n_slices):
            p = tmp_path / f"s{s}_z{z}.png"
            arr = (32768 + 10 * np.random.default_rng(s * 10 + z).integers(-1000, 1000, size=(size, size))).astype(np.uint16)
            Image.fromarray(arr).save(p)
            rows.append(dict(png_path=str(p), series_dir=f"series{s}", slice_index=z, encoding="hu16_png", spacing_x=0.5 + 0.1 * s,
                             spacing_y=0.5 + 0.1 * s, spacing_z=1.0 + s, dataset="toy"))
    csv_path = tmp_path / "index.csv"
    import csv as _csv
    with open(csv_path, "w", newline="") as f:
        w = _csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    return csv_path


def test_slice_cache_behind_png_dataset(cli, tmp_path):
    """--stack-cache: every PNG is decoded once into a uint16 memmap keyed by (path, size, mtime); the dataset then returns the
    same pixels as the decoder, a second process-lifetime (new SliceCache on the same directory) decodes nothing, a changed
    file gets a fresh cache, and a parallel prefill fills every entry."""
    import os
    from dinox.stackcache import SliceCache, decode_png_u16, png_shape
    from dinox.views import collate_stacks
    rows = cli._load_index_rows(_write_pngs(tmp_path, n_series=2, n_slices=3, size=48), require_spacing=True)
    root = tmp_path / "cache"
    assert png_shape(rows[0].png_path) == (48, 48)
    ds_plain = cli.PngDataset(rows, img_size=32)
    ds = cli.PngDataset(rows, img_size=32)
    ds.cache = SliceCache([r.png_path for r in rows], root)
    assert len(ds.cache) == 6 and ds.cache.filled() == 0 and ds.cache.total == 6 * 48 * 48
    for i in range(len(rows)):                                                   # first touch: decode + store
        for a, b in zip(ds._stack(rows[i]), ds_plain._stack(rows[i])):
            assert a.dtype == np.uint16 and np.array_equal(a, b)
    assert ds.cache.filled() == 6 and ds.cache.misses == 6 and ds.cache.hits == 12     # 3 slices per stack, 6 distinct files
    ds.raw_views = True                                                          # the --gpu-views item: memmap views straight into the batch buffer
    sb = collate_stacks([ds[0], ds[4]])
    back = sb.raw.numpy().view(np.uint16)
    assert np.array_equal(back[:3 * 48 * 48].reshape(3, 48, 48), np.stack(ds_plain._stack(rows[0])))
    again = SliceCache([r.png_path for r in rows], root)                         # a later run: everything is there
    assert again.key == ds.cache.key and again.filled() == 6
    assert np.array_equal(again.get(rows[3].png_path), decode_png_u16(rows[3].png_path)) and again.misses == 0
    crc = again.checksum(again.index[str(rows[3].png_path)])
    from PIL import Image                                                        # the file changes: new key, nothing stale
    Image.fromarray(np.full((48, 48), 40000, dtype=np.uint16)).save(rows[3].png_path)
    os.utime(rows[3].png_path, ns=(1, 1))
    fresh = SliceCache([r.png_path for r in rows], root)
    assert fresh.key != again.key and fresh.filled() == 0
    assert fresh.prefill(workers=2) == 6 and fresh.filled() == 6 and fresh.prefill(workers=2) == 0
    assert int(fresh.get(rows[3].png_path)[5, 7]) == 40000 and fresh.checksum(fresh.index[str(rows[3].png_path)]) != crc
    with pytest.raises(KeyError):
        fresh.get(tmp_path / "not_in_the_index.png")
    with pytest.raises(FileNotFoundError):
        SliceCache([rows[0].png_path], tmp_path / "other", create=False)


def test_slice_cache_file_kinds_and_ragged_shapes(tmp_path):
    """The cache holds what the dataset's decoder returns, whatever the file is: 16-bit grey as stored, the first channel of an RGB
    file, the VALUES of an 8-bit file (as np.asarray(.., uint16) would give) -- and slices of different sizes side by side (one entry
    per file at its own H x W)."""
    from PIL import Image
    from dinox.stackcache import SliceCache, decode_png_u16, png_shape
    g = np.random.default_rng(0)
    files = []
    a16 = g.integers(0, 65535, (40, 56), dtype=np.uint16)
    Image.fromarray(a16).save(tmp_path / "g16.png"); files.append(tmp_path / "g16.png")
    rgb = g.integers(0, 255, (32, 32, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "rgb.png"); files.append(tmp_path / "rgb.png")
    a8 = g.integers(0, 255, (24, 72), dtype=np.uint8)
    Image.fromarray(a8).save(tmp_path / "g8.png"); files.append(tmp_path / "g8.png")
    assert [png_shape(f) for f in files] == [(40, 56), (32, 32), (24, 72)]
    cache = SliceCache(files, tmp_path / "c")
    assert cache.total == 40 * 56 + 32 * 32 + 24 * 72 and sorted(cache.shapes) == sorted([(40, 56), (32, 32), (24, 72)])
    for f, want in zip(files, (a16, rgb[:, :, 0], a8)):
        got = cache.get(f)
        assert got.dtype == np.uint16 and got.shape == want.shape and np.array_equal(got, want.astype(np.uint16))
        assert np.array_equal(np.asarray(decode_png_u16(f)).astype(np.uint16), got)
    assert cache.filled() == 3 and np.array_equal(cache.get(files[0]), a16) and cache.hits == 1
    (tmp_path / "bad.png").write_bytes(b"not a png at all, just bytes......")
    with pytest.raises(ValueError, match="not a PNG"):
        SliceCache([tmp_path / "bad.png"], tmp_path / "c2")


def test_png_dataset_views_and_collate(cli, tmp_path):
    rows = cli._load_index_rows(_write_pngs(tmp_path), require_spacing=True)
    assert len(rows) == 12 and rows[5].spacing_z == 2.0 and rows[0].dataset == "toy"
    ds = cli.PngDataset(rows, img_size=32, scale_aware=True)
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)          # the augmentation draws windows and crops at random
    (v1, v2), sp = ds[5]
    assert v1.shape == v2.shape == (3, 32, 32) and v1.dtype == torch.float32 and sp.tolist() == pytest.approx([0.6, 0.6, 2.0])
    assert not torch.equal(v1, v2)                                   # independent window + crop per view
    lo, hi = (0 - 0.485) / 0.229 - 0.6, (1 - 0.406) / 0.225 + 0.6    # normalised [0,1] range (+ bicubic overshoot)
    assert lo <= float(v1.min()) and float(v1.max()) <= hi
    views, spb = cli.dino_collate([ds[i] for i in range(4)])
    assert views[0].shape == views[1].shape == (4, 3, 32, 32) and spb.shape == (4, 3)
    syn = cli.SyntheticSliceDataset(10, img_size=32, seed=3)
    (a, b), s = syn[2]
    (a2, _), _ = cli.SyntheticSliceDataset(10, img_size=32, seed=3)[2]
    assert a.shape == (3, 32, 32) and 0.46 <= float(s[0]) <= 0.98 and 0.625 <= float(s[2]) <= 5.0


def test_hu_window(cli):
    u = np.array([[32768 + 400, 32768 - 10000, 65535]], dtype=np.uint16)       # 40 HU, -1000 HU, +3276.7 HU
    w = cli.hu_window01(u, level=40.0, width=400.0)
    np.testing.assert_allclose(w, [[0.5, 0.0, 1.0]], atol=1e-6)


def test_diverse_batch_sampler(cli):
    rows = [cli.IndexRow(png_path=f"{s}_{z}", series_dir=f"s{s}", slice_index=z, encoding="hu16_png") for s in range(6) for z in range(5)]
    smp = cli.DiverseBatchSampler(rows, batch_size=4, drop_last=True, generator=torch.Generator().manual_seed(0))
    batches = list(smp)
    assert len(batches) == len(smp) == 7 and sorted(i for b in batches for i in b) != []
    assert len({i for b in batches for i in b}) == 28                 # no repeats
    for b in batches[:6]:                                             # while >= 4 series remain a batch never repeats a series
        assert len({rows[i].series_dir for i in b}) == 4


def test_detect_anomaly_and_rotation(cli, tmp_path):
    assert cli.detect_anomaly(float("nan"), [])[0] and "NaN" in cli.detect_anomaly(float("nan"), [])[1]
    assert cli.detect_anomaly(float("inf"), [])[1] == "Loss is Inf"
    assert cli.detect_anomaly(5.0, [1.0] * 10)[0] and not cli.detect_anomaly(1.5, [1.0] * 10)[0]
    assert cli.detect_anomaly(1.0, [], embedding_std=0.001)[0]
    for i in (1, 2, 3, 4):
        (tmp_path / f"checkpoint_{i:08d}.pth").write_bytes(b"x")
    cli.rotate_checkpoints(tmp_path, 2)
    assert sorted(p.name for p in tmp_path.glob("checkpoint_*.pth")) == ["checkpoint_00000003.pth", "checkpoint_00000004.pth"]
    assert cli.find_latest_checkpoint(tmp_path).name == "checkpoint_00000004.pth"


def test_checkpoint_payload_roundtrip_and_adamw_format(cli, tmp_path):
    from dinox.engine import StepHyperParams, TrainEngine
    import zoo.arch as arch
    kw = dict(img_size=28, patch=14, dim=32, depth=1, heads=2, scale_aware=True)
    torch.manual_seed(0)
    student, teacher = arch.DinoStudentTeacher(arch.PatchViT(**kw), 48), arch.DinoStudentTeacher(arch.PatchViT(**kw), 48)
    teacher.load_state_dict(student.state_dict())
    eng = TrainEngine(student, teacher, 48, StepHyperParams())
    eng.adam_m.normal_(); eng.adam_v.uniform_(0, 1); eng.center.normal_()
    eng.opt_steps, eng.step_count = 7, 7
    cfg = cli.TrainingConfig(model=cli.ModelConfig("custom", 14, 32, 1, 2, 4.0, 48), img_size=28, scale_aware=True)
    path = tmp_path / "checkpoint_00000007.pth"
    cli.save_checkpoint(path, 7, student, teacher, eng, cfg)
    payload = torch.load(path, weights_only=False)
    assert set(payload) == {"step", "student", "teacher", "opt", "scaler", "dino_loss", "rng", "config"}
    assert payload["scaler"] is None and set(payload["rng"]) >= {"python", "numpy", "torch"} and payload["dino_loss"]["center"].shape == (1, 48)
    assert list(payload["student"]) == list(student.state_dict()) and payload["config"]["model"]["dim"] == 32
    # "opt" must be loadable by a stock torch.optim.AdamW over the same parameters (what the reference resumes with)
    ref_opt = torch.optim.AdamW(student.parameters(), lr=1e-4, weight_decay=0.04)
    ref_opt.load_state_dict(payload["opt"])
    st = ref_opt.state[next(iter(student.parameters()))]
    assert float(st["step"]) == 7.0 and st["exp_avg"].shape == next(iter(student.parameters())).shape
    # and back into a fresh engine
    s2, t2 = arch.DinoStudentTeacher(arch.PatchViT(**kw), 48), arch.DinoStudentTeacher(arch.PatchViT(**kw), 48)
    eng2 = TrainEngine(s2, t2, 48, StepHyperParams())
    step, loaded = cli.load_checkpoint(path, s2, t2, eng2, "cpu", scale_aware=True)
    assert step == 7 and eng2.opt_steps == 7 and eng2.step_count == 7 and loaded.model.dim == 32 and loaded.scale_aware
    assert torch.equal(eng2.flat_p, eng.flat_p) and torch.equal(eng2.adam_m, eng.adam_m) and torch.equal(eng2.adam_v, eng.adam_v)
    assert torch.equal(eng2.center, eng.center)
    with pytest.raises(FileNotFoundError):
        cli.load_checkpoint(tmp_path / "missing.pth", s2, t2, eng2, "cpu")


def test_sharded_batch_sampler_partitions_global_batches(cli):
    """Data parallel: every rank walks the SAME global batch sequence (shared seed) and keeps a disjoint contiguous slice; the
    union over ranks is what one process would draw at the global batch, so an epoch sees no sample twice."""
    n, B, world = 64, 4, 4
    per_rank = []
    for rank in range(world):
        gen = torch.Generator().manual_seed(7)
        inner = torch.utils.data.BatchSampler(torch.utils.data.RandomSampler(range(n), generator=gen), batch_size=B * world, drop_last=True)
        per_rank.append(list(cli.ShardedBatchSampler(inner, rank, world)))
    gen = torch.Generator().manual_seed(7)
    single = list(torch.utils.data.BatchSampler(torch.utils.data.RandomSampler(range(n), generator=gen), batch_size=B * world, drop_last=True))
    assert all(len(p) == len(single) == n // (B * world) for p in per_rank)
    for i, g in enumerate(single):
        assert sum((per_rank[r][i] for r in range(world)), []) == g
    seen = [j for p in per_rank for b in p for j in b]
    assert len(seen) == len(set(seen)) == n
    # the series-diverse sampler shards the same way
    rows = [cli.IndexRow(png_path=f"s{i % 8}/{i}.png", series_dir=f"s{i % 8}", slice_index=i // 8, encoding="u16") for i in range(n)]
    shards = []
    for rank in range(2):
        inner = cli.DiverseBatchSampler(rows, batch_size=8, drop_last=True, generator=torch.Generator().manual_seed(3))
        shards.append(list(cli.ShardedBatchSampler(inner, rank, 2)))
    flat = [j for p in shards for b in p for j in b]
    assert len(flat) == len(set(flat)) == n and all(len(b) == 4 for p in shards for b in p)


def test_collate_stacks_shared_ring_and_slice_form(cli):
    """--gpu-views host path: `collate_stacks` takes a stack as a (3,H,W) array or as its three (H,W) slices and writes them into one flat
    buffer; inside a DataLoader worker that buffer is a shared-memory tensor from a per-worker ring (views.SHM_RING) -- reused storage, same
    bytes.  The synthetic dataset's stacks are a function of the index, so every batch can be checked against a direct rebuild."""
    import dinox.views as V
    ds = cli.SyntheticSliceDataset(24, img_size=32)
    ds.raw_views, ds.local_crops = True, 0
    items = [ds[i] for i in range(6)]
    assert isinstance(items[0][0], list) and len(items[0][0]) == 3 and items[0][0][0].dtype == np.uint16
    want = np.concatenate([np.stack(s, 0).reshape(-1) for s, _, _ in items]).view(np.int16)
    a = V.collate_stacks(items)
    b = V.collate_stacks([(np.stack(s, 0), v, sp) for s, v, sp in items])
    assert np.array_equal(a.raw.numpy(), want) and np.array_equal(b.raw.numpy(), want)
    assert a.offsets == b.offsets == [i * 3 * 40 * 40 for i in range(6)] and a.shapes == [(40, 40)] * 6 and len(a.views) == 2
    assert a.pin_memory is not None and a.spacing.shape == (6, 3)
    old = V.SHM_RING
    V.SHM_RING = 4          # > prefetch_factor (2) + 1: a worker is never more than two batches ahead of the one being read
    try:
        sampler = torch.utils.data.BatchSampler(torch.utils.data.SequentialSampler(ds), batch_size=4, drop_last=True)
        dl = torch.utils.data.DataLoader(ds, batch_sampler=sampler, num_workers=1, collate_fn=V.collate_stacks, persistent_workers=True)
        seen = []
        for epoch in range(2):
            for k, batch in enumerate(dl):
                assert batch.raw.is_shared()
                ref = np.concatenate([np.stack(ds._make(4 * k + j), 0).reshape(-1) for j in range(4)]).view(np.int16)
                assert np.array_equal(batch.raw.numpy(), ref), (epoch, k)
                seen.append(batch.raw.untyped_storage().data_ptr() if hasattr(batch.raw, "untyped_storage") else 0)
        assert len(seen) == 12
    finally:
        V.SHM_RING = old
