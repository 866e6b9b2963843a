#!/usr/bin/env python3
"""Phase-5 "big run" training CLI on the MI355X engine -- drop-in for the reference script of the same name.

Same flags, config dataclasses, printed keys (``model_config= ... device= ... run_dir= ... step= ...
checkpoint_saved= ... final_checkpoint=``), ``--log-json`` schema ``{step, loss, lr}``, checkpoint naming,
rotation and payload keys ``{step, student, teacher, opt, scaler, dino_loss, rng, config}`` as
``scripts/phase5_big_run.py`` of timlawrenz/DINO-X (flags :1238-1331, config :153-306, checkpoint :1104-1207,
loop :1686-1997), and the same importable names other reference scripts/tests use (``DINOLoss``,
``DinoStudentTeacher``, ``PatchViT``, ``get_lr``, ``IndexRow``, ``PngDataset``, ``_load_index_rows``, ``dino_collate``,
``ModelConfig``, ``MODEL_CONFIGS``).  What differs:

* the optimiser step runs on ``dinox.engine.TrainEngine`` (HIP kernels, flat arenas, no per-step host sync: the
  loss is fetched only when it is logged); ``opt`` in the checkpoint is written in ``torch.optim.AdamW``
  ``state_dict`` format so reference checkpoints resume here and vice versa;
* one process per GPU under ``torch.distributed.run`` gives data parallelism (each rank its own data shard);
* the input pipeline is torchvision-free (PIL + torch CPU ops) and there is an explicit ``--synthetic N`` source of
  seeded 16-bit HU slice stacks for runs without a dataset (this environment has none); ``--gpu-views`` moves everything
  after the PNG decode (window, antialiased bicubic RandomResizedCrop, flip, normalise) into one HIP kernel (dinox/views.py);
* ``--koleo-weight`` searches nearest neighbours over the GLOBAL batch under data parallelism (dinox.ops.KoLeoFn);
* not wired to the engine yet (exit with a message): ``--loss-type simclr|mae``, ``--device cpu`` (there is no CPU
  compute path).
"""
from __future__ import annotations

import argparse
import csv
import hashlib
import json
import math
import os
import random
import signal
import subprocess
import sys
import time
import warnings
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from dinox import ops  # noqa: E402
from dinox.dp import init_process_group  # noqa: E402
from dinox.engine import StepHyperParams, TrainEngine  # noqa: E402
from dinox.hostinfo import usable_cpus  # noqa: E402
from dinox.schedule import get_lr  # noqa: E402,F401
from zoo.arch import DinoStudentTeacher, PatchViT, ScaleEmbedding, TransformerBlock, migrate_state_dict, needs_migration  # noqa: E402,F401


# ------------------------------------------------------------------------------------------ configuration
@dataclass
class ModelConfig:
    name: str
    patch: int
    dim: int
    depth: int
    heads: int
    mlp_ratio: float = 4.0
    out_dim: int = 8192

    def __post_init__(self):
        if self.dim % self.heads != 0:
            raise ValueError(f"dim ({self.dim}) must be divisible by heads ({self.heads})")
        if self.patch not in [8, 14, 16]:
            warnings.warn(f"Unusual patch size: {self.patch}")

    @property
    def params_millions(self) -> float:
        blocks = self.depth * (4 * self.dim ** 2 + 8 * self.dim ** 2 * self.mlp_ratio)
        return (3 * self.patch ** 2 * self.dim + blocks + 2 * self.dim * self.out_dim) / 1e6


def _presets() -> dict:
    return {
        "vit-tiny": ModelConfig("vit-tiny", 14, 192, 12, 3, 4.0, 4096),
        "vit-small": ModelConfig("vit-small", 14, 384, 12, 6, 4.0, 8192),
        "vit-large": ModelConfig("vit-large", 14, 1024, 24, 16, 4.0, 8192),
        "vit-giant": ModelConfig("vit-giant", 14, 1408, 40, 16, 4.0, 8192),
    }


MODEL_CONFIGS = _presets()


@dataclass
class HardwareConfig:
    device_type: str
    device_name: str
    is_rocm: bool
    num_workers: int
    pin_memory: bool
    batch_size_recommendation: int


@dataclass
class TrainingConfig:
    model: ModelConfig
    img_size: int = 224
    hardware: Optional[HardwareConfig] = None
    rw_level_min: float = -400.0
    rw_level_max: float = 400.0
    rw_width_min: float = 800.0
    rw_width_max: float = 2000.0
    batch_size: int = 64
    accumulation_steps: int = 1
    lr: float = 1e-4
    min_lr: float = 1e-6
    warmup_steps: int = 2500
    weight_decay: float = 0.04
    max_steps: Optional[int] = None
    ema: float = 0.996
    teacher_temp: float = 0.04
    student_temp: float = 0.1
    center_momentum: float = 0.9
    loss_type: str = "dino"
    gram_enabled: bool = True
    gram_weight: float = 1.0
    koleo_weight: float = 0.0
    scale_aware: bool = False
    crop_scale_min: float = 0.3
    crop_scale_max: float = 1.0
    z_stride: int = 1
    diverse_batches: bool = False
    ckpt_every: int = 100
    ckpt_keep_last: int = 5
    monitor_every: int = 1000
    train_seed: int = 0
    sdp_backend: str = "auto"
    amp_dtype: str = "bfloat16"
    index_csv: str = "data/processed/_index/index.csv"
    split_manifest: Optional[str] = None
    git_commit: Optional[str] = None
    data_manifest_hash: Optional[str] = None
    created_at: str = field(default_factory=lambda: time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()))

    @property
    def effective_batch_size(self) -> int:
        return self.batch_size * self.accumulation_steps


def detect_hardware() -> HardwareConfig:
    if not torch.cuda.is_available():
        return HardwareConfig("cpu", "CPU", False, 4, False, 8)
    name = torch.cuda.get_device_name(0)
    is_rocm = getattr(torch.version, "hip", None) is not None
    if "MI355" in name or "MI350" in name or "gfx950" in name:
        # 288 GB HBM3E: the whole bs-256 step with every activation saved is ~25 GB; host side feeds with up to 16 workers
        return HardwareConfig("cuda", name, True, min(16, usable_cpus()), True, 256)
    return HardwareConfig("cuda", name, is_rocm, min(4, usable_cpus()), True, 32)


def get_git_commit() -> Optional[str]:
    try:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, check=True, timeout=5).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain"], capture_output=True, text=True, check=True, timeout=5).stdout.strip()
        return commit + ("-dirty" if dirty else "")
    except Exception:
        return None


def compute_data_manifest_hash(index_csv: Path) -> Optional[str]:
    try:
        return hashlib.sha256(Path(index_csv).read_bytes()).hexdigest()[:16] if Path(index_csv).exists() else None
    except Exception:
        return None


# ------------------------------------------------------------------------------------------ data (host side)
@dataclass
class IndexRow:
    png_path: Path
    series_dir: str
    slice_index: int
    encoding: str
    spacing_x: float = 1.0
    spacing_y: float = 1.0
    spacing_z: float = 1.0
    dataset: str = ""


def _load_index_rows(index_csv: Path, require_spacing: bool = False) -> list:
    """CSV columns: png_path, series_dir, slice_index, encoding [, spacing_x, spacing_y, spacing_z] [, dataset]."""
    rows = []
    with open(index_csv, newline="") as f:
        reader = csv.DictReader(f)
        cols = reader.fieldnames or []
        has_spacing = all(c in cols for c in ("spacing_x", "spacing_y", "spacing_z"))
        if require_spacing and not has_spacing:
            warnings.warn(f"--scale-aware is enabled but {index_csv} lacks spacing_x/spacing_y/spacing_z columns. "
                          "Defaulting to (1.0, 1.0, 1.0) — the model won't learn real scale awareness.")
        for r in reader:
            row = IndexRow(Path(r["png_path"]), r["series_dir"], int(r["slice_index"]), r["encoding"])
            if has_spacing:
                row.spacing_x, row.spacing_y, row.spacing_z = float(r["spacing_x"]), float(r["spacing_y"]), float(r["spacing_z"])
            if "dataset" in cols:
                row.dataset = r["dataset"]
            rows.append(row)
    return rows


_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
_STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


def random_resized_crop_flip_normalize(x: torch.Tensor, size: int, scale=(0.3, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)) -> torch.Tensor:
    """(3,H,W) in [0,1] -> (3,size,size): RandomResizedCrop (bicubic) + horizontal flip (p=.5) + ImageNet normalise,
    the augmentation of the reference pipeline (scripts/phase5_big_run.py:493-497), drawn from Python's ``random``
    (dinox.views.draw_crop_box: the same draws feed the device-side pipeline of ``--gpu-views``)."""
    from dinox.views import draw_crop_box
    _, H, W = x.shape
    top, left, h, w = draw_crop_box(H, W, scale, ratio)
    crop = x[:, top:top + h, left:left + w].unsqueeze(0)
    out = F.interpolate(crop, size=(size, size), mode="bicubic", align_corners=False, antialias=True)[0]
    if random.random() < 0.5:
        out = out.flip(-1)
    return (out - _MEAN) / _STD


def hu_window01(u16: np.ndarray, level: float, width: float) -> np.ndarray:
    """stored = HU*10 + 32768  ->  HU  ->  clip((HU - (level - width/2)) / max(width, 1), 0, 1)."""
    hu = (u16.astype(np.float32) - 32768.0) * 0.1
    return np.clip((hu - (level - width / 2.0)) / max(width, 1.0), 0.0, 1.0)


class PngDataset(torch.utils.data.Dataset):
    """16-bit HU PNG slices with (z-1, z, z+1) context and two independently windowed/augmented views."""

    def __init__(self, rows, img_size=224, rw_level_min=-400.0, rw_level_max=400.0, rw_width_min=800.0, rw_width_max=2000.0,
                 scale_aware=False, crop_scale_min=0.3, crop_scale_max=1.0):
        self.rows, self.img_size, self.scale_aware = rows, img_size, scale_aware
        self.rw_level_min, self.rw_level_max, self.rw_width_min, self.rw_width_max = rw_level_min, rw_level_max, rw_width_min, rw_width_max
        self.crop_scale = (crop_scale_min, crop_scale_max)
        self.raw_views = False          # True: __getitem__ returns (u16 stack, view draws, spacing) for the device-side pipeline
        self.local_crops, self.local_scale = 0, (0.05, 0.3)      # multi-crop extension: extra student-only views (DINO's local-crop scale)
        self.cache = None               # dinox.stackcache.SliceCache (--stack-cache): every PNG is decoded once, then read from a u16 memmap
        self._series_map: dict = {}
        for r in rows:
            self._series_map.setdefault(r.series_dir, {})[r.slice_index] = r.png_path
        self._series_minmax = {s: (min(m), max(m)) for s, m in self._series_map.items() if m}

    def __len__(self):
        return len(self.rows)

    def _read(self, p) -> np.ndarray:
        if self.cache is not None:
            return self.cache.get(p)
        from dinox.stackcache import decode_png_u16
        return decode_png_u16(p)

    def _stack(self, row: IndexRow) -> list:
        z0, z1 = self._series_minmax.get(row.series_dir, (row.slice_index, row.slice_index))
        mp = self._series_map.get(row.series_dir, {})
        return [self._read(mp.get(max(z0, min(z1, row.slice_index + dz)), row.png_path)) for dz in (-1, 0, 1)]

    def _view(self, slices: list) -> torch.Tensor:
        level = random.uniform(self.rw_level_min, self.rw_level_max)
        width = random.uniform(self.rw_width_min, self.rw_width_max)
        x = torch.from_numpy(np.stack([hu_window01(s, level, width) for s in slices], 0)).contiguous()
        return random_resized_crop_flip_normalize(x, self.img_size, scale=self.crop_scale)

    def __getitem__(self, idx: int):
        for _ in range(10):
            try:
                row = self.rows[idx]
                slices = self._stack(row)
                spacing = torch.tensor([row.spacing_x, row.spacing_y, row.spacing_z], dtype=torch.float32)
                if self.raw_views:      # --gpu-views: ship the u16 stack and the draws of both views; dinox_slice_views does the rest
                    from dinox.views import draw_view
                    stack = [np.asarray(sl, dtype=np.uint16) for sl in slices]          # collate_stacks writes the slices straight into the batch
                    H, W = stack[0].shape
                    kw = dict(rw_level=(self.rw_level_min, self.rw_level_max), rw_width=(self.rw_width_min, self.rw_width_max),
                              crop_scale=self.crop_scale)
                    views = [draw_view(H, W, **kw), draw_view(H, W, **kw)]
                    views += [draw_view(H, W, **dict(kw, crop_scale=self.local_scale)) for _ in range(self.local_crops)]
                    return stack, views, spacing
                return [self._view(slices), self._view(slices)], spacing
            except Exception as e:
                print(f"⚠️  Data loading error at index {idx} ({self.rows[idx].png_path}): {e}")
                idx = random.randint(0, len(self.rows) - 1)
        raise RuntimeError("Failed to load data after 10 attempts")


class SyntheticSliceDataset(PngDataset):
    """``--synthetic N``: N seeded 3-slice stacks of 16-bit HU data (stored range [-1000, 4000] HU) with random
    physical spacing; goes through the same windowing/augmentation as PngDataset.  Not in the reference."""

    def __init__(self, n: int, img_size: int, raw_size: int = 0, seed: int = 0, **kw):
        rows = [IndexRow(Path(f"synthetic/{i:06d}.png"), f"series{i // 64:04d}", i % 64, "hu16_png") for i in range(n)]
        super().__init__(rows, img_size=img_size, **kw)
        self.raw = raw_size or int(img_size * 1.25)
        self.seed = seed
        self._cache: dict = {}
        g = np.random.default_rng(seed)
        for r in rows:
            r.spacing_x = r.spacing_y = float(g.uniform(0.46, 0.98))
            r.spacing_z = float(g.uniform(0.625, 5.0))

    def _stack(self, row: IndexRow) -> list:
        i = int(row.png_path.stem)
        hit = self._cache.get(i)             # (per worker process) a stack is a pure function of its index: generate it once --
        if hit is not None:                  # drawing 3 x raw^2 normals per item costs more host time than the step costs GPU time
            return hit
        self._cache[i] = out = self._make(i)
        return out

    def _make(self, i: int) -> list:
        g = np.random.default_rng(self.seed * 1_000_003 + i)
        base = g.integers(22768, 72768, size=(self.raw // 8 + 1, self.raw // 8 + 1)).astype(np.float32)
        img = np.kron(base, np.ones((8, 8), dtype=np.float32))[: self.raw, : self.raw]      # blocky "anatomy"
        return [np.clip(img + g.normal(0, 300, img.shape), 0, 65535).astype(np.uint16) for _ in range(3)]


class DiverseBatchSampler(torch.utils.data.Sampler):
    """Round-robin over series so a batch holds at most one slice per series (while series last)."""

    def __init__(self, rows, batch_size: int, drop_last: bool = True, generator: Optional[torch.Generator] = None):
        self.batch_size, self.drop_last, self.generator = batch_size, drop_last, generator
        self._series: dict = {}
        for i, r in enumerate(rows):
            self._series.setdefault(r.series_dir, []).append(i)
        self._total = len(rows)

    def __len__(self):
        return self._total // self.batch_size if self.drop_last else -(-self._total // self.batch_size)

    def __iter__(self):
        queues = []
        for idxs in self._series.values():
            queues.append([idxs[i] for i in torch.randperm(len(idxs), generator=self.generator).tolist()])
        queues = [queues[i] for i in torch.randperm(len(queues), generator=self.generator).tolist()]
        order = []
        while queues:
            for q in queues:
                order.append(q.pop())
            queues = [q for q in queues if q]
        full = len(order) // self.batch_size * self.batch_size
        for i in range(0, full, self.batch_size):
            yield order[i:i + self.batch_size]
        if order[full:] and not self.drop_last:
            yield order[full:]


class ShardedBatchSampler(torch.utils.data.Sampler):
    """Data-parallel sharding of a batch sampler: ``inner`` yields GLOBAL batches (world x per-rank batch indices) from a
    generator seeded identically on every rank, and rank r keeps the contiguous slice dp.shard_range gives it.  The union over
    ranks is exactly the batch sequence a single process would draw at the global batch size: no sample is seen twice in an
    epoch, and an epoch has the same length on every rank (so collectives stay matched)."""

    def __init__(self, inner, rank: int, world: int):
        self.inner, self.rank, self.world = inner, rank, world

    def __len__(self):
        return len(self.inner)

    def __iter__(self):
        from dinox.dp import shard_range
        for global_batch in self.inner:
            lo, hi = shard_range(len(global_batch), self.rank, self.world)
            yield global_batch[lo:hi]


def dino_collate(batch):
    views, spacings = zip(*batch)
    return [torch.stack([v[0] for v in views]), torch.stack([v[1] for v in views])], torch.stack(list(spacings))


# ------------------------------------------------------------------------------------------ loss module (importable name)
class DINOLoss(nn.Module):
    """DINO centring/sharpening CE on the HIP kernels; ``center`` buffer (1, out_dim) as in the reference."""

    def __init__(self, out_dim: int, center_momentum: float = 0.999) -> None:
        super().__init__()
        self.center_momentum = center_momentum
        self.register_buffer("center", torch.zeros(1, out_dim))

    @torch.no_grad()
    def update_center(self, teacher_output: torch.Tensor) -> None:
        ops.center_ema_(self.center.view(-1), ops.colmean(teacher_output), self.center_momentum)

    def forward(self, student_out, teacher_out, student_temp: float, teacher_temp: float) -> torch.Tensor:
        loss = ops.DinoCEFn.apply(student_out, teacher_out, self.center, student_temp, teacher_temp)
        self.update_center(teacher_out)
        return loss


def compute_gram_anchoring_loss(student_feats: torch.Tensor, teacher_feats: torch.Tensor) -> torch.Tensor:
    return ops.GramLossFn.apply(student_feats, teacher_feats)


# ------------------------------------------------------------------------------------------ rng / checkpoint / anomaly
class _StopFlag:
    def __init__(self):
        self.stop = False


def _seed_all(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _get_rng_state() -> dict:
    st = {"python": random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state().cpu()}
    if torch.cuda.is_available():
        st["cuda"] = [s.cpu() for s in torch.cuda.get_rng_state_all()]
    return st


def _set_rng_state(state: dict) -> None:
    random.setstate(state["python"])
    np.random.set_state(state["numpy"])
    torch.set_rng_state(state["torch"].cpu() if isinstance(state["torch"], torch.Tensor) else state["torch"])
    if torch.cuda.is_available() and "cuda" in state:
        cs = state["cuda"]
        cs = [s.cpu() if isinstance(s, torch.Tensor) else s for s in cs] if isinstance(cs, list) else cs
        if len(cs) == torch.cuda.device_count():
            torch.cuda.set_rng_state_all(cs)


def adamw_state_dict(eng: TrainEngine, lr: float) -> dict:
    """The engine's flat Adam moments in ``torch.optim.AdamW.state_dict()`` format (what the reference stores)."""
    template = torch.optim.AdamW(eng.params, lr=lr, betas=(eng.hp.beta1, eng.hp.beta2), eps=eng.hp.adam_eps,
                                 weight_decay=eng.hp.weight_decay).state_dict()
    state = {}
    if eng.opt_steps > 0:
        for i, (p, off) in enumerate(zip(eng.params, eng.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(eng.opt_steps)),
                        "exp_avg": eng.adam_m[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": eng.adam_v[off:off + n].view(p.shape).clone()}
    template["state"] = state
    return template


def load_adamw_state_dict(eng: TrainEngine, sd: dict) -> None:
    state = sd.get("state", {})
    steps = 0
    for i, (p, off) in enumerate(zip(eng.params, eng.offsets)):
        st = state.get(i, state.get(str(i)))
        if st is None:
            continue
        n = p.numel()
        eng.adam_m[off:off + n].copy_(st["exp_avg"].reshape(-1))
        eng.adam_v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        steps = max(steps, int(float(st["step"])))
    eng.opt_steps = steps


def save_checkpoint(path: Path, step: int, student: nn.Module, teacher: nn.Module, eng: TrainEngine, config: TrainingConfig) -> None:
    payload = {
        "step": step,
        "student": {k: v.detach().cpu().clone() for k, v in student.state_dict().items()},
        "teacher": {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()},
        "opt": adamw_state_dict(eng, config.lr),
        "scaler": None,                                   # bf16 needs no GradScaler (reference: None unless fp16)
        "dino_loss": {"center": eng.center.detach().cpu().clone()},
        "rng": _get_rng_state(),
        "config": asdict(config),
    }
    torch.save(payload, path)


def load_checkpoint(path: Path, student: nn.Module, teacher: nn.Module, eng: TrainEngine, device, scale_aware: bool = False):
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    from zoo.hub import read_checkpoint
    payload = read_checkpoint(path, "cpu")          # restricted unpickler + allow-list for the NumPy RNG blob (zoo/hub.py)
    for key in ("student", "teacher"):
        if key in payload and needs_migration(payload[key]):
            warnings.warn(f"Migrating old-format {key} state dict keys to timm-style")
            payload[key] = migrate_state_dict(payload[key])
    cfg_d = payload.get("config", {})
    strict = bool(cfg_d.get("scale_aware", False)) == bool(scale_aware)
    if not strict:
        warnings.warn(f"Scale-aware mismatch: checkpoint={cfg_d.get('scale_aware', False)}, current={scale_aware}. "
                      "Loading with strict=False (scale_embed weights will be freshly initialized).")
    student.load_state_dict(payload["student"], strict=strict)            # in-place copies keep the flat-arena views
    teacher.load_state_dict(payload["teacher"], strict=strict)
    if payload.get("opt") is not None and strict:
        load_adamw_state_dict(eng, payload["opt"])
    if payload.get("dino_loss") is not None:
        eng.center.copy_(payload["dino_loss"]["center"].to(eng.center.device))
    if payload.get("rng") is not None:
        _set_rng_state(payload["rng"])
    step = int(payload.get("step", 0))
    eng.step_count = step
    ops.weight_cache.clear()
    model_cfg = ModelConfig(**cfg_d.get("model", {})) if cfg_d.get("model") else None
    hw = HardwareConfig(**cfg_d["hardware"]) if cfg_d.get("hardware") else None
    known = {f for f in TrainingConfig.__dataclass_fields__}
    config = TrainingConfig(model=model_cfg, hardware=hw, **{k: v for k, v in cfg_d.items() if k in known and k not in ("model", "hardware")}) \
        if model_cfg else None
    return step, config


def find_latest_checkpoint(run_dir: Path) -> Optional[Path]:
    ckpts = sorted(Path(run_dir).glob("checkpoint_*.pth"))
    return ckpts[-1] if ckpts else None


def rotate_checkpoints(run_dir: Path, keep_last: int) -> None:
    ckpts = sorted(Path(run_dir).glob("checkpoint_*.pth"))
    for c in ckpts[:-keep_last] if len(ckpts) > keep_last else []:
        c.unlink()


def detect_anomaly(loss: float, loss_history: list, embedding_std: Optional[float] = None):
    if not np.isfinite(loss):
        return True, f"Loss is {'NaN' if np.isnan(loss) else 'Inf'}"
    if len(loss_history) >= 10:
        recent = float(np.mean(loss_history[-10:]))
        if loss > recent * 2.0:
            return True, f"Loss spike detected: {loss:.4f} > 2x recent mean {recent:.4f}"
    if embedding_std is not None and embedding_std < 0.01:
        return True, f"Feature collapse detected: embedding std={embedding_std:.6f}"
    return False, None


# ------------------------------------------------------------------------------------------ CLI
def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="Phase 5: Big Run - Production DINOv3 training (MI355X engine)")
    ap.add_argument("--config", choices=list(MODEL_CONFIGS.keys()) + ["custom"], default="vit-large", help="Model configuration preset")
    ap.add_argument("--vit-patch", type=int, help="Custom: patch size")
    ap.add_argument("--vit-dim", type=int, help="Custom: model dimension")
    ap.add_argument("--vit-depth", type=int, help="Custom: number of transformer blocks")
    ap.add_argument("--vit-heads", type=int, help="Custom: number of attention heads")
    ap.add_argument("--out-dim", type=int, help="Override output dimension (default: 8192)")
    ap.add_argument("--device", choices=["auto", "cuda", "cpu"], default="auto", help="Hardware target (auto-detect or override)")
    ap.add_argument("--num-workers", type=int, help="Override num_workers")
    ap.add_argument("--pin-memory", type=bool, help="Override pin_memory")
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--accumulation-steps", type=int, default=1)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--min-lr", type=float, default=1e-6)
    ap.add_argument("--warmup-steps", type=int, default=2500)
    ap.add_argument("--weight-decay", type=float, default=0.04)
    ap.add_argument("--max-steps", type=int, help="Maximum training steps (None = unlimited)")
    ap.add_argument("--grad-checkpoint", action="store_true", help="Enable gradient checkpointing (saves memory)")
    ap.add_argument("--ema", type=float, default=0.996)
    ap.add_argument("--teacher-temp", type=float, default=0.04)
    ap.add_argument("--student-temp", type=float, default=0.1)
    ap.add_argument("--center-momentum", type=float, default=0.9, help="Momentum for DINO centering (default: 0.9)")
    ap.add_argument("--gram-weight", type=float, default=1.0, help="Gram Anchoring weight (default: 1.0)")
    ap.add_argument("--koleo-weight", type=float, default=0.0, help="KoLeo regularization weight (default: 0.0)")
    ap.add_argument("--loss-type", choices=["dino", "simclr", "mae"], default="dino", help="Objective function")
    ap.add_argument("--scale-aware", action="store_true", help="Enable scale embedding (pixel spacing x/y + slice thickness)")
    ap.add_argument("--ckpt-every", type=int, default=100)
    ap.add_argument("--ckpt-keep-last", type=int, default=5)
    ap.add_argument("--resume", type=str, help="Resume from checkpoint ('auto' or path)")
    ap.add_argument("--monitor-every", type=int, default=1000)
    ap.add_argument("--index-csv", type=Path, default=Path("data/processed/_index/index.csv"))
    ap.add_argument("--split-manifest", type=Path, help="Split manifest JSON (excludes val set)")
    ap.add_argument("--crop-scale-min", type=float, default=0.3, help="Min scale for RandomResizedCrop (default: 0.3)")
    ap.add_argument("--crop-scale-max", type=float, default=1.0, help="Max scale for RandomResizedCrop (default: 1.0)")
    ap.add_argument("--z-stride", type=int, default=1, help="Keep every Nth slice per series")
    ap.add_argument("--diverse-batches", action="store_true", help="Series-diverse batch sampling")
    ap.add_argument("--train-seed", type=int, default=0)
    ap.add_argument("--sdp-backend", choices=["auto", "math", "mem_efficient", "flash"], default="auto",
                    help="Accepted for compatibility; attention always runs on the MFMA kernels of libdinox_hip")
    ap.add_argument("--run-dir", type=Path, default=Path("data/runs"))
    ap.add_argument("--run-suffix", type=str, help="Optional suffix for run directory name")
    ap.add_argument("--amp", action="store_true", help="Use mixed precision training (bf16 MFMA operands)")
    ap.add_argument("--amp-dtype", choices=["float16", "bfloat16"], default="bfloat16", help="AMP dtype (bfloat16 only on this engine)")
    ap.add_argument("--log-json", type=Path, default=None, help="Write one JSON line per training step to this file")
    # extension (not in the reference): data source for environments without a dataset
    ap.add_argument("--synthetic", type=int, default=0, metavar="N", help="Train on N seeded synthetic HU slice stacks instead of --index-csv")
    ap.add_argument("--local-crops", type=int, default=0, metavar="L",
                    help="Multi-crop extension (the reference trains on 2 global views): L extra student-only local views per sample, "
                         "crop scale 0.05-0.3; needs --gpu-views")
    ap.add_argument("--local-size", type=int, default=96, help="Side of the local views (a multiple of the patch size)")
    ap.add_argument("--hip-graph", action="store_true",
                    help="Replay the whole optimiser step as one captured hipGraph after two eager steps (extension; single GPU, "
                         "--accumulation-steps 1): for small batches, where launching ~300 kernels per step costs more than running them")
    ap.add_argument("--streams", choices=("auto", "on", "off"), default="auto",
                    help="Concurrent launch chains (extension; results are bit-identical either way): the teacher's forward beside the "
                         "student's and the weight-gradient products beside backward, each on a second HIP stream.  auto = both, the "
                         "teacher's stream only below width 1024 (measured: ViT-S +3 .. +9 %, ViT-L -5 % with it); not under --hip-graph")
    ap.add_argument("--stack-cache", type=Path, default=None, metavar="DIR",
                    help="Keep every decoded PNG slice as raw uint16 in a memory-mapped file under DIR (keyed by the file list, sizes and "
                         "mtimes): a slice is decoded once, later epochs read it from the page cache")
    ap.add_argument("--stack-cache-prefill", action="store_true", help="Decode the whole index into --stack-cache before the first step")
    ap.add_argument("--gpu-views", action="store_true",
                    help="Build both views on the GPU (HU decode, window, antialiased bicubic RandomResizedCrop, flip, normalise in one "
                         "kernel); DataLoader workers then only decode PNGs")
    return ap


def resolve_model_config(args) -> ModelConfig:
    presets = _presets()
    if args.config == "custom":
        if not all([args.vit_patch, args.vit_dim, args.vit_depth, args.vit_heads]):
            raise ValueError("Custom config requires: --vit-patch, --vit-dim, --vit-depth, --vit-heads")
        cfg = ModelConfig("custom", args.vit_patch, args.vit_dim, args.vit_depth, args.vit_heads)
    else:
        cfg = presets[args.config]
        over = {k: v for k, v in dict(patch=args.vit_patch, dim=args.vit_dim, depth=args.vit_depth, heads=args.vit_heads).items() if v}
        if over:
            for k, v in over.items():
                setattr(cfg, k, v)
            key = (cfg.patch, cfg.dim, cfg.depth, cfg.heads)
            match = [n for n, c in _presets().items() if (c.patch, c.dim, c.depth, c.heads) == key]
            cfg.name = match[0] if match else "custom"
    if args.out_dim:
        cfg.out_dim = args.out_dim
    return cfg


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    if args.loss_type != "dino":
        raise SystemExit(f"--loss-type {args.loss_type} is not wired to the MI355X engine yet (only 'dino'; see DESIGN.md section 7)")
    if args.amp and args.amp_dtype != "bfloat16":
        raise SystemExit("the HIP path supports --amp-dtype bfloat16 only")
    rank, world, local = init_process_group()
    main_rank = rank == 0

    def say(*a):
        if main_rank:
            print(*a, flush=True)

    model_cfg = resolve_model_config(args)
    say(f"model_config={model_cfg.name} patch={model_cfg.patch} dim={model_cfg.dim} depth={model_cfg.depth} heads={model_cfg.heads} "
        f"out_dim={model_cfg.out_dim} entropy_wall={math.log(model_cfg.out_dim):.4f} params={model_cfg.params_millions:.1f}M "
        f"grad_checkpoint={args.grad_checkpoint}")
    hw = detect_hardware()
    if args.device != "auto":
        hw.device_type = args.device
    if args.num_workers is not None:
        hw.num_workers = args.num_workers
    if args.pin_memory is not None:
        hw.pin_memory = args.pin_memory
    if hw.device_type != "cuda" or not torch.cuda.is_available():
        raise SystemExit("this engine computes on MI355X only: no CUDA/HIP device available or --device cpu requested")
    say(f"hardware={hw.device_name} device_type={hw.device_type} is_rocm={hw.is_rocm} num_workers={hw.num_workers} "
        f"pin_memory={hw.pin_memory} batch_size_rec={hw.batch_size_recommendation}")
    git_commit, data_hash = get_git_commit(), (None if args.synthetic else compute_data_manifest_hash(args.index_csv))
    if git_commit:
        say(f"git_commit={git_commit}")
    if data_hash:
        say(f"data_manifest_hash={data_hash}")
    cfg = TrainingConfig(
        model=model_cfg, hardware=hw, img_size=args.img_size, batch_size=args.batch_size, accumulation_steps=args.accumulation_steps,
        lr=args.lr, min_lr=args.min_lr, warmup_steps=args.warmup_steps, weight_decay=args.weight_decay, max_steps=args.max_steps,
        ema=args.ema, teacher_temp=args.teacher_temp, student_temp=args.student_temp, center_momentum=args.center_momentum,
        loss_type=args.loss_type, gram_enabled=True, gram_weight=args.gram_weight, koleo_weight=args.koleo_weight,
        scale_aware=args.scale_aware, crop_scale_min=args.crop_scale_min, crop_scale_max=args.crop_scale_max, z_stride=args.z_stride,
        diverse_batches=args.diverse_batches, ckpt_every=args.ckpt_every, ckpt_keep_last=args.ckpt_keep_last,
        monitor_every=args.monitor_every, train_seed=args.train_seed, sdp_backend=args.sdp_backend, amp_dtype=args.amp_dtype,
        index_csv=str(args.index_csv), split_manifest=str(args.split_manifest) if args.split_manifest else None,
        git_commit=git_commit, data_manifest_hash=data_hash)
    say(f"effective_batch_size={cfg.effective_batch_size * world} (batch={args.batch_size} × accum={args.accumulation_steps} × ranks={world})")
    _seed_all(args.train_seed)
    local = local % torch.cuda.device_count()          # (rehearsals may put several gloo ranks on one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    say(f"device={device.type}")
    say(f"torch_version={torch.__version__}")
    say(f"amp={args.amp} dtype={args.amp_dtype} grad_scaler=False")

    # ---- run directory / resume
    resume_from: Optional[Path] = None
    if args.resume:
        if args.resume == "auto":
            runs = sorted(d for d in args.run_dir.iterdir() if d.is_dir()) if args.run_dir.exists() else []
            if not runs:
                raise FileNotFoundError(f"No run directories found in {args.run_dir}")
            run_dir = runs[-1]
            resume_from = find_latest_checkpoint(run_dir)
            if not resume_from:
                raise FileNotFoundError(f"No checkpoint found in {run_dir}")
        else:
            resume_from = Path(args.resume)
            run_dir = args.run_dir / f"{time.strftime('%Y%m%d_%H%M%S')}_{args.run_suffix}" if args.run_suffix else resume_from.parent
    else:
        rid = time.strftime("%Y%m%d_%H%M%S") + (f"_{args.run_suffix}" if args.run_suffix else "")
        run_dir = args.run_dir / rid
    if main_rank:
        run_dir.mkdir(parents=True, exist_ok=True)
        (run_dir / "config.json").write_text(json.dumps(asdict(cfg), indent=2) + "\n")
    say(f"run_dir={run_dir}")

    # ---- data
    ds_kw = dict(img_size=args.img_size, rw_level_min=cfg.rw_level_min, rw_level_max=cfg.rw_level_max, rw_width_min=cfg.rw_width_min,
                 rw_width_max=cfg.rw_width_max, scale_aware=args.scale_aware, crop_scale_min=args.crop_scale_min, crop_scale_max=args.crop_scale_max)
    if args.synthetic:
        ds = SyntheticSliceDataset(args.synthetic, seed=args.train_seed, **ds_kw)
        rows = ds.rows
        say(f"synthetic_rows={len(rows)} scale_aware={args.scale_aware}")
    else:
        rows = _load_index_rows(args.index_csv, require_spacing=args.scale_aware)
        say(f"loaded_rows={len(rows)} scale_aware={args.scale_aware}")
        if args.split_manifest and args.split_manifest.exists():
            val = set(str(s) for s in json.loads(args.split_manifest.read_text()).get("val", {}).get("series_dir", []))
            before = len(rows)
            rows = [r for r in rows if str(r.series_dir) not in val]
            say(f"excluded_val_series={len(val)} excluded_rows={before - len(rows)}")
        if args.z_stride > 1:
            by_series: dict = {}
            for r in rows:
                by_series.setdefault(r.series_dir, []).append(r)
            strided = []
            for s in sorted(by_series):
                strided.extend(sorted(by_series[s], key=lambda r: r.slice_index)[::args.z_stride])
            say(f"z_stride={args.z_stride} rows_before={len(rows)} rows_after={len(strided)}")
            rows = strided
        ds = PngDataset(rows, **ds_kw)
        if args.stack_cache is not None:
            from dinox.stackcache import SliceCache
            t_c = time.time()
            ds.cache = SliceCache([r.png_path for r in rows], args.stack_cache)
            if args.stack_cache_prefill:
                if main_rank:
                    ds.cache.prefill(workers=hw.num_workers, say=say)
                if world > 1:
                    torch.distributed.barrier()
            say(f"stack_cache={ds.cache.dir} slices={len(ds.cache)} cached={ds.cache.filled()} bytes={2 * ds.cache.total} "
                f"setup_s={time.time() - t_c:.1f}")
    if len(ds) < args.batch_size:
        say(f"⚠️  Dataset size ({len(ds)}) is smaller than batch size ({args.batch_size}). Reducing batch size to {len(ds)}.")
        args.batch_size = cfg.batch_size = len(ds)
    gen = torch.Generator().manual_seed(args.train_seed)      # the SAME sample order on every rank; ranks keep disjoint slices of it

    def _worker_init(worker_id: int) -> None:
        _seed_all(args.train_seed + 1000 * rank + worker_id)

    if args.local_crops and not args.gpu_views:
        raise SystemExit("--local-crops needs --gpu-views (local crops are produced by the device-side view pipeline)")
    if args.local_crops and args.local_size % model_cfg.patch:
        raise SystemExit(f"--local-size {args.local_size} is not a multiple of the patch size {model_cfg.patch}")
    ds.raw_views = bool(args.gpu_views)
    ds.local_crops = int(args.local_crops)
    if args.gpu_views:
        import dinox.views as _views
        from dinox.views import collate_stacks, make_views
        if hw.pin_memory and hw.num_workers > 0:          # (before the workers are forked: they inherit the setting)
            _views.SHM_RING = 4                           # prefetch_factor 2 + the batch being pinned + one spare
        say(f"gpu_views=True shm_ring={_views.SHM_RING}")
    common = dict(num_workers=hw.num_workers, pin_memory=hw.pin_memory, worker_init_fn=_worker_init,       # (StackBatch.pin_memory for --gpu-views)
                  collate_fn=collate_stacks if args.gpu_views else dino_collate, persistent_workers=hw.num_workers > 0)
    if len(ds) < args.batch_size * world:
        raise SystemExit(f"dataset size ({len(ds)}) is smaller than the global batch ({args.batch_size} x {world} ranks)")
    if args.diverse_batches:
        sampler = DiverseBatchSampler(rows, batch_size=args.batch_size * world, drop_last=True, generator=gen)
        say(f"diverse_batches=True batches_per_epoch={len(sampler)}")
    else:
        sampler = torch.utils.data.BatchSampler(torch.utils.data.RandomSampler(ds, generator=gen), batch_size=args.batch_size * world,
                                                drop_last=True)
    if world > 1:
        sampler = ShardedBatchSampler(sampler, rank, world)
    dl = torch.utils.data.DataLoader(ds, batch_sampler=sampler, **common)
    # (iter(dl) draws the loader's base seed from torch's global generator: both pipelines do it here, before the model is initialised,
    #  so that the same --train-seed gives the same initial weights with and without --gpu-views)
    it, prefetch = None, None
    # --gpu-views writes the patch-embed operand itself (dinox_slice_views_patches): the fp32 image batch never exists.
    # DINOX_VIEWS_IMAGE=1: the image batch + the separate unfold launch instead (A/B; bit-identical operand)
    view_kw = {} if os.environ.get("DINOX_VIEWS_IMAGE") else dict(patch=model_cfg.patch, operand_dtype=torch.bfloat16 if args.amp else torch.float32)
    if args.gpu_views:
        from dinox.views import DevicePrefetcher
        prefetch = DevicePrefetcher(dl, device, ahead=not os.environ.get("DINOX_NO_PREFETCH"))     # the next batch crosses PCIe under this step
    else:
        it = iter(dl)

    # ---- model / engine
    vit_kw = dict(img_size=args.img_size, patch=model_cfg.patch, dim=model_cfg.dim, depth=model_cfg.depth, heads=model_cfg.heads,
                  mlp_ratio=model_cfg.mlp_ratio, use_grad_checkpoint=args.grad_checkpoint, scale_aware=args.scale_aware)
    student = DinoStudentTeacher(PatchViT(**vit_kw), out_dim=model_cfg.out_dim).to(device)
    teacher = DinoStudentTeacher(PatchViT(**vit_kw), out_dim=model_cfg.out_dim).to(device)
    teacher.load_state_dict(student.state_dict())
    hp = StepHyperParams(lr=args.lr, min_lr=args.min_lr, warmup_steps=args.warmup_steps, max_steps=args.max_steps,
                         weight_decay=args.weight_decay, ema=args.ema, teacher_temp=args.teacher_temp, student_temp=args.student_temp,
                         center_momentum=args.center_momentum, gram_weight=args.gram_weight,
                         koleo_weight=args.koleo_weight)
    if args.hip_graph and (world > 1 or args.accumulation_steps != 1 or args.local_crops):
        raise SystemExit("--hip-graph: single GPU, --accumulation-steps 1 and no --local-crops (the captured step has one fixed batch layout)")
    if device.type == "cuda" and not args.hip_graph and args.streams != "off":
        # (the environment wins: DINOX_SIDE_STREAM / DINOX_DW_STREAM set by the user are left alone)
        if "DINOX_SIDE_STREAM" not in os.environ and (args.streams == "on" or model_cfg.dim < 1024):
            os.environ["DINOX_SIDE_STREAM"] = "1"
        if "DINOX_DW_STREAM" not in os.environ:
            ops.dw_stream.enabled = True
        say(f"streams side={int(bool(os.environ.get('DINOX_SIDE_STREAM')))} dw={int(ops.dw_stream.enabled)}")
    eng = TrainEngine(student, teacher, model_cfg.out_dim, hp, amp_dtype=torch.bfloat16 if args.amp else None,
                      accumulation_steps=args.accumulation_steps, use_graph=bool(args.hip_graph))
    start_step = 0
    if resume_from:
        say(f"resume=true checkpoint={resume_from}")
        start_step, loaded = load_checkpoint(resume_from, student, teacher, eng, device, scale_aware=args.scale_aware)
        say(f"resumed_from_step={start_step}")
        if loaded and loaded.model and loaded.model.name != model_cfg.name:
            warnings.warn(f"Model config mismatch: checkpoint={loaded.model.name} requested={model_cfg.name}")
    say("tensorboard not installed, skipping TB logging")

    stop = _StopFlag()
    signal.signal(signal.SIGINT, lambda *_: setattr(stop, "stop", True))
    loss_history: list = []
    t0 = last_log = time.time()
    max_steps = args.max_steps if args.max_steps else 10 ** 9
    say(f"Starting training from step {start_step} to {max_steps if args.max_steps else 'unlimited'}")
    say("─" * 80)
    step = start_step - 1
    # (step, loss tensor, lr) of earlier steps, read back LOSS_LAG steps late: the device always has queued work while the host waits
    # for data (one step of lag left the GPU idle ~5 ms per step whenever a batch arrived late; the logged values are the same)
    LOSS_LAG = 2
    pending: list = []
    loss_pin = torch.empty(LOSS_LAG + 1, dtype=torch.float32).pin_memory() if device.type == "cuda" else None

    def read_loss(ref) -> float:
        if isinstance(ref, tuple):
            slot, ev = ref
            ev.synchronize()
            return float(loss_pin[slot])
        return float(ref)

    _prof = {"data": 0.0, "views": 0.0, "step": 0.0, "n": 0, "t0": time.time()}     # host time per phase: DINOX_CLI_PROFILE=<first step counted>
    _prof_from = int(os.environ.get("DINOX_CLI_PROFILE") or 0)
    stop_every = 10                     # under DP the ranks agree on an interrupt only at these steps (one tiny all-reduce + sync)
    for step in range(start_step, int(max_steps)):
        if world > 1:
            if (step - start_step) % stop_every == 0:
                flag = torch.tensor([1.0 if stop.stop else 0.0], device=device)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
                stop.agreed = bool(flag.item())
            stopping = getattr(stop, "agreed", False)
        else:
            stopping = stop.stop
        if stopping:
            say("interrupt=true")
            step -= 1
            break
        loc = spl = None
        if args.gpu_views:
            if _prof_from and prefetch.timing is None:
                prefetch.timing = []
            _t0 = time.perf_counter()
            sb = prefetch.next()
            _prof["data"] += time.perf_counter() - _t0
            if _prof_from:
                if _prof.get("ev") is not None and _prof["ev"][1].query():
                    _prof["gpu"] = _prof.get("gpu", 0.0) + _prof["ev"][0].elapsed_time(_prof["ev"][1])
                    _prof["gpu_n"] = _prof.get("gpu_n", 0) + 1
                _prof["ev"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                _prof["ev"][0].record()
            _t0 = time.perf_counter()
            batch, spacing = make_views(sb, args.img_size, views=sb.views[:2], **view_kw), sb.spacing
            _prof["views"] += time.perf_counter() - _t0
            if args.local_crops:
                loc = make_views(sb, args.local_size, views=sb.views[2:], **view_kw)
                spl = torch.cat([spacing] * args.local_crops, 0) if args.scale_aware else None
        else:
            try:
                item = next(it)
            except StopIteration:
                it = iter(dl)
                item = next(it)
            views, spacing = item
            # (each view batch is page-locked by the loader: copy first, concatenate on the device -- a host-side cat would make a pageable
            #  tensor whose copy is synchronous)
            batch = torch.cat([v.to(device, non_blocking=True) for v in views], 0)
        sp2 = torch.cat([spacing, spacing], 0).to(device, non_blocking=True) if args.scale_aware else None
        _t0 = time.perf_counter()
        out = eng.step(batch, sp2, loc, spl)
        _prof["step"] += time.perf_counter() - _t0
        _prof["n"] += 1
        _prof["t1"] = time.time()
        if _prof_from and _prof.get("ev") is not None:
            _prof["ev"][1].record()
        if _prof_from and step == start_step + _prof_from:      # steady state only: drop what the start-up steps accumulated
            _prof.update(data=0.0, views=0.0, step=0.0, n=0, t0=time.time(), gpu=0.0, gpu_n=0)
        loss_t = out["loss"]
        if world > 1:                   # the logged loss (and the NaN guard on it) is the global-batch mean, identical on every rank
            loss_t = loss_t.clone()
            torch.distributed.all_reduce(loss_t, op=torch.distributed.ReduceOp.SUM)
            loss_t = loss_t / world
        # The loss of step s travels to a page-locked slot by an asynchronous copy and is read LOSS_LAG steps later, after waiting for ITS
        # event only.  (float(tensor) / .item() synchronises the whole stream: with it the host could never run ahead of the device,
        # which then idled for the ~7 ms the host needs to fetch and launch the next step.)
        if loss_pin is not None:
            slot = step % (LOSS_LAG + 1)
            loss_pin[slot:slot + 1].copy_(loss_t.detach().reshape(1).float(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            cur = (step, (slot, ev), out["lr"])
        else:
            cur = (step, loss_t, out["lr"])
        pending.append(cur)
        for (s_, loss_t, lr_) in ([pending.pop(0)] if len(pending) > LOSS_LAG else []):
            loss_val = read_loss(loss_t)
            loss_history.append(loss_val)
            if args.log_json is not None and main_rank:
                with open(args.log_json, "a") as jf:
                    jf.write(json.dumps({"step": s_, "loss": round(loss_val, 6), "lr": lr_}) + "\n")
            now = time.time()
            if now - last_log >= 10.0 or s_ == start_step:
                sps = (s_ - start_step + 1) / max(now - t0, 1e-6)
                say(f"step={s_:6d} loss={loss_val:.4f} lr={lr_:.2e} steps/s={sps:.2f} samples/s={sps * cfg.effective_batch_size * world:.1f} "
                    f"elapsed={now - t0:.1f}s")
                last_log = now
            bad, msg = detect_anomaly(loss_val, loss_history[:-1])
            if bad and ("NaN" in msg or "Inf" in msg):
                say(f"❌ CRITICAL: {msg}")
                if main_rank:
                    save_checkpoint(run_dir / f"emergency_checkpoint_step{s_}.pth", s_, student, teacher, eng, cfg)
                raise RuntimeError(msg)
            if bad:
                say(f"⚠️  WARNING: {msg}")
        if (step + 1) % args.ckpt_every == 0 and main_rank:
            path = run_dir / f"checkpoint_{step + 1:08d}.pth"
            save_checkpoint(path, step + 1, student, teacher, eng, cfg)
            say(f"checkpoint_saved={path}")
            rotate_checkpoints(run_dir, args.ckpt_keep_last)
    for s_, loss_t, lr_ in pending:
        loss_val = read_loss(loss_t)
        loss_history.append(loss_val)
        if args.log_json is not None and main_rank:
            with open(args.log_json, "a") as jf:
                jf.write(json.dumps({"step": s_, "loss": round(loss_val, 6), "lr": lr_}) + "\n")
    final_step = step + 1
    if main_rank:
        final = run_dir / f"checkpoint_final_{final_step:08d}.pth"
        save_checkpoint(final, final_step, student, teacher, eng, cfg)
        say(f"final_checkpoint={final}")
    say("─" * 80)
    say(f"Training complete: {final_step - start_step} steps in {time.time() - t0:.1f}s")
    if os.environ.get("DINOX_CLI_PROFILE") and _prof["n"]:
        n = _prof["n"]
        say(f"host ms/step: wait for data {1e3 * _prof['data'] / n:.1f}, view parameters + launch {1e3 * _prof['views'] / n:.1f}, "
            f"engine enqueue {1e3 * _prof['step'] / n:.1f}, wall {1e3 * (_prof.get('t1', time.time()) - _prof['t0']) / n:.1f} (over the last {n} steps); "
            f"device time views + step {_prof.get('gpu', 0.0) / max(_prof.get('gpu_n', 0), 1):.1f} ms")
        if prefetch is not None and prefetch.timing:
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in prefetch.timing[-200:])
            say(f"H2D copy of a batch on the copy stream: median {ts[len(ts) // 2]:.1f} ms, max {ts[-1]:.1f} ms ({sb.raw.numel() * 2 / 1e6:.0f} MB)")
    say(f"Final loss: {loss_history[-1]:.4f}" if loss_history else "Final loss: N/A")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
