"""``zoo.hub`` surface: build a PatchViT (HIP engine) from a training checkpoint, a hub directory or a
HuggingFace id, and export it back.  Behaviour mirrors the reference (zoo/hub.py:74-327):

* ``load_model(path_or_id, *, device="cpu", config_override=None) -> PatchViT`` in eval mode;
* ``*.pth`` file  -> training checkpoint (``student``/``model``/bare state dict; config =
  DEFAULT_CONFIG <- ckpt["config"]["model"] <- {img_size, scale_aware} <- override; old keys migrated;
  ``backbone.`` stripped; ``head.*`` dropped; ``scale_embed.*`` dropped when not scale-aware; strict=False);
* directory with ``config.json`` -> hub format (``backbone.safetensors`` preferred, else ``backbone.pth``;
  strict=True);
* anything else -> ``huggingface_hub.snapshot_download`` (needs network);
* errors: ``FileNotFoundError`` for missing files, ``ImportError`` for missing optional packages.

The returned model computes only on a CUDA/HIP device (pass ``device="cuda"`` to run ``encode``).
"""
from __future__ import annotations

import json
import logging
import os
import pickle
from pathlib import Path
from typing import Any, Dict, Union

import torch

from zoo.arch import DinoStudentTeacher, PatchViT, migrate_state_dict, needs_migration  # noqa: F401

log = logging.getLogger(__name__)

DEFAULT_CONFIG: Dict[str, Any] = dict(
    img_size=224, patch=16, dim=384, depth=6, heads=6, mlp_ratio=4.0, num_registers=4, scale_aware=False, out_dim=8192)

_VIT_KEYS = ("img_size", "patch", "dim", "depth", "heads", "mlp_ratio", "num_registers", "scale_aware")


def _build_backbone(config: Dict[str, Any]) -> PatchViT:
    return PatchViT(**{k: config.get(k, DEFAULT_CONFIG[k]) for k in _VIT_KEYS})


def _strip_prefix(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    n = len(prefix)
    return {(k[n:] if k.startswith(prefix) else k): v for k, v in sd.items()}


def _rng_blob_globals() -> list:
    """The only non-tensor objects a training checkpoint of this path carries (reference scripts/phase5_big_run.py:1041-1057):
    ``np.random.get_state()`` -- a tuple holding one uint32 ndarray -- beside plain python containers."""
    import numpy as np
    try:
        from numpy._core.multiarray import _reconstruct
    except ImportError:                                     # numpy < 2
        from numpy.core.multiarray import _reconstruct
    return [_reconstruct, np.ndarray, np.dtype, type(np.dtype(np.uint32)), type(np.dtype(np.float64)), type(np.dtype(np.int64))]


def read_checkpoint(path: Path, device) -> Any:
    """torch.load with the restricted (weights_only) unpickler; training checkpoints, whose RNG blob holds a NumPy array,
    get an allow-list of exactly those NumPy globals.  Nothing else is ever unpickled unless the user opts in with
    DINOX_ALLOW_PICKLE=1 (the reference itself loads these files with weights_only=False, zoo/hub.py:99)."""
    try:
        return torch.load(path, map_location=device, weights_only=True)
    except pickle.UnpicklingError:
        pass
    try:
        with torch.serialization.safe_globals(_rng_blob_globals()):
            return torch.load(path, map_location=device, weights_only=True)
    except pickle.UnpicklingError as e:
        if os.environ.get("DINOX_ALLOW_PICKLE") == "1":
            log.warning("%s: full unpickling (DINOX_ALLOW_PICKLE=1)", path.name)
            return torch.load(path, map_location=device, weights_only=False)
        raise pickle.UnpicklingError(
            f"{path}: holds objects beyond tensors, containers and the NumPy RNG state; set DINOX_ALLOW_PICKLE=1 to load a "
            f"file you trust with the full unpickler ({e})") from e


_read_checkpoint = read_checkpoint


def load_from_training_checkpoint(path: Union[str, Path], *, device: Union[str, torch.device] = "cpu",
                                  config_override: Dict[str, Any] | None = None) -> PatchViT:
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    payload = _read_checkpoint(path, device)

    config = dict(DEFAULT_CONFIG)
    ck = payload.get("config") if isinstance(payload, dict) else None
    if isinstance(ck, dict):
        if isinstance(ck.get("model"), dict):
            config.update(ck["model"])
        config.update({k: ck[k] for k in ("img_size", "scale_aware") if k in ck})
    if config_override:
        config.update(config_override)

    backbone = _build_backbone(config)

    sd = payload
    for name in ("student", "model"):
        if isinstance(payload, dict) and name in payload:
            sd = payload[name]
            break
    if needs_migration(sd):
        log.info("migrating old-format state dict keys")
        sd = migrate_state_dict(sd)
    if any(k.startswith("backbone.") for k in sd):
        sd = _strip_prefix(sd, "backbone.")
    drop = ("head.",) if config.get("scale_aware", False) else ("head.", "scale_embed.")
    sd = {k: v for k, v in sd.items() if not k.startswith(drop)}

    backbone.load_state_dict(sd, strict=False)
    backbone.to(device).eval()
    log.info("loaded backbone from %s (dim=%d depth=%d scale_aware=%s)", path.name, config["dim"], config["depth"],
             config.get("scale_aware", False))
    return backbone


def load_from_hub_dir(model_dir: Union[str, Path], *, device: Union[str, torch.device] = "cpu") -> PatchViT:
    model_dir = Path(model_dir)
    cfg_path = model_dir / "config.json"
    if not cfg_path.exists():
        raise FileNotFoundError(f"config.json not found in {model_dir}")
    backbone = _build_backbone(json.loads(cfg_path.read_text()))

    st_path, pth_path = model_dir / "backbone.safetensors", model_dir / "backbone.pth"
    if st_path.exists():
        try:
            from safetensors.torch import load_file
        except ImportError:
            raise ImportError("safetensors is required to load .safetensors files. Install with: pip install safetensors")
        sd = load_file(str(st_path), device=str(device))
    elif pth_path.exists():
        sd = torch.load(pth_path, map_location=device, weights_only=True)
    else:
        raise FileNotFoundError(f"No weights found in {model_dir}. Expected backbone.safetensors or backbone.pth")
    if needs_migration(sd):
        sd = migrate_state_dict(sd)
    backbone.load_state_dict(sd, strict=True)
    backbone.to(device).eval()
    return backbone


def load_model(model_id_or_path: str, *, device: Union[str, torch.device] = "cpu",
               config_override: Dict[str, Any] | None = None) -> PatchViT:
    p = Path(model_id_or_path)
    if p.is_file() and p.suffix == ".pth":
        return load_from_training_checkpoint(p, device=device, config_override=config_override)
    if p.is_dir() and (p / "config.json").exists():
        return load_from_hub_dir(p, device=device)
    try:
        from huggingface_hub import snapshot_download
    except ImportError:
        raise ImportError(f"Cannot load '{model_id_or_path}': not a local file/directory, and huggingface_hub is not "
                          "installed. Install with: pip install huggingface_hub")
    log.info("downloading %s from the HuggingFace Hub", model_id_or_path)
    return load_from_hub_dir(snapshot_download(model_id_or_path), device=device)


def export_hub_checkpoint(backbone: PatchViT, output_dir: Union[str, Path], *, config: Dict[str, Any] | None = None,
                          use_safetensors: bool = False) -> Path:
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    if config is None:
        config = dict(img_size=backbone.img_size, patch=backbone.patch, dim=backbone.dim, depth=len(backbone.blocks),
                      heads=backbone.blocks[0].attn.num_heads, mlp_ratio=4.0, num_registers=backbone.num_registers,
                      scale_aware=backbone.scale_aware)
    (out / "config.json").write_text(json.dumps(config, indent=2))
    sd = {k: v.detach().cpu().contiguous() for k, v in backbone.state_dict().items()}
    if use_safetensors:
        try:
            from safetensors.torch import save_file
        except ImportError:
            raise ImportError("safetensors required. Install with: pip install safetensors")
        save_file(sd, str(out / "backbone.safetensors"))
    else:
        torch.save(sd, out / "backbone.pth")
    return out
