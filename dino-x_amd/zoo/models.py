"""``zoo.models`` surface: Pydantic schemas of the DINO-X dataset catalogue and training lineage.

Not neural-network code: kept so that ``from zoo.models import ...`` keeps working next to the HIP
engine.  The reference declares these records as hand-written ``BaseModel`` classes (zoo/models.py:20-141); here each record is
one row of a field table -- (name, type, default) -- turned into a model by ``pydantic.create_model``, which is all a record
with no behaviour needs.  Field names, types and defaults are the reference's; ``timezone.utc`` stands in for ``datetime.UTC``
so the module imports on Python 3.10.
"""
from __future__ import annotations

from datetime import datetime, timezone
from typing import Dict, List, Literal, Optional, Tuple, Union

from pydantic import BaseModel, Field, create_model

Modality = Literal["ct", "mri", "xray"]
_REQUIRED = ...
_Range = Tuple[float, float]
_Scalar = Union[str, int, float, bool]


def _utc_now() -> str:
    return datetime.now(timezone.utc).isoformat()


def _record(name: str, doc: str, fields, base=BaseModel):
    """fields: iterable of (name, type, default); a callable default becomes a default_factory."""
    spec = {}
    for fname, ftype, default in fields:
        spec[fname] = (ftype, Field(default_factory=default) if callable(default) else default)
    model = create_model(name, __base__=base, __module__=__name__, **spec)
    model.__doc__ = doc
    return model


def _zeros(*names):
    return [(n, float, 0.0) for n in names]


PreprocessingConfig = _record(
    "PreprocessingConfig", "Raw data -> training-ready encoding (16-bit PNG: stored = HU * scale + hu_shift).",
    [("format", Literal["png_16bit", "png_8bit", "npy", "nifti"], "png_16bit"), ("hu_shift", int, 32768), ("scale", int, 10),
     ("index_csv", str, "")])

DatasetEntry = _record(
    "DatasetEntry", "One dataset of the catalogue (a YAML file under zoo/datasets/<modality>/).",
    [("name", str, _REQUIRED), ("modality", Modality, _REQUIRED), ("organs", List[str], _REQUIRED)]
    + [(n, str, "") for n in ("source_url", "license")] + [(n, int, 0) for n in ("total_slices", "total_series")]
    + [("pixel_spacing_range", _Range, (0.0, 0.0)), ("slice_thickness_range", _Range, (0.0, 0.0)), ("hu_range", Tuple[int, int], (-1024, 3071)),
       ("annotations", List[str], list), ("preprocessing", PreprocessingConfig, PreprocessingConfig), ("citation", str, ""), ("notes", str, "")])

SliceMetadata = _record(
    "SliceMetadata", "Per-slice physical metadata (Parquet rows); the spacing triple feeds ScaleEmbedding.",
    [("dataset", str, _REQUIRED), ("series_id", str, _REQUIRED), ("slice_idx", int, _REQUIRED)]
    + [(n, float, _REQUIRED) for n in ("pixel_spacing_x", "pixel_spacing_y", "slice_thickness")]
    + [("image_path", str, _REQUIRED), ("organs_present", List[str], list), ("patient_id", Optional[str], None), ("study_date", Optional[str], None)])

DatasetUsage = _record(
    "DatasetUsage", "How one dataset entered a training run.",
    [("name", str, _REQUIRED), ("slices_used", int, _REQUIRED), ("weight", float, _REQUIRED)]
    + _zeros("pixel_spacing_min", "pixel_spacing_max", "slice_thickness_min", "slice_thickness_max"))

SpacingStats = _record(
    "SpacingStats", "Corpus-level spacing statistics.",
    _zeros(*[f"{axis}_{stat}" for axis in ("pixel_spacing_x", "pixel_spacing_y", "slice_thickness") for stat in ("min", "max", "mean")]))


class _LineageBehaviour(BaseModel):
    def total_weight(self) -> float:
        return sum(d.weight for d in self.datasets)


TrainingLineage = _record(
    "TrainingLineage", "Provenance record written as lineage.json beside a checkpoint.",
    [("model_name", str, _REQUIRED), ("architecture", str, "vit-small"), ("modality", Modality, "ct"), ("datasets", List[DatasetUsage], list),
     ("total_slices", int, 0), ("spacing_stats", SpacingStats, SpacingStats), ("scale_aware", bool, False),
     ("training_config", Dict[str, _Scalar], dict), ("random_seed", int, 42), ("timestamp", str, _utc_now)]
    + [(n, str, "") for n in ("tool_version", "training_code_commit", "data_catalog_hash")],
    base=_LineageBehaviour)
