"""``zoo.models`` surface: Pydantic schemas of the DINO-X dataset catalogue and training lineage.

Not neural-network code: kept so that ``from zoo.models import ...`` keeps
working next to the HIP engine.  Field names, types and defaults follow the reference schemas
(zoo/models.py:20-141); ``timezone.utc`` replaces ``datetime.UTC`` so the module imports on py3.10.
"""
from __future__ import annotations

from datetime import datetime, timezone
from typing import Dict, List, Literal, Optional, Tuple, Union

from pydantic import BaseModel, Field

Modality = Literal["ct", "mri", "xray"]


class PreprocessingConfig(BaseModel):
    """Raw data -> training-ready encoding (16-bit PNG: stored = HU * scale + hu_shift)."""
    format: Literal["png_16bit", "png_8bit", "npy", "nifti"] = "png_16bit"
    hu_shift: int = 32768
    scale: int = 10
    index_csv: str = ""


class DatasetEntry(BaseModel):
    """One dataset of the catalogue (a YAML file under zoo/datasets/<modality>/)."""
    name: str
    modality: Modality
    organs: List[str]
    source_url: str = ""
    license: str = ""
    total_slices: int = 0
    total_series: int = 0
    pixel_spacing_range: Tuple[float, float] = (0.0, 0.0)
    slice_thickness_range: Tuple[float, float] = (0.0, 0.0)
    hu_range: Tuple[int, int] = (-1024, 3071)
    annotations: List[str] = Field(default_factory=list)
    preprocessing: PreprocessingConfig = Field(default_factory=PreprocessingConfig)
    citation: str = ""
    notes: str = ""


class SliceMetadata(BaseModel):
    """Per-slice physical metadata (Parquet rows); the spacing triple feeds ScaleEmbedding."""
    dataset: str
    series_id: str
    slice_idx: int
    pixel_spacing_x: float
    pixel_spacing_y: float
    slice_thickness: float
    image_path: str
    organs_present: List[str] = Field(default_factory=list)
    patient_id: Optional[str] = None
    study_date: Optional[str] = None


class DatasetUsage(BaseModel):
    """How one dataset entered a training run."""
    name: str
    slices_used: int
    weight: float
    pixel_spacing_min: float = 0.0
    pixel_spacing_max: float = 0.0
    slice_thickness_min: float = 0.0
    slice_thickness_max: float = 0.0


class SpacingStats(BaseModel):
    """Corpus-level spacing statistics."""
    pixel_spacing_x_min: float = 0.0
    pixel_spacing_x_max: float = 0.0
    pixel_spacing_x_mean: float = 0.0
    pixel_spacing_y_min: float = 0.0
    pixel_spacing_y_max: float = 0.0
    pixel_spacing_y_mean: float = 0.0
    slice_thickness_min: float = 0.0
    slice_thickness_max: float = 0.0
    slice_thickness_mean: float = 0.0


def _utc_now() -> str:
    return datetime.now(timezone.utc).isoformat()


class TrainingLineage(BaseModel):
    """Provenance record written as lineage.json beside a checkpoint."""
    model_name: str
    architecture: str = "vit-small"
    modality: Modality = "ct"
    datasets: List[DatasetUsage] = Field(default_factory=list)
    total_slices: int = 0
    spacing_stats: SpacingStats = Field(default_factory=SpacingStats)
    scale_aware: bool = False
    training_config: Dict[str, Union[str, int, float, bool]] = Field(default_factory=dict)
    random_seed: int = 42
    timestamp: str = Field(default_factory=_utc_now)
    tool_version: str = ""
    training_code_commit: str = ""
    data_catalog_hash: str = ""

    def total_weight(self) -> float:
        return sum(d.weight for d in self.datasets)
