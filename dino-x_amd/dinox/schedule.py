"""Learning-rate schedule of the reference loop (scripts/phase5_big_run.py:653-676): linear warm-up
``base*(step+1)/warmup``, cosine decay to ``min_lr`` at ``total_steps``, constant ``base`` when the
run is unlimited (``total_steps is None``), ``min_lr`` past the end.  ``step`` counts micro-batches."""
from __future__ import annotations

import math
from typing import Optional


def get_lr(step: int, total_steps: Optional[int], warmup_steps: int, base_lr: float, min_lr: float) -> float:
    if step < warmup_steps:
        return base_lr * (step + 1) / warmup_steps
    if total_steps is None:
        return base_lr
    if step >= total_steps:
        return min_lr
    progress = (step - warmup_steps) / (total_steps - warmup_steps)
    return min_lr + (base_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * progress))
