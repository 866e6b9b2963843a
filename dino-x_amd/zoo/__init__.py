"""Drop-in ``zoo`` surface of DINO-X, backed by the MI355X HIP engine.

    from zoo.hub import load_model
    from zoo.encode import encode
    from zoo.arch import PatchViT, ScaleEmbedding
"""
from __future__ import annotations
