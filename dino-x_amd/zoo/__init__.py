"""The ``zoo`` package name of timlawrenz/DINO-X, served by the MI355X HIP engine (dinox).

Sub-modules mirror the reference's import paths so its call sites keep working: ``zoo.arch`` (model classes), ``zoo.hub``
(``load_model``), ``zoo.encode`` (``encode``, ``encode_batch``), ``zoo.models`` (catalogue / lineage records).
Nothing is imported eagerly: importing ``zoo.models`` alone must not load the kernel library.
"""
