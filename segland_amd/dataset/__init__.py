"""Datasets of the drivers.  `synthetic` ships with the build (benchmarks, smoke runs, tests); `oem` (OpenEarthMap
GeoTIFF tiles, dataset/oem.py + oem_ft.py of the reference) is row f-2 of SURVEY.md section 8 -- next, not built yet."""
from . import synthetic, synthetic_ft  # noqa: F401
