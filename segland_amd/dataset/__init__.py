"""Datasets of the drivers.  `synthetic` (ready float tiles) and `synthetic_raw` (raw uint8 tiles through the GPU tile preparation) ship with
the build for benchmarks, smoke runs and tests; `oem` reads OpenEarthMap GeoTIFF tiles (dataset/oem.py of the reference; needs rasterio, which
this image does not have -- the readers raise without it) and prepares them on the GPU; `oem_ft` /
`synthetic_raw_ft` are the fine-tune PAIR readers (dataset/oem_ft.py of the reference)."""
from . import oem, oem_ft, synthetic, synthetic_ft, synthetic_raw, synthetic_raw_ft  # noqa: F401
