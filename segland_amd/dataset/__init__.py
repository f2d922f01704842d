"""Datasets of the drivers.  `synthetic` (ready float tiles) and `synthetic_raw` (raw uint8 tiles through the GPU tile preparation) ship with
the build for benchmarks, smoke runs and tests; `oem` reads OpenEarthMap GeoTIFF tiles (dataset/oem.py of the reference; decode: dataset/tiff.py --
rasterio when installed, else Pillow) and prepares them on the GPU; `oem_ft` / `synthetic_raw_ft` are the fine-tune PAIR readers
(dataset/oem_ft.py of the reference); `synthetic_tiff` / `synthetic_tiff_ft` run the real `oem` / `oem_ft` readers on a generated directory of TIFF tiles."""
from . import oem, oem_ft, synthetic, synthetic_ft, synthetic_raw, synthetic_raw_ft, synthetic_tiff, synthetic_tiff_ft  # noqa: F401
