"""Tile decode for the OpenEarthMap readers: `read_tiff(path) -> uint8 [bands, H, W]`, what `rasterio.open(path).read()` returns for the 8-bit RGB
image tiles and single-band label tiles the reference reads (dataset/oem.py:57-58, dataset/oem_ft.py:193-200).

rasterio is used when it is installed.  It is absent from the build image, so the default decoder is Pillow, which parses the same files: baseline / tiled TIFF,
uncompressed or LZW / Deflate / PackBits compressed (through libtiff), 8 bits per sample; the GeoTIFF georeferencing tags are ignored by both paths (the
reference never looks at them in training, only when it writes predictions).  Anything else -- 16-bit samples, more than four bands, a palette -- raises: a wrong
guess about the pixel format would silently change every crop.

`write_tiff` writes such a file (tests, `dataset/synthetic_tiff.py`, tools/feed_rate.py): the readers of this package are exercised on real files."""
import numpy as np

_BACKEND = None


def backend():
    """'rasterio' or 'PIL' (checked once)."""
    global _BACKEND
    if _BACKEND is None:
        try:
            import rasterio  # noqa: F401
            _BACKEND = 'rasterio'
        except ImportError:
            try:
                from PIL import Image, features  # noqa: F401
            except ImportError as e:
                raise RuntimeError('the OpenEarthMap readers need rasterio or Pillow to decode GeoTIFF tiles; neither is installed '
                                   '(use --dataset synthetic_raw for decode-free synthetic tiles)') from e
            _BACKEND = 'PIL'
    return _BACKEND


def read_tiff(path):
    if backend() == 'rasterio':
        import rasterio
        with rasterio.open(path) as f:
            return f.read()
    from PIL import Image
    with Image.open(path) as im:
        if im.format != 'TIFF':
            raise RuntimeError('%s: not a TIFF file (%s)' % (path, im.format))
        if im.mode not in ('L', 'RGB', 'RGBA'):
            raise RuntimeError('%s: pixel format %r is not supported by the Pillow decoder (8-bit grey / RGB / RGBA tiles only; install rasterio for others)' % (path, im.mode))
        a = np.asarray(im)                                   # [H,W] or [H,W,bands], uint8
    if a.dtype != np.uint8:
        raise RuntimeError('%s: %s samples (8-bit tiles only)' % (path, a.dtype))
    return a[None] if a.ndim == 2 else np.moveaxis(a, 2, 0)


def write_tiff(path, arr, compression=None):
    """arr: uint8 [H,W] (label tile) or [H,W,3] (image tile).  compression: None, 'tiff_lzw', 'tiff_adobe_deflate' or 'packbits' (Pillow / libtiff names)."""
    from PIL import Image
    arr = np.ascontiguousarray(arr)
    if arr.dtype != np.uint8 or arr.ndim not in (2, 3) or (arr.ndim == 3 and arr.shape[2] != 3):
        raise ValueError('write_tiff: uint8 [H,W] or [H,W,3] arrays only')
    kw = {'compression': compression} if compression else {}
    Image.fromarray(arr, 'L' if arr.ndim == 2 else 'RGB').save(path, format='TIFF', **kw)
