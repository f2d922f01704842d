"""CPU tests of the tile decode and the OpenEarthMap readers on real files: dataset/tiff.py parses what it is asked to parse -- and
refuses what it would have to guess about -- and dataset/oem.py / oem_ft.py return exactly the arrays that were written (dataset/oem.py:55-76,
dataset/oem_ft.py:189-220 of the reference: rasterio.open(...).read() -> np.rollaxis(image, 0, 3), label[0])."""
import os
import random

import numpy as np
import pytest

from segland_amd.dataset import oem, oem_ft, synthetic_tiff, synthetic_tiff_ft, tiff


@pytest.mark.parametrize('compression', [None, 'tiff_lzw', 'tiff_adobe_deflate', 'packbits'])
def test_tiff_round_trip(tmp_path, compression):
    img, lab = synthetic_tiff.tile_arrays(5, (200, 176), 7, novel=True)
    pi, pl = str(tmp_path / 'i.tif'), str(tmp_path / 'l.tif')
    tiff.write_tiff(pi, img, compression)
    tiff.write_tiff(pl, lab, compression)
    a, b = tiff.read_tiff(pi), tiff.read_tiff(pl)
    assert a.dtype == np.uint8 and a.shape == (3, 200, 176) and b.shape == (1, 200, 176)          # bands first, like rasterio's read()
    assert np.array_equal(np.rollaxis(a, 0, 3), img) and np.array_equal(b[0], lab)


def test_tiff_decoder_refuses_what_it_cannot_represent(tmp_path):
    from PIL import Image
    p16 = str(tmp_path / 'u16.tif')
    Image.fromarray((np.arange(64 * 64, dtype=np.uint16).reshape(64, 64))).save(p16, format='TIFF')
    ppng = str(tmp_path / 'x.tif')
    Image.fromarray(np.zeros((8, 8), np.uint8)).save(ppng, format='PNG')
    if tiff.backend() == 'PIL':
        with pytest.raises(RuntimeError, match='pixel format|8-bit'):
            tiff.read_tiff(p16)
        with pytest.raises(RuntimeError, match='not a TIFF'):
            tiff.read_tiff(ppng)
    with pytest.raises(ValueError):
        tiff.write_tiff(str(tmp_path / 'f.tif'), np.zeros((4, 4), np.float32))


def test_oem_readers_on_tiff_files(tmp_path):
    root = synthetic_tiff.make_dataset(str(tmp_path / 'oem'), n=9, tile=(160, 144), seed=3, shot=2, compression='tiff_lzw', n_val=3)
    lst = os.path.join(root, 'list', 'train.txt')
    tr = oem.GFSSegTrain(root, lst, 0, crop_size=(128, 128))
    assert len(tr) == 9
    np.random.seed(1); random.seed(1)
    img, lab, prm, id_ = tr[4]
    wi, wl = synthetic_tiff.tile_arrays(4, (160, 144), 3, novel=False)
    assert id_ == 't0004' and img.flags['C_CONTIGUOUS'] and np.array_equal(img, wi) and np.array_equal(lab, wl)
    assert 0 <= prm[0] <= 32 and 0 <= prm[1] <= 16 and prm[3] in (0, 1, 2, 3)
    va = oem.GFSSegVal(root, os.path.join(root, 'list', 'val.txt'), 0, base_size=(160, 144))
    vi, vl, vp, vid = va[1]
    wi, wl = synthetic_tiff.tile_arrays(9 + 1, (160, 144), 3, novel=((9 + 1) % 3 == 2))
    assert vid == 'v0001' and np.array_equal(vi, wi) and np.array_equal(vl, wl) and vp == (0, 0, False, 0)
    os.remove(os.path.join(root, 'labels', 'v0002.tif'))                     # unlabeled test tile (eval_base.py:178-191)
    assert va[2][1] is None
    # the fine-tune pair reader: class lists built by reading every label file, cached next to the list like the reference does
    random.seed(5); np.random.seed(5)
    ft = oem_ft.GFSSegTrain(root, lst, 0, shot=2, crop_size=(128, 128), seed=3)
    assert len(ft) == 7 * 2 and os.path.exists(os.path.join(root, 'list', 'train_base_class1.txt'))
    (nov, base), (pn, pb), nid = ft[0]
    k = int(nid[1:])
    assert k % 3 == 2 and np.array_equal(nov[0], synthetic_tiff.tile_arrays(k, (160, 144), 3, True)[0])
    assert 0 not in np.unique(nov[1]) and 255 in np.unique(nov[1])           # oem_ft.py:197: unlabeled pixels of the novel tile become ignore
    kb = int(ft.base_id_list[0][1:])
    assert np.array_equal(base[1], synthetic_tiff.tile_arrays(kb, (160, 144), 3, kb % 3 == 2)[1])


def test_synthetic_tiff_datasets_resolve_for_the_drivers():
    from segland_amd import dataset as pkg
    from segland_amd.drivers import resolve
    assert resolve(pkg, 'synthetic_tiff') is synthetic_tiff and resolve(pkg, 'synthetic_tiff_ft') is synthetic_tiff_ft
    ds = synthetic_tiff.GFSSegTrain(crop_size=(64, 64), length=6)
    assert len(ds) == 6 and ds[0][0].shape == (160, 128, 3) and ds.raw_tiles
    f = synthetic_tiff_ft.GFSSegTrain(crop_size=(64, 64), length=12, shot=1)
    assert len(f) == 7 and f.pair_tiles


def test_collates_pack_and_crop_rows():
    """RawCollate / PairCollate (they run in the DataLoader workers): tiles cut down to the rows their crops read, the batch in ONE uint8 buffer; PackedTiles hands back
    views that equal the arrays that went in, and the draws' h_off becomes 0 exactly where rows were dropped."""
    import torch
    from segland_amd.dataset.oem import PackedTiles, RawCollate, crop_rows
    from segland_amd.dataset.oem_ft import PairCollate
    rng = np.random.RandomState(0)
    tiles = [(rng.randint(0, 256, (h, w, 3)).astype(np.uint8), rng.randint(0, 12, (h, w)).astype(np.uint8)) for h, w in ((96, 80), (64, 70), (40, 33), (65, 64))]
    prm = [(17, 5, True, 1), (0, 3, False, 0), (0, 0, True, 2), (1, 0, False, 3)]
    img, lab, p = crop_rows(tiles[0][0], tiles[0][1], prm[0], 64)
    assert img.shape == (64, 80, 3) and p == (0, 5, True, 1) and np.array_equal(img, tiles[0][0][17:81]) and np.array_equal(lab, tiles[0][1][17:81])
    assert crop_rows(tiles[2][0], tiles[2][1], prm[2], 64)[0].shape[0] == 40                    # smaller than the crop: padded later on the GPU, nothing dropped here
    assert crop_rows(tiles[1][0], None, prm[1], 64)[1] is None
    batch = [(t[0], t[1], q, 'id%d' % i) for i, (t, q) in enumerate(zip(tiles, prm))]
    packed, params, ids = RawCollate(64)(batch)
    assert isinstance(packed, PackedTiles) and len(packed) == 4 and ids == ['id0', 'id1', 'id2', 'id3'] and packed.buf.dtype == torch.uint8
    assert [q[0] for q in params] == [0, 0, 0, 0] and params[0][1:] == prm[0][1:]
    want = [tiles[0][0][17:81], tiles[1][0], tiles[2][0], tiles[3][0][1:65]]
    for (im, lb), w in zip(packed, want):
        assert np.array_equal(im.numpy(), w) and lb.shape == w.shape[:2]
    assert all(m[0] % 16 == 0 and m[3] % 16 == 0 for m in packed.meta)                          # 16-byte aligned tiles inside the buffer
    full, params_full, _ = RawCollate()(batch)                                                  # validation readers: whole tiles
    assert np.array_equal(full[0][0].numpy(), tiles[0][0]) and params_full[0] == prm[0]
    unl, _, _ = RawCollate()([(tiles[0][0], None, prm[1], 'u')])                                # unlabeled test tile
    assert unl[0][1] is None and unl.meta[0][3] == -1
    pairs, pp, pid = PairCollate(64)([(((tiles[0]), (tiles[3])), (prm[0], prm[3]), 'n0'), (((tiles[1]), (tiles[2])), (prm[1], prm[2]), 'n1')])
    assert len(pairs) == 4 and pid == ['n0', 'n1'] and pp[0] == ((0, 5, True, 1), (0, 0, False, 3))
    assert np.array_equal(pairs[1][0].numpy(), tiles[3][0][1:65]) and np.array_equal(pairs[2][1].numpy(), tiles[1][1])
    import pickle
    p2 = pickle.loads(pickle.dumps(packed))                                                     # what crosses the process boundary
    assert np.array_equal(p2[3][0].numpy(), want[3]) and p2.meta == packed.meta


def test_loader_workers_context():
    from segland_amd.engine import worker_context
    assert worker_context(0, True) is None and worker_context(4, False) is None
    ctx = worker_context(2, True)
    assert ctx.get_start_method() == 'forkserver' and worker_context(3, True) is ctx
