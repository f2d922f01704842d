"""Shared by the CPU and GPU tests of golden G19: the PRODUCT pair reader (segland_amd/dataset/oem_ft.py PairReader: list logic, draws, pair rule)
over the synthetic tiles the golden was generated on, decoded from memory instead of GeoTIFF files."""
import os

from oracle import data_oracle as do


def product_reader(tmp_path, filt, shot=2, seed=123, crop=(64, 64), novel_ids=None):
    from segland_amd.dataset.oem_ft import PairReader
    ids, imgs, labs = do.ft_tiles()
    list_dir = os.path.join(str(tmp_path), 'list')
    for d in (list_dir, list_dir + '_filter'):
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, 'all_%dshot_seed%d.txt' % (shot, seed)), 'w').write(''.join(i + '\n' for i in novel_ids))
    lst = os.path.join(list_dir, 'train.txt')
    open(lst, 'w').write(''.join(i + '\n' for i in ids))

    class Reader(PairReader):
        num_classes = 11

        def __init__(self):
            self._init_lists(lst, shot, 'train', crop, 255, seed, filt, True)

        def read_image(self, id_):
            return imgs[id_]

        def read_label(self, id_):
            return labs[id_]
    return Reader
