#!/usr/bin/env python3
"""Multi-model fusion of probability maps (counterpart of the reference's fusemat.py:35-52) with the mean + argmax on the GPU.

    python -m segland_amd.fusemat --inputs run_a/prob run_b/prob run_c/prob --output fused [--size 1024,1024]

Every input directory holds the per-tile `<id>.mat` dumps of `eval_base --save-prob` ({'outputs': [1,K,H,W] upsampled logits}, eval_base.py:189-190).
Tiles are matched by file name; their maps are summed in directory order and the argmax of the mean is written as a palette PNG, resized with
nearest neighbour to --size (fusemat.py:47-52).  Reading .mat (scipy) and writing PNG (PIL) stay on the host."""
import argparse
import os

import numpy as np
import torch

COLORMAP = np.array([[147, 147, 147], [49, 139, 87], [0, 255, 0], [128, 0, 0], [75, 181, 73], [245, 245, 245], [35, 91, 200], [247, 142, 82]], dtype=np.uint8)   # fusemat.py:20-27


def collect(fusion_list):
    """{file name: [map of model 0, map of model 1, ...]} in the reference's walk order (fusemat.py:36-46)."""
    import scipy.io
    fns, mats = [], []
    for path in fusion_list:
        for root, _, files in os.walk(path):
            for f in sorted(files):
                if not f.endswith('.mat'):
                    continue
                prob = scipy.io.loadmat(os.path.join(root, f))['outputs'][0]
                if f not in fns:
                    fns.append(f); mats.append([prob])
                else:
                    mats[fns.index(f)].append(prob)
    return fns, mats


def fuse_tile(maps, device='cuda'):
    from . import ops
    return ops.fuse_argmax([torch.as_tensor(np.ascontiguousarray(m, dtype=np.float32)).to(device) for m in maps])


def fuse(fusion_list, output_path, size=(1024, 1024), n_models=None, device='cuda'):
    from PIL import Image
    os.makedirs(output_path, exist_ok=True)
    fns, mats = collect(fusion_list)
    n = len(fusion_list) if n_models is None else n_models
    out = {}
    for fn, maps in zip(fns, mats):
        if len(maps) != n:
            # the reference divides by len(fusion_list) whatever the number of maps found; the argmax is invariant to that positive factor
            pass
        lab = fuse_tile(maps, device).cpu().numpy()
        img = Image.fromarray(lab, 'P').resize((size[1], size[0]), Image.NEAREST)
        img.putpalette(COLORMAP)
        img.save(os.path.join(output_path, fn.split('.')[0] + '.png'))
        out[fn] = lab
    return out


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    p.add_argument('--inputs', nargs='+', required=True)
    p.add_argument('--output', required=True)
    p.add_argument('--size', default='1024,1024')
    a = p.parse_args(argv)
    fuse(a.inputs, a.output, tuple(map(int, a.size.split(','))))


if __name__ == '__main__':
    main()
