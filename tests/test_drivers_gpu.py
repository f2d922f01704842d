"""The two entry points run end to end on the GPU with the synthetic dataset (same flags as scripts/train_oem.sh /
scripts/ft_oem.sh of the reference, only the dataset swapped), G8 loss trajectory, and IoU through the HIP kernels."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import formula as fm

pytestmark = pytest.mark.gpu


def test_train_base_entry_point(hip, tmp_path):
    from segland_amd import train_base
    snap = str(tmp_path / 'snap')
    train_base.main(['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic', '--batch-size', '4', '--input-size', '128,128',
                     '--base-size', '128,128', '--num-epoch', '1', '--learning-rate', '1e-4', '--print-frequency', '4', '--snapshot-dir', snap,
                     '--num-workers', '0', '--restore-from', '/nonexistent', '--fp16'])
    ck = glob.glob(os.path.join(snap, 'epoch_1.pth'))
    assert ck, 'no checkpoint written'
    sd = torch.load(ck[0], map_location='cpu')
    assert 'module.base_emb' in sd and all(torch.isfinite(v.float()).all() for v in sd.values())
    assert int(sd['module.backbone.bn1.num_batches_tracked']) == 16          # 64 tiles / batch 4


def test_ft_pop_entry_point(hip, tmp_path):
    from segland_amd import ft_pop
    snap = str(tmp_path / 'snap_ft')
    ft_pop.main(['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic', '--batch-size', '1', '--input-size', '128,128',
                 '--base-size', '128,128', '--num-epoch', '1', '--learning-rate', '1e-3', '--print-frequency', '5', '--snapshot-dir', snap,
                 '--num-workers', '0', '--restore-from', '/nonexistent', '--random-seed', '123', '--freeze-backbone', '--fix-bn', '--update-base'])
    assert glob.glob(os.path.join(snap, 'epoch_0_123.pth'))


def test_g8_loss_trajectory_fp32(hip):
    """3 iterations of the train_base.py loop body (with the reference's double AdamW step) from the formula weights.
    Step 0 is a pure forward/backward parity check; later steps amplify rounding differences through the chaotic
    B=2 train-mode network (numerics), hence the growing tolerance."""
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    from segland_amd.train_base import train_iteration
    from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters
    g = golden('g8_traj')
    m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=torch.float32)
    fm.load_formula_weights(m)
    m = m.cuda().train()
    opt = torch.optim.AdamW(get_parameters(m, lr=1e-5), lr=1e-5, weight_decay=1e-4)
    img = fm.formula_image(2, 512, 512, 'g6/img').cuda(); mask = fm.formula_mask(2, 512, 512, 8, 'g6/mask').cuda()
    scaler = NativeScalerWithGradNormCount()
    for step, tol in enumerate([2e-4, 2e-2, 5e-2]):
        d, norm = train_iteration(m, opt, scaler, img, mask, double_step=True)
        got = [float(d['total_loss']), float(d['seg_loss']), float(d['orth_loss'])]
        np.testing.assert_allclose(got, g['losses'][step], rtol=tol, atol=1e-5)
        np.testing.assert_allclose(float(norm), g['norms'][step], rtol=max(tol * 10, 5e-3))
    assert int(m.backbone.bn1.num_batches_tracked) == 3
