"""Child process of tests/test_round2_gpu.py::test_hip_model_under_rccl_ddp (not collected by pytest).

world_size-1 RCCL run of the PRODUCT: segland_amd.GFSS_Model on the HIP kernels, wrapped by Engine.data_parallel
(DistributedDataParallel, gradient_as_bucket_view, engine.py:71 of the reference), stepped by segland_amd.optim.AdamW with the
train_base.py:250-264 loop body.  The process group is created before any GPU work of this process.  Prints one JSON line.

    python tests/ddp_child.py <mode: 0 | force (SyncBN semantics) | inplace (sum-only all-reduce + gradients written into the bucket views)
                               | bucket (bucket_step.BucketedReplica: two HIP graphs around RCCL all-reduces of the build's own buckets)
                               | bucket_force (the same replica with SyncBatchNorm semantics: its step is issued kernel by kernel)> <port>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sync, port = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', SEGLAND_FORCE_DDP='1',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch
    import torch.distributed as dist
    import torch.nn as nn

    from oracle import formula as fm
    from segland_amd import functional as sf
    from segland_amd.drivers import build_parser
    from segland_amd.engine import Engine
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    from segland_amd.optim import AdamW
    from segland_amd.train_base import train_iteration
    from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters

    out = {}
    with Engine(custom_parser=build_parser(False), argv=['--model', 'pspnet_pop', '--batch-size', '4']) as engine:
        assert engine.distributed and dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
        dev = engine.device
        img = fm.formula_image(4, 128, 128, 'ddp1/img').to(dev)
        mask = fm.formula_mask(4, 128, 128, 8, 'ddp1/mask', block=16, ignore_rows=6).to(dev)

        inplace = sync in ('inplace', 'bucket', 'bucket_force')
        bucket = sync in ('bucket', 'bucket_force')
        hits = [0]
        if inplace:
            real = sf.grad_dst

            def counted(p):
                v = real(p)
                hits[0] += v is not None
                return v
            sf.grad_dst = counted

        def run(wrapped, norm):
            torch.manual_seed(0)
            m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8,
                           norm_layer=norm, compute_dtype=torch.float32)
            fm.load_formula_weights(m)
            m = m.to(dev).train()
            opt = AdamW(get_parameters(m, lr=1e-4), lr=1e-4, weight_decay=1e-4)
            net = engine.data_parallel(m, sum_gradients=inplace, graphable=bucket) if wrapped else m
            step = None
            if wrapped and bucket:
                from segland_amd import bucket_step
                assert isinstance(net, bucket_step.BucketedReplica)
                step = bucket_step.GraphedBucketStep(net, opt, double_step=True, warmup=1)
            elif wrapped:
                assert isinstance(net, nn.parallel.DistributedDataParallel)
            scaler = NativeScalerWithGradNormCount(engine.grad_div if wrapped else 1)
            losses = []
            for it in range(5 if bucket else (3 if inplace else 2)):
                hits[0] = 0
                d, gn = step(img, mask) if step is not None else train_iteration(net, opt, scaler, img, mask, double_step=True)
                losses.append([float(d['total_loss'].detach()), float(gn)])
                print('ddp_child: wrapped %s iteration %d done' % (wrapped, it), file=sys.stderr, flush=True)
            if step is not None:
                out['bucket_replays'], out['bucket_failures'], out['buckets'], out['eager_reason'] = step.replays, step.failures, len(net.buckets), step.eager_reason
            if wrapped and inplace and not bucket:
                out['inplace_writes_last_step'] = hits[0]
                out['grads_alias_cached_views'] = sum(1 for p in m.parameters() if p.grad is not None and getattr(p, '_sl_gview', None) is not None
                                                      and p.grad.data_ptr() == p._sl_gview.data_ptr())
            if wrapped and bucket:
                out['grads_alias_cached_views'] = sum(1 for p in m.parameters() if p.grad is not None and p.grad.data_ptr() == net.views[id(p)].data_ptr())
            if wrapped:                        # gradients are views into DDP's flat buckets (gradient_as_bucket_view)
                bucket_views = sum(1 for p in m.parameters() if p.grad is not None and p.grad._base is not None)
                out['bucket_view_grads'] = bucket_views
            print('ddp_child: wrapped %s training done' % wrapped, file=sys.stderr, flush=True)
            m.eval()
            with torch.no_grad():
                logits = m(img).float().cpu()
            print('ddp_child: wrapped %s eval done' % wrapped, file=sys.stderr, flush=True)
            return {k: v.detach().float().cpu() for k, v in m.state_dict().items()}, losses, logits

        if sync in ('force', 'bucket_force'):
            # the comparison below is between a SyncBatchNorm run and a plain-BatchNorm run to 2e-4 after AdamW iterations: that needs the SAME kernels on both sides (Adam's
            # first steps turn last-bit differences of near-zero gradients into O(lr) parameter differences).  The one-launch backward of the pyramid stages' BatchNorms
            # (round 5) cannot serve SyncBatchNorm (a collective sits between its two sweeps), so the plain run takes the per-level chain here as well.
            sf._STAGE_BN_GROUPED = False
        sf.set_sync_bn('0')
        ref_sd, ref_losses, ref_logits = run(False, nn.BatchNorm2d)
        sf.set_sync_bn('force' if sync in ('force', 'bucket_force') else '0')
        sd, losses, logits = run(True, nn.SyncBatchNorm if sync in ('force', 'bucket_force') else nn.BatchNorm2d)
        sf.set_sync_bn('0')
        worst, worst_key = 0.0, ''
        for k in ref_sd:
            a, b = sd[k], ref_sd[k]
            e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))
            if e > worst:
                worst, worst_key = e, k
        out.update(sync=sync, worst_param_rel=worst, worst_key=worst_key, losses=losses, ref_losses=ref_losses,
                   logits_rel=float((logits - ref_logits).abs().max() / ref_logits.abs().max()),
                   n_params=len(ref_sd), bn1_tracked=int(sd['backbone.bn1.num_batches_tracked']), ref_bn1_tracked=int(ref_sd['backbone.bn1.num_batches_tracked']))
    print('DDP_CHILD ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
