"""Child processes of tests/test_round3_gpu.py::test_bucket_step_two_ranks_equals_ddp (not collected by pytest).

Two data-parallel ranks on the ONE GPU of the test box (gloo group: RCCL refuses two ranks per device), per-GPU BatchNorm statistics, each
rank on its half of a batch that changes every iteration.  mode 'ddp': Engine.data_parallel -> DistributedDataParallel with the in-place
bucket gradients, steps issued kernel by kernel (the round-2 path).  mode 'bucket': bucket_step.BucketedReplica + GraphedBucketStep -- graph A
(forward + backward into the build's own gradient buckets), one all-reduce per bucket, graph B (clip + both AdamW steps).  Same arithmetic in
the same order: the two modes must end with the same parameters.

    python tests/bucket2_child.py <rank> <port> <out.pt> <ddp | bucket>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, port, out_path, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, WORLD_SIZE='2', RANK=str(rank), LOCAL_RANK='0', SEGLAND_SYNC_BN='0')
    import torch
    import torch.distributed as dist
    import torch.nn as nn

    from oracle import formula as fm
    from segland_amd import bucket_step, graph_step
    from segland_amd.drivers import build_parser
    from segland_amd.engine import Engine
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    from segland_amd.optim import AdamW
    from segland_amd.train_base import train_iteration
    from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters

    dist.init_process_group('gloo', init_method='env://')        # before any GPU work of this process
    argv = ['--model', 'pspnet_pop', '--batch-size', '4'] + (['--no-step-graph'] if mode == 'ddp' else [])
    with Engine(custom_parser=build_parser(False), argv=argv) as engine:
        dev = engine.device
        B, H, W, iters = 4, 96, 128, 6
        torch.manual_seed(0)
        m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8,
                       norm_layer=nn.BatchNorm2d, compute_dtype=torch.float32)
        fm.load_formula_weights(m)
        if rank == 1:                                   # rank 1 starts from different weights: the replica must take rank 0's (broadcast at construction)
            with torch.no_grad():
                m.base_emb.add_(1.0)
        m = m.to(dev).train()
        opt = AdamW(get_parameters(m, lr=1e-4), lr=1e-4, weight_decay=1e-4)
        net = engine.data_parallel(m, sum_gradients=True, graphable=(mode == 'bucket'))
        assert engine.grad_div == 2
        if mode == 'bucket':
            assert isinstance(net, bucket_step.BucketedReplica) and len(net.buckets) >= 2
            step = bucket_step.GraphedBucketStep(net, opt, double_step=True, warmup=2)
        else:
            assert isinstance(net, nn.parallel.DistributedDataParallel)
            scaler = NativeScalerWithGradNormCount(engine.grad_div)
            step = lambda img, mask: train_iteration(net, opt, scaler, img, mask, double_step=True)      # noqa: E731
        losses = []
        for it in range(iters):
            img = fm.formula_image(B, H, W, 'b2/img%d' % it)[2 * rank:2 * rank + 2].to(dev)
            mask = fm.formula_mask(B, H, W, 8, 'b2/mask%d' % it, block=16, ignore_rows=0)[2 * rank:2 * rank + 2].to(dev)
            d, gn = step(img, mask)
            vals = engine.reduce_loss_dict(d)
            losses.append([float(vals['total_loss']), float(gn)])
        if mode == 'bucket':
            assert step.graph is not None and step.replays >= iters - 3, (step.failures, step.replays)
        m.eval()
        with torch.no_grad():
            logits = m(fm.formula_image(2, H, W, 'b2/eval').to(dev)).float().cpu()
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        # both ranks hold the same parameters (the buffers -- per-GPU BatchNorm statistics -- differ by design)
        flat = torch.cat([p.detach().flatten() for p in m.parameters()])
        both = [torch.zeros_like(flat) for _ in range(2)]
        dist.all_gather(both, flat)
        same = bool(torch.equal(both[0], both[1]))
        if rank == 0:
            torch.save({'sd': sd, 'losses': losses, 'logits': logits, 'ranks_equal': same,
                        'replays': graph_step.STATS['replays'], 'buckets': len(net.buckets) if mode == 'bucket' else 0}, out_path)
    print('BUCKET2_CHILD rank %d mode %s done' % (rank, mode), flush=True)


if __name__ == '__main__':
    main()
