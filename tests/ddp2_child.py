"""Child processes of tests/test_round2_gpu.py::test_two_ranks_equal_one_full_batch (not collected by pytest).

SURVEY.md 8(e) "equivalence test": a 2-rank data-parallel step on a split batch equals the 1-process step on the full batch.  Both ranks
run on the ONE GPU of the test box (RCCL refuses two ranks per device, so the group is gloo, which moves CUDA tensors through the host);
everything else is the product path: segland_amd.GFSS_Model on the HIP kernels under Engine.data_parallel -- since round 4 the build's own
bucket_step.BucketedReplica also with SyncBatchNorm (flat gradient buckets written in place, one SUM all-reduce per bucket around the two backward halves, the step
issued kernel by kernel because SyncBatchNorm's collectives sit inside it) --, nn.SyncBatchNorm with SEGLAND_SYNC_BN=1 (global batch statistics: per-rank partial
sums all-reduced in fp64), segland_amd.optim.AdamW with the 1 / world_size in its kernel.

    python tests/ddp2_child.py <rank> <port> <out.pt>        rank 0/1: the 2-rank run;   rank -1: the single-process full-batch run
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, port, out_path = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    two = rank >= 0
    if two:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, WORLD_SIZE='2', RANK=str(rank), LOCAL_RANK='0', SEGLAND_SYNC_BN='1')
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

    if two:
        dist.init_process_group('gloo', init_method='env://')        # before any GPU work of this process
    with Engine(custom_parser=build_parser(False), argv=['--model', 'pspnet_pop', '--batch-size', '4', '--no-step-graph']) as engine:
        dev = engine.device
        assert engine.distributed == two
        B, H, W = 4, 96, 128
        img = fm.formula_image(B, H, W, 'ddp2/img')
        mask = fm.formula_mask(B, H, W, 8, 'ddp2/mask', block=16, ignore_rows=0)      # no ignored pixels: per-rank means average to the global mean
        if two:
            img, mask = img[2 * rank:2 * rank + 2], mask[2 * rank:2 * rank + 2]
            sf.set_sync_bn('1')
        img, mask = img.to(dev), mask.to(dev)
        torch.manual_seed(0)
        m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8,
                       norm_layer=nn.SyncBatchNorm if two else nn.BatchNorm2d, compute_dtype=torch.float32)
        fm.load_formula_weights(m)
        m = m.to(dev).train()
        opt = AdamW(get_parameters(m, lr=1e-4), lr=1e-4, weight_decay=1e-4)
        net = engine.data_parallel(m, sum_gradients=two, graphable=two)
        scaler = NativeScalerWithGradNormCount(engine.grad_div)
        step = None
        if two:
            from segland_amd import bucket_step
            assert isinstance(net, bucket_step.BucketedReplica) and engine.grad_div == 2
            step = bucket_step.GraphedBucketStep(net, opt, double_step=True)
            assert step.eager_reason is not None and 'SyncBatchNorm' in step.eager_reason
        # gradients of ONE backward before any optimizer step: the quantity the equivalence is exact for (up to the order of the sums);
        # Adam's first steps are ~ lr * sign(g), which turns last-bit differences of near-zero gradient entries into O(lr) parameter differences
        opt.zero_grad()
        if two:                                  # one forward / backward of the replica's step, cut at the all-reduces like BucketedReplica._run, without the update
            parts = net.train_step_parts(opt, True)
            d0 = parts[0](img, mask)
            halves = net.cut and net.late_buckets < len(net.buckets)
            works = net.all_reduce('late' if halves else None, async_op=True)
            if halves:
                parts[1]()
                works += net.all_reduce('early', async_op=True)
            for w in works:
                w.wait()
        else:
            d0 = net(img, mask)
            d0['total_loss'].backward()
        grads0 = {n: p.grad.detach().float().cpu() / engine.grad_div for n, p in m.named_parameters() if p.grad is not None}
        loss0 = float(engine.reduce_loss_dict(d0)['total_loss']) if two else float(d0['total_loss'].detach())
        stats0 = {k: v.detach().float().cpu() for k, v in m.state_dict().items() if 'running_' in k}
        del d0
        # (1b) TWO iterations of the same step machinery with a plain SGD update (linear in the gradient: no sign-like first steps), then everything is restored.
        # The summed update of the two steps must agree between the configurations to the gradient tolerance: a 1 / world_size (or a stale bucket, or a
        # double all-reduce) that slips in at the SECOND step would be a factor, not a rounding difference.
        class PlainSGD(torch.optim.Optimizer):
            """theta -= lr * repeat * grad_scale * grad, with segland_amd.optim.AdamW's step() signature (the clip coefficient carries the 1 / world_size)."""
            def __init__(self, params, lr):
                super().__init__(params, dict(lr=lr))
                self.repeat_next = 1

            @torch.no_grad()
            def step(self, closure=None, repeat=None, grad_scale=None):
                repeat = self.repeat_next if repeat is None else repeat
                self.repeat_next = 1
                for g in self.param_groups:
                    for p_ in g['params']:
                        if p_.grad is not None:
                            p_.sub_(p_.grad * (grad_scale if grad_scale is not None else 1.0) * (g['lr'] * repeat))
        saved = {k: v.detach().clone() for k, v in m.state_dict().items()}
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        sgd = PlainSGD(get_parameters(m, lr=1e-3), lr=1e-3)
        sgd_step = bucket_step.GraphedBucketStep(net, sgd, double_step=True) if two else None
        sgd_losses = []
        for it in range(2):
            d, gn = sgd_step(img, mask) if sgd_step is not None else train_iteration(net, sgd, scaler, img, mask, double_step=True)
            vals = engine.reduce_loss_dict(d) if two else {k: float(v) for k, v in d.items()}
            sgd_losses.append([float(vals['total_loss']), float(gn)])
        sgd_update = {n: (p.detach() - before[n]).float().cpu() for n, p in m.named_parameters() if p.requires_grad}
        m.load_state_dict(saved)
        for p_ in m.parameters():
            p_.grad = None
        del saved, before, sgd, sgd_step
        losses = []
        for it in range(3):
            d, gn = step(img, mask) if step is not None else train_iteration(net, opt, scaler, img, mask, double_step=True)
            vals = engine.reduce_loss_dict(d) if two else {k: float(v) for k, v in d.items()}
            losses.append([float(vals['total_loss']), float(gn)])
        m.eval()
        with torch.no_grad():
            logits = m(fm.formula_image(2, H, W, 'ddp2/eval').to(dev)).float().cpu()
        if rank <= 0:
            torch.save({'grads0': grads0, 'loss0': loss0, 'stats0': stats0, 'sgd_update': sgd_update, 'sgd_losses': sgd_losses, 'sd': {k: v.detach().float().cpu() for k, v in m.state_dict().items()}, 'losses': losses, 'logits': logits}, out_path)
    print('DDP2_CHILD rank %d done' % rank, flush=True)


if __name__ == '__main__':
    main()
