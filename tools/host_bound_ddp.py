import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29544', RANK='0', WORLD_SIZE='1')
import torch, torch.distributed as dist
import bench
from segland_amd.loss.criterion import OrthLoss
from segland_amd.networks.pspnet_pop import GFSS_Model
mode = sys.argv[1]; B = int(sys.argv[2])
torch.cuda.set_device(0)
if mode != 'plain': dist.init_process_group('nccl', init_method='env://')
torch.manual_seed(0); dev = torch.device('cuda', 0)
model = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8).to(dev).train()
opt = bench.make_optimizer(model)
net, gd = model, 1
if mode != 'plain':
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=64)
    if mode == 'inplace':
        from segland_amd.engine import enable_inplace_bucket_gradients
        enable_inplace_bucket_gradients(net)
    if mode == 'builtin':        # C++ all-reduce hook (averages), bucket views cached like the in-place mode
        from segland_amd import engine as _e
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        net._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
        register_optimizer_step_pre_hook(_e._cache_bucket_views)
params = [p for p in model.parameters() if p.requires_grad]
variant = sys.argv[3] if len(sys.argv) > 3 else ''
batches = [bench.synthetic_batch(B, 512, dev, seed=k) for k in range(4 if 'b' in variant else 1)]
for i in range(10): bench.train_step(net, opt, *batches[i % len(batches)], params, True, gd)
torch.cuda.synchronize()
K = 50; marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
t0 = time.perf_counter()
for i in range(K):
    bench.train_step(net, opt, *batches[i % len(batches)], params, True, gd)
    if 'e' in variant: marks[i + 1].record()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(variant, '%s batch %d: enqueue %.2f ms/step, complete %.2f ms/step' % (mode, B, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K))
if mode != 'plain': dist.destroy_process_group()
