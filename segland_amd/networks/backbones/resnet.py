"""Dilated ResNet-50/101 backbone on the HIP kernels (drop-in for networks/backbones/resnet.py of the reference).

Same parameter names / shapes as the reference (`conv1`, `bn1`, `layer{1..4}.{i}.conv{1,2,3}`, `.bn{1,2,3}`,
`.downsample.{0,1}`), so reference checkpoints load unchanged.  The modules only hold parameters; arithmetic is in
segland_amd.functional (StemFn, BottleneckFn).  Activations between blocks are NHWC in the compute dtype.
"""
import torch.nn as nn

from ...functional import BottleneckFn, StemFn, bottleneck_params


class Bottleneck(nn.Module):
    """networks/backbones/resnet.py:40-78.  forward(x_nhwc) -> y_nhwc."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, multi_grid=1, norm_layer=nn.BatchNorm2d,
                 last_relu=True):
        super().__init__()
        d = dilation * multi_grid
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=d, dilation=d, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.downsample = downsample
        self.dilation, self.stride, self.last_relu = dilation, stride, last_relu

    def forward(self, x):
        return BottleneckFn.apply(x, self, *bottleneck_params(self))


class ResNet(nn.Module):
    """networks/backbones/resnet.py:80-131 (Bottleneck variant).  base_forward(img NCHW float) -> x4 NHWC."""

    def __init__(self, block, layers, norm_layer=nn.BatchNorm2d, dilated=True, multi_grid=False, os=8, relu_l3=True,
                 relu_l4=True, compute_dtype=None, **kwargs):
        super().__init__()
        import torch
        self.inplanes = 64
        self.deep_channels, self.dsn_channels = 2048, 1024
        self.compute_dtype = compute_dtype or torch.bfloat16
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.layer1 = self._make_layer(block, 64, layers[0], norm_layer=norm_layer)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, norm_layer=norm_layer)
        grid = (1, 2, 4) if multi_grid else (1, 1, 1)
        if dilated and os == 8:
            self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2, norm_layer=norm_layer, last_relu=relu_l3)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4, multi_grid=grid, norm_layer=norm_layer, last_relu=relu_l4)
        elif dilated:
            self.layer3 = self._make_layer(block, 256, layers[2], stride=2, norm_layer=norm_layer, last_relu=relu_l3)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=2, multi_grid=grid, norm_layer=norm_layer, last_relu=relu_l4)
        else:
            self.layer3 = self._make_layer(block, 256, layers[2], stride=2, norm_layer=norm_layer, last_relu=relu_l3)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=2, norm_layer=norm_layer, last_relu=relu_l4)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, multi_grid=1, norm_layer=nn.BatchNorm2d, last_relu=True):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                               norm_layer(planes * block.expansion))
        mg = (lambda i: multi_grid[i % len(multi_grid)]) if isinstance(multi_grid, tuple) else (lambda i: 1)
        seq = [block(self.inplanes, planes, stride, dilation=dilation, downsample=ds, multi_grid=mg(0), norm_layer=norm_layer)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            seq.append(block(self.inplanes, planes, dilation=dilation, multi_grid=mg(i), norm_layer=norm_layer,
                             last_relu=True if i != blocks - 1 else last_relu))
        return nn.Sequential(*seq)

    def forward_base_in(self, img):
        return StemFn.apply(img.float().contiguous(), self.conv1.weight, self.bn1.weight, self.bn1.bias, self, self.compute_dtype)

    def base_forward(self, img):
        self.__dict__['_sl_cut'] = None                # first: the stashed tensors hold the previous step's autograd graph (and its AccumulateGrad nodes) alive
        x = self.forward_base_in(img)
        prev = None
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                blk.__dict__['_sl_prev'] = prev       # the block whose output is this block's ONLY input: its bn3 backward statistics can ride on this block's conv1 data gradient
                x = blk(x)
                prev = blk
            if stage is self.layer3 and self.__dict__.get('_sl_want_cut') and x.requires_grad:
                # GFSS_Model.cut_tensors(): where a data-parallel backward is cut in two (bucket_step.py).  The graph is really cut: layer4 continues on a
                # detached leaf whose .grad the first half fills; the second half restarts from (x, that gradient).  (A non-leaf tensor in backward(inputs=...)
                # makes autograd EXECUTE its producing node, and a second pass through it would find its saved tensors freed.)
                leaf = x.detach().requires_grad_(True)
                self.__dict__['_sl_cut'] = (x, leaf)
                x = leaf
        return x
