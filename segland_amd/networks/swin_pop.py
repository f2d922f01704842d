"""Swin-POP on MI355X: drop-in for networks/swin_pop.py of LiZhuoHong/SegLand (BASELINE config 5).

Same constructor, module tree and parameter names / shapes as the reference (`backbone.*` Swin-T/S/B/L, `decoder.psp.*`,
`decoder.lateral_convs.*`, `decoder.fpn_convs.*`, `classifier`, `classifier_n`, `base_emb`, `novel_emb`; d_model = backbone.get_filters()[0]),
same forward dispatch / return conventions / side effects as pspnet_pop (swin_pop.py:266-386 is that code on another feature extractor), and
`train_mode(backbone_only=False)` (swin_pop.py:220-228).  All arithmetic runs in libsegland_hip.so: segland_amd.functional_swin for the
backbone and the UperNet_Decoder_Plus, the POP head / loss kernels of the PSPNet path unchanged (at 128 channels = d_model 96 + zero pad).
"""
import types

import torch
import torch.nn as nn
from torch.nn import functional as F

from ..functional import PopHeadFn, flush_num_batches_tracked
from ..functional_swin import AddResizedFn, ConvBnReluFn, PspSwinFn, ResizeFn, psp_params
from ..ops_swin import pad_to
from . import pspnet_pop
from .backbones import get_backbone


class PSPModule(nn.Module):
    """swin_pop.py:7-35 (1x1 bottleneck, align_corners=True priors, Dropout2d(0.1)); parameter holder."""

    def __init__(self, features, out_features=512, sizes=(1, 2, 3, 6), norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.sizes = tuple(sizes)
        self.stages = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(output_size=(s, s)), nn.Conv2d(features, out_features, kernel_size=1, bias=False),
                                                   norm_layer(out_features), nn.ReLU()) for s in sizes])
        self.bottleneck = nn.Sequential(nn.Conv2d(features + len(sizes) * out_features, out_features, kernel_size=1, padding=0, dilation=1, bias=False),
                                        norm_layer(out_features), nn.ReLU(), nn.Dropout2d(0.1))


def _cbr(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class UperNet_Decoder_Plus(nn.Module):
    """swin_pop.py:104-173.  forward(four NHWC maps) -> [B, H/4, W/4, pad_to(dim)]."""

    def __init__(self, filters, dim=512, ppm_size=(1, 2, 3, 6)):
        super().__init__()
        self.dim = dim
        self.psp = PSPModule(filters[-1], dim, sizes=ppm_size)
        self.lateral_convs = nn.ModuleList([_cbr(c, dim) for c in filters[:-1]])
        self.fpn_convs = nn.ModuleList()
        for c in filters:
            n = max(1, int(torch.log2(torch.tensor(c)) - torch.log2(torch.tensor(filters[0]))))      # swin_pop.py:120-122
            head = []
            for _ in range(n):
                head.append(_cbr(dim, dim))
                if c != filters[0]:
                    head.append(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True))
            self.fpn_convs.append(nn.Sequential(*head))
        self.dropout2d_hook = None            # (B, C, p) -> [B, C] scale tensor | None: parity tests feed the oracle's mask

    def _dropout_scale(self, B, device):
        p, Cn = self.psp.bottleneck[3].p, self.dim
        if self.dropout2d_hook is not None:
            m = self.dropout2d_hook(B, Cn, p)
            return None if m is None else m.to(device=device, dtype=torch.float32).contiguous()
        if not self.psp.bottleneck[3].training or p <= 0.0:
            return None
        # ATen's feature_dropout (F.dropout2d): noise.bernoulli_(1 - p).div_(1 - p) -- two launches, the reference's own draw
        return torch.empty(B, Cn, dtype=torch.float32, device=device).bernoulli_(1.0 - p).div_(1.0 - p)

    @staticmethod
    def _cbr_apply(seq, x):
        return ConvBnReluFn.apply(x, seq, seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias)

    def forward(self, xs):
        lat = [self._cbr_apply(seq, x) for seq, x in zip(self.lateral_convs, xs[:-1])]
        lat.append(PspSwinFn.apply(xs[-1], self.psp, self._dropout_scale(xs[-1].shape[0], xs[-1].device), *psp_params(self.psp)))
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = AddResizedFn.apply(lat[i - 1], lat[i], True)
        size = tuple(xs[0].shape[1:3])
        out = None
        for head, f in zip(self.fpn_convs, lat):
            mods = list(head)
            for k, m in enumerate(mods):
                if isinstance(m, nn.Upsample):
                    up = (2 * f.shape[1], 2 * f.shape[2])
                    if k == len(mods) - 1 and out is not None and up == size:
                        break                      # the head's last x2 upsample lands on the output grid: fused with the sum below (same taps)
                    f = ResizeFn.apply(f, up, True)
                else:
                    f = self._cbr_apply(m, f)
            if out is None:
                out = f if tuple(f.shape[1:3]) == size else ResizeFn.apply(f, size, True)
            else:
                out = AddResizedFn.apply(out, f, True)
        return out


class _ConvView:
    """What functional.spec_of / prepared need from a 1x1 conv whose weight is a zero-padded copy of the parameter."""
    kernel_size, stride, padding, dilation = (1, 1), (1, 1), (0, 0), (1, 1)

    def __init__(self, weight):
        self.weight, self.out_channels, self.in_channels = weight, weight.shape[0], weight.shape[1]


class GFSS_Model(pspnet_pop.GFSS_Model):
    """Segmenter for Generalized Few-shot Semantic Segmentation (networks/swin_pop.py:175-386)."""

    def __init__(self, n_base, criterion=None, norm_layer=nn.BatchNorm2d, use_base=True, is_ft=False, n_novel=0,
                 compute_dtype=torch.bfloat16, **kwargs):
        nn.Module.__init__(self)
        self.backbone = get_backbone(norm_layer=norm_layer, compute_dtype=compute_dtype, **kwargs)
        d_model = self.backbone.get_filters()[0]
        self.decoder = UperNet_Decoder_Plus(self.backbone.get_filters(), d_model)
        self.classifier = pspnet_pop._classifier(d_model)
        if is_ft:
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=False)
            self.novel_emb = nn.Parameter(torch.zeros(n_novel, d_model), requires_grad=True)
            self.classifier_n = pspnet_pop._classifier(d_model)
            nn.init.orthogonal_(self.novel_emb)
            self.ft_freeze()
        else:
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=True)
            nn.init.orthogonal_(self.base_emb)
            self.novel_emb = None
        self.n_novel, self.use_base, self.is_ft = n_novel, use_base, is_ft
        self.criterion, self.n_base = criterion, n_base
        self.compute_dtype, self.d_model = compute_dtype, d_model
        print('n_novel:', n_novel)

    def train_mode(self, backbone_only=False):
        self.train()
        self.backbone.eval()       # swin_pop.py:223: no DropPath; LayerNorm has no running state
        if not backbone_only:
            self.decoder.eval()
            for p in self.decoder.parameters():
                p.requires_grad = False

    def _features_eager(self, img):
        """Backbone + decoder as an eager kernel sequence.  pspnet_pop.GFSS_Model._features (inherited) replays it from a HIP graph when everything
        in it is frozen and in eval mode (ft_pop training, no-grad evaluation): ~600 launches of a few microseconds each are launch-bound otherwise."""
        if not img.is_cuda:
            raise RuntimeError('segland_amd.GFSS_Model runs on the GPU only (no CPU fallback): move the model and inputs to cuda')
        from .. import functional_swin as fs
        plan = fs.model_plan(self)
        plan.refresh()                      # every GEMM weight whose parameter changed since the last step: one batched launch
        fs.CURRENT_PLAN[0] = plan
        try:
            self.__dict__['_sl_cut'] = None
            feats = self.backbone(img)
            if self.__dict__.get('_sl_want_cut') and all(f.requires_grad for f in feats):
                leaves = tuple(f.detach().requires_grad_(True) for f in feats)      # the graph is cut here (see resnet.py base_forward)
                self.__dict__['_sl_cut'] = list(zip(feats, leaves))
                feats = leaves
            feat = self.decoder(feats)
        finally:
            fs.CURRENT_PLAN[0] = None
        flush_num_batches_tracked()
        return feat

    # two-part backward (bucket_step.BucketedReplica): the cut is between the Swin backbone and the UperNet decoder -- the decoder's and the head's gradients are
    # complete when the backward reaches the backbone's four feature maps
    bucket_cut_default = False         # measured: the cut costs a Swin-T step 0.8 ms at world size 1 and would hide only the decoder's share of a ~0.8 ms all-reduce (SEGLAND_BUCKET_CUT=1 enables it)

    def enable_backward_cut(self, flag):
        self.__dict__['_sl_want_cut'] = bool(flag)

    def late_parameters(self):
        early = {id(p) for p in self.backbone.parameters()}
        return [p for p in self.parameters() if id(p) not in early]

    def cut_tensors(self):
        return self.__dict__.get('_sl_cut')

    def _padded_cls(self, cls, P):
        d = self.d_model
        return [_ConvView(F.pad(cls[0].weight, (0, 0, 0, 0, 0, P - d, 0, P - d))), None, _ConvView(F.pad(cls[2].weight, (0, 0, 0, 0, 0, P - d, 0, P - d))), None,
                _ConvView(F.pad(cls[4].weight, (0, 0, 0, 0, 0, P - d, 0, 0)))]

    def _head(self, feat):
        P, d = feat.shape[-1], self.d_model
        sb0, sn0, orth = self._protos()                    # normalised prototypes + orthogonality term (one kernel, pspnet_pop.GFSS_Model._protos)
        sb = F.pad(sb0, (0, P - d))
        cls = self._padded_cls(self.classifier, P)
        pc = [cls[0].weight, cls[2].weight, cls[4].weight]
        if self.is_ft:
            sn = F.pad(sn0, (0, P - d))
            cls_n = self._padded_cls(self.classifier_n, P)
            holder = types.SimpleNamespace(classifier=cls, classifier_n=cls_n)
            preds = PopHeadFn.apply(feat, sb, sn, holder, *pc, cls_n[0].weight, cls_n[2].weight, cls_n[4].weight)
            return preds, sb0, sn0, orth
        holder = types.SimpleNamespace(classifier=cls, classifier_n=None)
        return PopHeadFn.apply(feat, sb, None, holder, *pc), sb0, None, orth
