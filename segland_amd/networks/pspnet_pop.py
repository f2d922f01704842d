"""PSPNet-POP on MI355X: drop-in for networks/pspnet_pop.py of LiZhuoHong/SegLand.

Same constructor, same parameter names/shapes, same forward dispatch and return conventions, same side effects
(forward_novel mutates mask_b in place; train-mode BN updates running stats; train_mode() leaves backbone+decoder in
eval).  All arithmetic runs in libsegland_hip.so (segland_amd.functional); there is no PyTorch/CPU fallback.

Extra (keyword-only) knob: compute_dtype = torch.bfloat16 (default, MFMA bf16 with fp32 accumulate) or torch.float32
(exact-fp32 MFMA; the parity mode).  BatchNorm statistics are always per-GPU (nn.SyncBatchNorm modules passed as
norm_layer are accepted as parameter holders; see DESIGN.md, multi-GPU).
"""
import torch
import torch.nn as nn
from torch.nn import functional as F

import os

from .. import ops
from ..functional import PopHeadFn, PPMFn, ProtoFn, cls_params, flush_num_batches_tracked, ppm_params, refresh_weights
from ..loss.criterion import OrthLoss, OrthTerm
from .backbones import get_backbone

_PROTO_FUSED = os.environ.get('SEGLAND_PROTO_FUSED', '1') != '0'


class PSPModule(nn.Module):
    """networks/pspnet_pop.py:8-35.  forward(x4 NHWC) -> features NHWC [B,h,w,out_features]."""

    def __init__(self, features, out_features=256, sizes=(1, 2, 3, 6), norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.sizes = tuple(sizes)
        self.stages = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(output_size=(s, s)), nn.Conv2d(features, out_features, kernel_size=1, bias=False),
                          norm_layer(out_features), nn.ReLU(inplace=True)) for s in sizes])
        self.bottleneck = nn.Sequential(
            nn.Conv2d(features + len(sizes) * out_features, out_features, kernel_size=3, padding=1, dilation=1, bias=False),
            norm_layer(out_features), nn.ReLU(inplace=True), nn.Conv2d(out_features, out_features, kernel_size=1))

    def forward(self, feats):
        return PPMFn.apply(feats, self, *ppm_params(self))


def _classifier(d):
    return nn.Sequential(nn.Conv2d(d, d, kernel_size=1, bias=False), nn.ReLU(inplace=True),
                         nn.Conv2d(d, d, kernel_size=1, bias=False), nn.ReLU(inplace=True),
                         nn.Conv2d(d, 1, kernel_size=1, bias=False))


_FEATURE_GRAPH = __import__('os').environ.get('SEGLAND_FEATURE_GRAPH', '1') != '0'


class GFSS_Model(nn.Module):
    """Segmenter for Generalized Few-shot Semantic Segmentation (networks/pspnet_pop.py:37-243)."""

    def __init__(self, n_base, criterion=None, norm_layer=nn.BatchNorm2d, use_base=True, is_ft=False, n_novel=0,
                 compute_dtype=torch.bfloat16, **kwargs):
        super().__init__()
        d_model = 512
        self.backbone = get_backbone(norm_layer=norm_layer, compute_dtype=compute_dtype, **kwargs)
        self.decoder = PSPModule(2048, out_features=d_model, norm_layer=norm_layer)
        self.classifier = _classifier(d_model)
        if is_ft:
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=False)
            self.novel_emb = nn.Parameter(torch.zeros(n_novel, d_model), requires_grad=True)
            self.classifier_n = _classifier(d_model)
            nn.init.orthogonal_(self.novel_emb)
            self.ft_freeze()
        else:
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=True)
            nn.init.orthogonal_(self.base_emb)
            self.novel_emb = None
        self.n_novel, self.use_base, self.is_ft = n_novel, use_base, is_ft
        self.criterion, self.n_base = criterion, n_base
        self.compute_dtype = compute_dtype

    # ---- helpers of the reference API
    def init_cls_n(self):
        for src, dst in zip(self.classifier.parameters(), self.classifier_n.parameters()):
            dst.data.copy_(src.data)

    def train_mode(self):
        self.train()
        self.backbone.eval()     # BN uses running statistics (pspnet_pop.py:82-84)
        self.decoder.eval()

    def ft_freeze(self):
        for part in (self.backbone, self.decoder, self.classifier):
            for p in part.parameters():
                p.requires_grad = False

    # ---- two-part backward for data parallelism (bucket_step.BucketedReplica): the gradients of everything behind the cut are complete when the backward
    # reaches the cut tensor(s), so their all-reduce can travel while the rest of the backward runs
    bucket_cut_default = True          # BucketedReplica: cut the data-parallel backward behind layer3 (82 % of the gradient bytes travel beside the second half)

    def enable_backward_cut(self, flag):
        self.backbone.__dict__['_sl_want_cut'] = bool(flag)

    def late_parameters(self):
        bb = self.backbone
        early = {id(p) for part in (bb.conv1, bb.bn1, bb.layer1, bb.layer2, bb.layer3) for p in part.parameters()}
        return [p for p in self.parameters() if id(p) not in early]

    def cut_tensors(self):
        """[(tensor, detached leaf the rest of the forward continued on)] of the last forward (ResNet: the output of layer3), or None.  With the cut
        enabled the backward MUST be run in two halves (bucket_step.BucketedReplica does): loss.backward(inputs = leaves + late_parameters()), then
        torch.autograd.backward(tensors, [leaf.grad ...], inputs = the other parameters)."""
        t = self.backbone.__dict__.get('_sl_cut')
        return [t] if t is not None else None

    def clear_cut(self):
        """Drop the stashed cut tensors: they keep the step's autograd graph -- and the parameters' AccumulateGrad nodes, which are bound to the stream they were
        created on -- alive; a HIP-graph capture on another stream would then reuse those nodes."""
        self.backbone.__dict__['_sl_cut'] = None
        self.__dict__['_sl_cut'] = None

    # ---- forward
    def forward(self, img, mask=None, img_b=None, mask_b=None):
        if self.is_ft:
            if self.training:
                return self.forward_novel(img, mask, img_b, mask_b)
            return self.forward_all(img, mask)
        return self.forward_base(img, mask)

    def _features_eager(self, img):
        refresh_weights(self)
        x4 = self.backbone.base_forward(img)
        feat = self.decoder(x4)
        flush_num_batches_tracked()
        return feat

    def _features(self, img):
        if not img.is_cuda:
            raise RuntimeError('segland_amd.GFSS_Model runs on the GPU only (no CPU fallback): move the model and inputs to cuda')
        if _FEATURE_GRAPH and not torch.cuda.is_current_stream_capturing() and ((self.is_ft and self.training) or (not self.training and not torch.is_grad_enabled())):
            feat = self._features_graphed(img)
            if feat is not None:
                return feat
        return self._features_eager(img)

    def _features_graphed(self, img):
        """ft_pop training (backbone + decoder frozen and in eval mode: train_mode(), pspnet_pop.py:80-85) and no-grad evaluation
        (validate(), eval_base.py): ~250 launches of a fixed kernel sequence per call with a call time of a few ms -- launch-bound.  The sequence is captured once per input shape into
        a HIP graph and replayed; any change of a frozen tensor (load_state_dict bumps the version counters) drops the graph."""
        frozen = list(self.backbone.parameters()) + list(self.decoder.parameters())
        trainable = any(p.requires_grad for p in frozen)
        if self.backbone.training or self.decoder.training or (trainable and torch.is_grad_enabled()):
            return None
        from ..functional import _OPT_EPOCH, _RS_EPOCH   # fused optimizers / graph replays change weights and running statistics without a version bump
        sig = (tuple(img.shape), img.dtype, img.device, _OPT_EPOCH[0] if trainable else 0, _RS_EPOCH[0],
               sum(p._version for p in frozen) + sum(b._version for b in self.backbone.buffers()) + sum(b._version for b in self.decoder.buffers()))
        ent = self.__dict__.get('_sl_graph')
        if ent is None or ent[0] != sig:
            try:
                static_in = img.detach().clone()
                cur = torch.cuda.current_stream()
                side = torch.cuda.Stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(2):                       # warm-up: weight copies, BN coefficients, workspaces, kernel attributes
                        self._features_eager(static_in)
                cur.wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                # thread_local: the drivers' DataLoader pin-memory thread calls hipHostMalloc / event queries concurrently; in the default
                # 'global' mode those calls fail in THAT thread (killing the loader) and invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode='thread_local'), torch.no_grad():
                    static_out = self._features_eager(static_in)
                ent = self.__dict__['_sl_graph'] = (sig, graph, static_in, static_out)
            except Exception as e:                            # stay on the eager HIP path
                import logging
                logging.warning('segland_amd: HIP graph capture of the frozen feature extractor failed (%s); running eagerly', e)
                ops.after_failed_capture()                    # a failed capture leaves work queued on the side stream (and the runtime's sticky error): drain before the eager path
                self.__dict__['_sl_graph'] = (sig, None, None, None)
                return None
        _, graph, static_in, static_out = ent
        if graph is None:
            return None
        static_in.copy_(img)
        graph.replay()
        return static_out.clone()

    def _protos(self):
        """(sb, sn | None, orth term): normalised prototypes + the orthogonality term of this mode's similarity matrix (base: sb sb^T, :185-186;
        ft: sn [sn ; sb]^T, :236-239), one kernel (functional.ProtoFn).  SEGLAND_PROTO_FUSED=0: the torch ops."""
        ka, kb = (self.novel_emb.shape[0], self.base_emb.shape[0]) if self.is_ft else (self.base_emb.shape[0], 0)
        if not _PROTO_FUSED or not self.base_emb.is_cuda or not ops.proto_fused_ok(ka, kb, self.base_emb.shape[1]):
            # torch ops: switched off, or more prototypes than the one-block kernels hold (--base-classes > 15 at 512 channels)
            sb = F.normalize(self.base_emb.float(), p=2, dim=-1)
            return sb, (F.normalize(self.novel_emb.float(), p=2, dim=-1) if self.is_ft else None), None
        if self.is_ft:
            sn, sb, orth = ProtoFn.apply(self.novel_emb, self.base_emb)
            return sb, sn, orth
        sb, _, orth = ProtoFn.apply(self.base_emb, None)
        return sb, None, orth

    def _head(self, feat):
        sb, sn, orth = self._protos()
        if self.is_ft:
            return PopHeadFn.apply(feat, sb, sn, self, *cls_params(self.classifier), *cls_params(self.classifier_n)), sb, sn, orth
        return PopHeadFn.apply(feat, sb, None, self, *cls_params(self.classifier)), sb, None, orth

    def forward_all(self, img, mask=None):
        return self._head(self._features(img))[0]

    def forward_base(self, img, mask=None):
        preds, sb, _, orth = self._head(self._features(img))
        if self.criterion is not None and mask is not None:
            if orth is not None and isinstance(self.criterion, OrthLoss):
                return self.criterion(preds, mask, proto_sim=OrthTerm(orth))
            proto_sim = torch.matmul(sb, sb.t())                       # [Kb,Kb] (pspnet_pop.py:185-186)
            return self.criterion(preds, mask, proto_sim=proto_sim)
        return preds

    def forward_novel(self, img, mask, img_b, mask_b):
        img_full = torch.cat([img, img_b], dim=0)
        preds, sb, sn, orth = self._head(self._features(img_full))
        B = img_full.shape[0]
        kb = self.n_base
        # pseudo-label the base tiles' background with the novel head (pspnet_pop.py:221-231); mutates mask_b in place
        preds2_b = torch.cat([preds[B // 2:, 0:1], preds[B // 2:, 1 + kb:]], dim=1).detach().contiguous()
        if not mask_b.is_contiguous():
            raise RuntimeError('mask_b must be contiguous (it is updated in place)')
        ops.pseudo_label_(preds2_b, mask_b, kb)
        if self.criterion is not None and mask is not None:
            mask_all = torch.cat([mask, mask_b], dim=0)
            if orth is not None and isinstance(self.criterion, OrthLoss):
                return self.criterion(preds, mask_all, is_ft=True, proto_sim=OrthTerm(orth))
            proto_sim = torch.matmul(sn, torch.cat([sn, sb], dim=0).t())   # [Kn, Kn+Kb] (pspnet_pop.py:236-239)
            return self.criterion(preds, mask_all, is_ft=True, proto_sim=proto_sim)
        return preds
