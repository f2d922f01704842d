"""OrthLoss (loss/criterion.py:29-65 of the reference) with the bilinear upsample + cross-entropy fused in one HIP
kernel pair (the H x W logits are never materialised).  Returns the same dict of scalar tensors."""
import torch
import torch.nn as nn

from ..functional import UpsampleCEFn


class OrthTerm:
    """The orthogonality term already evaluated by the model's fused prototype kernel (functional.ProtoFn), handed to OrthLoss in the place of
    the similarity matrix it would be computed from.  A plain tensor `proto_sim` still works (the reference's call)."""

    def __init__(self, value):
        self.value = value


class OrthLoss(nn.Module):
    def __init__(self, ignore_index=255, reduction='mean'):
        super().__init__()
        if reduction != 'mean':
            raise ValueError('segland_amd OrthLoss implements reduction="mean" (what the reference uses)')
        self.ignore_index = ignore_index
        self.w = 10.0
        self._triu = {}

    def get_orth_loss(self, proto_sim, is_ft=False):
        # mean |.| over the strict upper triangle, also of a rectangular [K1,K2] matrix (criterion.py:37-43)
        # the reference selects with a boolean mask (a host synchronisation: the element count comes back); the same elements in the same
        # row-major order through constant indices keep the step free of read-backs (HIP-graph capture, graph_step.py) and the sum bit-identical
        if isinstance(proto_sim, OrthTerm):
            return proto_sim.value
        key = (tuple(proto_sim.shape), proto_sim.device)
        idx = self._triu.get(key)
        if idx is None:
            idx = self._triu[key] = torch.triu_indices(proto_sim.shape[0], proto_sim.shape[1], offset=1).to(proto_sim.device)
        return torch.abs(proto_sim[idx[0], idx[1]]).mean()

    def seg_loss(self, preds, target):
        return UpsampleCEFn.apply(preds.float(), target.contiguous(), self.ignore_index)

    def forward(self, preds, target, is_ft=False, proto_sim=None, aux_preds=None):
        seg_loss = self.seg_loss(preds, target)
        orth_loss = self.get_orth_loss(proto_sim, is_ft=is_ft)
        if aux_preds is not None:
            aux_loss = self.seg_loss(aux_preds, target)
            total = seg_loss + orth_loss * self.w + 0.4 * aux_loss
            return {'total_loss': total, 'seg_loss': seg_loss, 'aux_loss': aux_loss, 'orth_loss': orth_loss}
        # one launch each way (add with alpha; its backward is one scale) instead of mul + add
        return {'total_loss': torch.add(seg_loss, orth_loss, alpha=self.w), 'seg_loss': seg_loss, 'orth_loss': orth_loss}
