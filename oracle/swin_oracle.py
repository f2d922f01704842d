"""CPU oracle of the Swin-POP path: TEST INFRASTRUCTURE ONLY.

Plain torch-CPU fp32 restatement of LiZhuoHong/SegLand's
    networks/backbones/swintransformer.py   (PatchEmbed :395-433, WindowAttention :71-149, SwinTransformerBlock :152-250,
                                             PatchMerging :252-290, BasicLayer :293-392, SwinTransformer :436-650)
    networks/swin_pop.py                    (PSPModule :7-35, UperNet_Decoder_Plus :104-173, GFSS_Model :175-386)
with the reference's state_dict keys, so formula weights load into the reference, this oracle and the HIP model alike.  The POP head
and the loss are the ones of oracle/pop_oracle.py (swin_pop.py:238-386 differs from pspnet_pop.py only in the feature extractor and
d_model = backbone.get_filters()[0]).  Pinned: tests/golden/make_golden.py asserts it equals the imported reference (state_dict key
list, block / decoder / full-network outputs and gradients) before the golden vectors G13-G16 are stored.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The two stochastic layers of the train-mode reference (DropPath, swintransformer.py:185,243-244 via timm; nn.Dropout2d(0.1),
swin_pop.py:21) take their random masks from `model.rng` hooks here, so a test can feed the SAME masks to the oracle and to the HIP model:
    model.drop_path_scale(block_index, B, p) -> tensor [B] (0 or 1/(1-p)) or None (identity)
    model.dropout2d_scale(B, C, p)           -> tensor [B, C] (0 or 1/(1-p)) or None
Defaults draw from torch's global generator exactly like timm's DropPath / nn.Dropout2d (train mode) and are None in eval mode.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pop_oracle as po

SWIN = {'swin-t': (96, (2, 2, 6, 2), (3, 6, 12, 24)), 'swin-s': (96, (2, 2, 18, 2), (3, 6, 12, 24)),
        'swin-b': (128, (2, 2, 18, 2), (4, 8, 16, 32)), 'swin-l': (192, (2, 2, 18, 2), (6, 12, 24, 48))}     # swintransformer.py:485-507
WINDOW = 7
DROP_PATH_RATE = 0.2          # swintransformer.py:478
LN_EPS = 1e-5


class _Box(nn.Module):
    """Parameter container (the forward passes below are functions)."""


def relative_position_index(ws):
    # swintransformer.py:100-110: index into the (2ws-1)^2 bias table for every (query, key) pair of a ws x ws window
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing='ij')
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    return (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)


def make_block(dim, heads, ws=WINDOW):
    b = _Box()
    b.norm1 = nn.LayerNorm(dim)
    b.attn = _Box()
    b.attn.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
    b.attn.register_buffer('relative_position_index', relative_position_index(ws))
    b.attn.qkv = nn.Linear(dim, 3 * dim)
    b.attn.proj = nn.Linear(dim, dim)
    b.norm2 = nn.LayerNorm(dim)
    b.mlp = _Box()
    b.mlp.fc1 = nn.Linear(dim, 4 * dim)
    b.mlp.fc2 = nn.Linear(4 * dim, dim)
    b.heads, b.dim = heads, dim
    return b


def make_swin(backbone='swin-t'):
    dim, depths, heads = SWIN[backbone]
    net = _Box()
    net.patch_embed = _Box()
    net.patch_embed.proj = nn.Conv2d(3, dim, 4, stride=4)
    net.patch_embed.norm = nn.LayerNorm(dim)
    net.layers = nn.ModuleList()
    rates = torch.linspace(0, DROP_PATH_RATE, sum(depths)).tolist()       # swintransformer.py:535
    k = 0
    for i, (d, h) in enumerate(zip(depths, heads)):
        st = _Box()
        st.blocks = nn.ModuleList()
        for j in range(d):
            blk = make_block(dim * 2 ** i, h)
            blk.shift = 0 if j % 2 == 0 else WINDOW // 2                 # swintransformer.py:336
            blk.drop_path_p, blk.index = rates[k], k
            k += 1
            st.blocks.append(blk)
        if i < len(depths) - 1:
            st.downsample = _Box()
            st.downsample.reduction = nn.Linear(4 * dim * 2 ** i, 2 * dim * 2 ** i, bias=False)
            st.downsample.norm = nn.LayerNorm(4 * dim * 2 ** i)
        else:
            st.downsample = None
        net.layers.append(st)
    for i in range(4):
        setattr(net, 'norm%d' % i, nn.LayerNorm(dim * 2 ** i))
    net.filters = [dim * 2 ** i for i in range(4)]
    return net


def _cbr(cin, cout, k, bias):
    return nn.Sequential(nn.Conv2d(cin, cout, k, padding=k // 2, bias=bias), nn.BatchNorm2d(cout), nn.Identity())


def make_decoder(filters, dim, sizes=po.PPM_BINS):
    # swin_pop.py:104-138
    dec = _Box()
    dec.psp = _Box()
    dec.psp.sizes = tuple(sizes)
    dec.psp.stages = nn.ModuleList([nn.Sequential(nn.Identity(), nn.Conv2d(filters[-1], dim, 1, bias=False), nn.BatchNorm2d(dim), nn.Identity()) for _ in sizes])
    dec.psp.bottleneck = nn.Sequential(nn.Conv2d(filters[-1] + len(sizes) * dim, dim, 1, bias=False), nn.BatchNorm2d(dim), nn.Identity(), nn.Identity())
    dec.lateral_convs = nn.ModuleList([_cbr(c, dim, 3, True) for c in filters[:-1]])
    dec.fpn_convs = nn.ModuleList()
    for c in filters:
        n = max(1, int(torch.log2(torch.tensor(c)) - torch.log2(torch.tensor(filters[0]))))        # swin_pop.py:120-122, verbatim arithmetic
        head = []
        for _ in range(n):
            head.append(_cbr(dim, dim, 3, True))
            if c != filters[0]:
                head.append(nn.Identity())        # the parameter-free nn.Upsample slot (keeps the Sequential indices of the reference)
        dec.fpn_convs.append(nn.Sequential(*head))
    return dec


class SwinPopOracle(nn.Module):
    """networks/swin_pop.py:175-229 (constructor surface, state_dict keys)."""

    def __init__(self, n_base, criterion=None, is_ft=False, n_novel=0, backbone='swin-t'):
        super().__init__()
        d_model = SWIN[backbone][0]
        if is_ft:
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=False)
            self.novel_emb = nn.Parameter(torch.zeros(n_novel, d_model), requires_grad=True)
        else:
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=True)
            self.novel_emb = None
        self.backbone = make_swin(backbone)
        self.decoder = make_decoder(self.backbone.filters, d_model)
        self.classifier = po.make_classifier(d_model)
        if is_ft:
            self.classifier_n = po.make_classifier(d_model)
            nn.init.orthogonal_(self.novel_emb)
            po.ft_freeze(self)
        else:
            nn.init.orthogonal_(self.base_emb)
        self.n_base, self.n_novel, self.is_ft, self.criterion = n_base, n_novel, is_ft, criterion
        self.drop_path_scale = self._default_drop_path
        self.dropout2d_scale = self._default_dropout2d

    # ---- default random masks: timm DropPath (scale_by_keep) / nn.Dropout2d semantics
    def _default_drop_path(self, index, B, p):
        if not self.backbone.training or p <= 0.0:
            return None
        keep = 1.0 - p
        return torch.floor(keep + torch.rand(B)) / keep

    def _default_dropout2d(self, B, C, p):
        if not self.decoder.training or p <= 0.0:
            return None
        return F.dropout2d(torch.ones(B, C, 1, 1), p, True).view(B, C)       # the same noise shape / generator draw as nn.Dropout2d on [B,C,h,w]

    def features(self, img):
        return decoder_forward(self, swin_forward(self, img))

    def forward(self, img, mask=None, img_b=None, mask_b=None):
        # swin_pop.py:266-279
        if self.is_ft:
            if self.training:
                return forward_novel(self, img, mask, img_b, mask_b)
            return po.head_all(self, self.features(img))[0]
        preds = po.head_base(self, self.features(img))
        if self.criterion is not None and mask is not None:
            e = F.normalize(self.base_emb.unsqueeze(0), p=2, dim=-1).squeeze(0)
            return self.criterion(preds, mask, proto_sim=torch.matmul(e, e.t()))
        return preds


def train_mode(model, backbone_only=False):
    # swin_pop.py:220-228
    model.train()
    model.backbone.eval()
    if not backbone_only:
        model.decoder.eval()
        for p in model.decoder.parameters():
            p.requires_grad = False


def forward_novel(model, img, mask, img_b, mask_b):
    # swin_pop.py:336-386 == pspnet_pop.py:191-243 on the Swin features
    feats = model.features(torch.cat([img, img_b], 0))
    preds, preds2 = po.head_all(model, feats)
    B = feats.shape[0]
    mask_new = torch.stack([po.pseudo_label(preds2[B // 2 + b], mask_b[b], model.n_base) for b in range(B // 2)], 0)
    if model.criterion is not None and mask is not None:
        sn = F.normalize(model.novel_emb.float(), p=2, dim=-1)
        sb = F.normalize(model.base_emb.float(), p=2, dim=-1)
        return model.criterion(preds.float(), torch.cat([mask, mask_new], 0), is_ft=True, proto_sim=torch.matmul(sn, torch.cat([sn, sb], 0).t()))
    return preds


# ------------------------------------------------------------------------------------------------ Swin backbone
def _ln(x, m):
    return F.layer_norm(x, (x.shape[-1],), m.weight, m.bias, LN_EPS)


def shift_mask(Hp, Wp, ws, shift):
    """swintransformer.py:363-379: additive mask [nW, ws*ws, ws*ws] (0 / -100) separating the three row bands x three column bands of the
    cyclically shifted map."""
    band = lambda n: torch.where(torch.arange(n) < n - ws, 0, torch.where(torch.arange(n) < n - shift, 1, 2))
    ids = (3 * band(Hp)[:, None] + band(Wp)[None, :]).float()                                   # region id per position
    w = ids.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)           # per window
    diff = w[:, None, :] - w[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def window_attention(xw, a, heads, mask):
    """swintransformer.py:118-149.  xw [nW*B, N, C] -> [nW*B, N, C]."""
    Bw, N, C = xw.shape
    hd = C // heads
    qkv = F.linear(xw, a.qkv.weight, a.qkv.bias).view(Bw, N, 3, heads, hd)
    q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)     # [Bw, heads, N, hd]
    att = torch.matmul(q * hd ** -0.5, k.transpose(-2, -1))
    bias = a.relative_position_bias_table[a.relative_position_index.view(-1)].view(N, N, heads).permute(2, 0, 1)
    att = att + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        att = (att.view(Bw // nW, nW, heads, N, N) + mask[None, :, None]).view(Bw, heads, N, N)
    att = torch.softmax(att, dim=-1)
    out = torch.matmul(att, v).transpose(1, 2).reshape(Bw, N, C)
    return F.linear(out, a.proj.weight, a.proj.bias)


def block_forward(model, blk, x, H, W, mask):
    """swintransformer.py:195-250.  x [B, H*W, C]."""
    B, L, C = x.shape
    ws, shift = WINDOW, blk.shift
    y = _ln(x, blk.norm1).view(B, H, W, C)
    pb, pr = (ws - H % ws) % ws, (ws - W % ws) % ws
    y = F.pad(y, (0, 0, 0, pr, 0, pb))                          # zeros AFTER the norm: padded tokens carry qkv = bias
    Hp, Wp = H + pb, W + pr
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = y.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    yw = window_attention(yw, blk.attn, blk.heads, mask if shift > 0 else None)
    y = yw.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y[:, :H, :W].reshape(B, L, C)
    s = model.drop_path_scale(blk.index, B, blk.drop_path_p)
    x = x + (y if s is None else y * s.view(B, 1, 1))
    z = F.linear(F.gelu(F.linear(_ln(x, blk.norm2), blk.mlp.fc1.weight, blk.mlp.fc1.bias)), blk.mlp.fc2.weight, blk.mlp.fc2.bias)
    s = model.drop_path_scale(blk.index, B, blk.drop_path_p)
    return x + (z if s is None else z * s.view(B, 1, 1))


def patch_merging(ds, x, H, W):
    # swintransformer.py:264-290
    B, L, C = x.shape
    x = F.pad(x.view(B, H, W, C), (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    return F.linear(_ln(x.view(B, -1, 4 * C), ds.norm), ds.reduction.weight)


def swin_forward(model, img):
    """swintransformer.py:617-641: four NCHW feature maps (strides 4, 8, 16, 32), each through its own output LayerNorm."""
    net = model.backbone
    pe = net.patch_embed
    H, W = img.shape[2:]
    img = F.pad(img, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))                                   # :417-421
    x = F.conv2d(img, pe.proj.weight, pe.proj.bias, stride=4)
    B, C, H, W = x.shape
    x = _ln(x.flatten(2).transpose(1, 2), pe.norm)
    outs = []
    for i, st in enumerate(net.layers):
        ws = WINDOW
        Hp, Wp = math.ceil(H / ws) * ws, math.ceil(W / ws) * ws
        mask = shift_mask(Hp, Wp, ws, ws // 2)
        for blk in st.blocks:
            x = block_forward(model, blk, x, H, W, mask)
        o = _ln(x, getattr(net, 'norm%d' % i))
        outs.append(o.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous())
        if st.downsample is not None:
            x = patch_merging(st.downsample, x, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    return outs


# ------------------------------------------------------------------------------------------------ UperNet_Decoder_Plus
def _cbr_forward(seq, x):
    return F.relu(po._bn(po._cv(x, seq[0]), seq[1]))


def _up(x, size):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=True)


def psp_forward(model, psp, feats):
    # swin_pop.py:31-35 (+ :15-29): align_corners=True priors, 1x1 bottleneck, Dropout2d(0.1)
    h, w = feats.shape[2:]
    priors = [_up(F.relu(po._bn(po._cv(F.adaptive_avg_pool2d(feats, (s, s)), st[1]), st[2])), (h, w)) for s, st in zip(psp.sizes, psp.stages)]
    y = F.relu(po._bn(po._cv(torch.cat(priors + [feats], 1), psp.bottleneck[0]), psp.bottleneck[1]))
    m = model.dropout2d_scale(y.shape[0], y.shape[1], 0.1)
    return y if m is None else y * m.view(y.shape[0], y.shape[1], 1, 1)


def decoder_forward(model, xs):
    """swin_pop.py:140-173: laterals, top-down sums, per-level conv(+x2 upsample) heads, sum at the finest resolution."""
    dec = model.decoder
    lat = [_cbr_forward(l, x) for l, x in zip(dec.lateral_convs, xs[:-1])] + [psp_forward(model, dec.psp, xs[-1])]
    for i in range(len(lat) - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + _up(lat[i], lat[i - 1].shape[2:])
    outs = []
    for head, f in zip(dec.fpn_convs, lat):
        for m in head:
            f = _up(f, (2 * f.shape[2], 2 * f.shape[3])) if isinstance(m, nn.Identity) else _cbr_forward(m, f)     # nn.Upsample(scale_factor=2)
        outs.append(f)
    size = xs[0].shape[-2:]
    outs = [f if f.shape[-2:] == size else _up(f, size) for f in outs]
    return torch.stack(outs, dim=-1).sum(-1)


def init_weights(model, seed=0):
    """Reference initialisation (swintransformer.py:587-596 is only applied through init_weights(), which swin_pop never calls: the model
    trains from torch's default Linear / LayerNorm init + trunc_normal(0.02) bias tables).  Provided for benchmarks."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, _Box) and hasattr(m, 'relative_position_bias_table'):
            with torch.no_grad():
                m.relative_position_bias_table.copy_(torch.empty_like(m.relative_position_bias_table).normal_(0, 0.02, generator=g).clamp_(-0.04, 0.04))
    return model
