"""Swin Transformer backbone on the HIP kernels (drop-in for networks/backbones/swintransformer.py of the reference, :436-650).

Same module tree / parameter names / shapes as the reference (`patch_embed.proj`, `patch_embed.norm`, `layers.{i}.blocks.{j}.{norm1, attn.
{relative_position_bias_table, relative_position_index, qkv, proj}, norm2, mlp.{fc1, fc2}}`, `layers.{i}.downsample.{reduction, norm}`,
`norm{0..3}`), so reference checkpoints load unchanged.  The modules only hold parameters; the arithmetic is in segland_amd.functional_swin
(PatchEmbedFn, SwinBlockFn, PatchMergeFn, LayerNormFn).  forward(img NCHW float) -> four NHWC token maps [B, H/4.., W/4.., P] in the compute
dtype (the reference returns NCHW; the decoder of networks/swin_pop.py consumes NHWC here).

DropPath (timm, drop_path_rate 0.2 growing linearly over the blocks, :478,535) is applied in train mode as a per-sample scale vector drawn with
torch.rand on the GPU; `drop_path_hook(block_index, B, p) -> tensor | None` on the backbone overrides the draw (parity tests feed the oracle's masks).
"""
import torch
import torch.nn as nn

from ...functional_swin import LayerNormFn, PatchEmbedFn, PatchMergeFn, SwinBlockFn, SwinLink, block_params

CONFIGS = {'swin-t': (96, (2, 2, 6, 2), (3, 6, 12, 24)), 'swin-s': (96, (2, 2, 18, 2), (3, 6, 12, 24)),
           'swin-b': (128, (2, 2, 18, 2), (4, 8, 16, 32)), 'swin-l': (192, (2, 2, 18, 2), (6, 12, 24, 48))}      # :485-507


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.act, self.fc2, self.drop = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim), nn.Dropout(0.0)


class WindowAttention(nn.Module):
    """Parameter holder of :71-116 (relative position bias table + index, qkv, proj)."""

    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        ws = window_size
        self.dim, self.window_size, self.num_heads = dim, (ws, ws), num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing='ij')
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        self.register_buffer('relative_position_index', (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1))
        self.qkv = nn.Linear(dim, dim * 3)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., drop_path=0., index=0):
        super().__init__()
        if window_size != 7 or dim != num_heads * 32:
            raise RuntimeError('segland_amd window attention: 7x7 windows and head_dim 32 (every Swin-T/S/B/L stage); got window %d, dim %d, heads %d'
                               % (window_size, dim, num_heads))
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.drop_path_p, self.index = float(drop_path), index
        self.drop_path = nn.Identity()            # parameter-free slot of the reference (timm DropPath): the scale vectors come from the backbone
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, drop_path, first_index, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([SwinTransformerBlock(dim, num_heads, 7, 0 if i % 2 == 0 else 3, 4., drop_path[i], first_index + i) for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm=True):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim) if norm else None


class SwinTransformer(nn.Module):
    def __init__(self, backbone='swin-t', drop_path_rate=0.2, compute_dtype=torch.bfloat16, **unused):
        super().__init__()
        if backbone not in CONFIGS:
            raise ValueError("Invalid indicator for backbone, which should be selected from {'swin-t', 'swin-s', 'swin-b', 'swin-l'}")
        dim, depths, heads = CONFIGS[backbone]
        self.filters = [dim * 2 ** i for i in range(4)]
        self.num_features, self.num_layers, self.embed_dim = self.filters, 4, dim
        self.compute_dtype = compute_dtype
        self.patch_embed = PatchEmbed(4, 3, dim, True)
        self.pos_drop = nn.Dropout(p=0.0)
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        self.layers = nn.ModuleList()
        k = 0
        for i in range(4):
            self.layers.append(BasicLayer(dim * 2 ** i, depths[i], heads[i], dpr[k:k + depths[i]], k, i < 3))
            k += depths[i]
        for i in range(4):
            self.add_module('norm%d' % i, nn.LayerNorm(self.filters[i]))
        self.drop_path_hook = None

    def get_filters(self):
        return self.filters

    def _drop_scale(self, blk, B, device):
        if self.drop_path_hook is not None:
            s = self.drop_path_hook(blk.index, B, blk.drop_path_p)
            return None if s is None else s.to(device=device, dtype=torch.float32).contiguous()
        if not self.training or blk.drop_path_p <= 0.0:
            return None
        keep = 1.0 - blk.drop_path_p
        return torch.floor(keep + torch.rand(B, device=device)) / keep          # timm DropPath, scale_by_keep

    def _drop_scales(self, B, device):
        """The two DropPath scales of every block (attention branch, MLP branch: `self.drop_path` is applied twice, swintransformer.py:246-249),
        drawn for the whole network with ONE torch.rand instead of one per call (4 tiny launches each, 24 calls per Swin-T forward)."""
        blocks = [blk for layer in self.layers for blk in layer.blocks]
        if self.drop_path_hook is not None or not self.training or all(b.drop_path_p <= 0.0 for b in blocks):
            out = []
            for blk in blocks:
                out += [self._drop_scale(blk, B, device), self._drop_scale(blk, B, device)]
            return out
        keep = self.__dict__.get('_sl_keep')
        if keep is None or keep.device != device:
            keep = self.__dict__['_sl_keep'] = torch.tensor([1.0 - b.drop_path_p for b in blocks for _ in (0, 1)], dtype=torch.float32, device=device).view(-1, 1)
        s = torch.floor(keep + torch.rand(len(blocks) * 2, B, device=device)) / keep        # timm DropPath, scale_by_keep: 0 or 1 / keep per sample
        return [None if blocks[k // 2].drop_path_p <= 0.0 else s[k] for k in range(2 * len(blocks))]

    def forward(self, img):
        if not img.is_cuda:
            raise RuntimeError('segland_amd SwinTransformer runs on the GPU only (no CPU fallback): move the model and inputs to cuda')
        pe = self.patch_embed
        x = PatchEmbedFn.apply(img.float().contiguous(), pe, self.compute_dtype, pe.proj.weight, pe.proj.bias,
                               *((pe.norm.weight, pe.norm.bias) if pe.norm is not None else (None, None)))
        B = x.shape[0]
        outs = []
        scales = self._drop_scales(B, x.device)
        for i, layer in enumerate(self.layers):
            Cn = self.filters[i]
            plink = None                      # DropPath hand-over between consecutive blocks of a stage (functional_swin.SwinLink)
            for blk in layer.blocks:
                s2 = scales[2 * blk.index + 1]
                link = SwinLink(s2) if s2 is not None else None
                x = SwinBlockFn.apply(x, blk, scales[2 * blk.index], s2, plink, link, *block_params(blk))
                plink = link
            nm = getattr(self, 'norm%d' % i)
            outs.append(LayerNormFn.apply(x, Cn, nm.weight, nm.bias))
            if layer.downsample is not None:
                ds = layer.downsample
                x = PatchMergeFn.apply(x, Cn, ds.reduction.weight, ds.norm.weight, ds.norm.bias)
        return tuple(outs)
