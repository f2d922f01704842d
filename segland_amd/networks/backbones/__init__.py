"""Backbone factory with the reference's signature (networks/backbones/__init__.py:8-43).  Only the backbones on the
MI355X hot path are built here: resnet50 / resnet101 (rows a-1..a-3) and swin-t / -s / -b / -l (row f-1)."""
from .resnet import Bottleneck, ResNet


def get_backbone(norm_layer, pretrained_model=None, backbone='resnet101', relu_l3=True, relu_l4=True, **kwargs):
    if backbone.startswith('swin-'):
        from .swintransformer import SwinTransformer          # networks/backbones/__init__.py:21-32: window 7, pretrain size 224
        model = SwinTransformer(backbone=backbone, compute_dtype=kwargs.get('compute_dtype') or __import__('torch').bfloat16)
    else:
        layers = {'resnet50': [3, 4, 6, 3], 'resnet101': [3, 4, 23, 3]}.get(backbone)
        if layers is None:
            raise RuntimeError('unknown backbone: {} (segland_amd builds resnet50 / resnet101 / swin-t / swin-s / swin-b / swin-l)'.format(backbone))
        model = ResNet(Bottleneck, layers, norm_layer=norm_layer, relu_l3=relu_l3, relu_l4=relu_l4, **kwargs)
    print('Backbone:' + backbone)
    if pretrained_model is not None:
        from ...utils.pyt_utils import load_model
        model = load_model(model, pretrained_model)
    return model
