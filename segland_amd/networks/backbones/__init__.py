"""Backbone factory with the reference's signature (networks/backbones/__init__.py:8-43).  Only the backbones on the
MI355X hot path are built here: resnet50 / resnet101."""
from .resnet import Bottleneck, ResNet


def get_backbone(norm_layer, pretrained_model=None, backbone='resnet101', relu_l3=True, relu_l4=True, **kwargs):
    layers = {'resnet50': [3, 4, 6, 3], 'resnet101': [3, 4, 23, 3]}.get(backbone)
    if layers is None:
        raise RuntimeError('unknown backbone: {} (segland_amd builds resnet50 / resnet101)'.format(backbone))
    model = ResNet(Bottleneck, layers, norm_layer=norm_layer, relu_l3=relu_l3, relu_l4=relu_l4, **kwargs)
    print('Backbone:' + backbone)
    if pretrained_model is not None:
        from ...utils.pyt_utils import load_model
        model = load_model(model, pretrained_model)
    return model
