"""get_nonspade_norm_layer (reference models/modules/spade_architecture/normalization.py:17-50): wraps a conv of the SPADE
discriminator with spectral normalisation and a parameter-free norm; the returned nn.Sequential(layer, norm) keeps the
reference's `0.*` / `1.*` state_dict keys."""
from . import nn as cnn


def get_nonspade_norm_layer(opt, norm_type='instance'):
    def get_out_channel(layer):
        if hasattr(layer, 'out_channels'):
            return getattr(layer, 'out_channels')
        return layer.weight.size(0)

    def add_norm_layer(layer):
        nonlocal norm_type
        subnorm_type = norm_type
        if norm_type.startswith('spectral'):
            layer = cnn.spectral_norm(layer)
            subnorm_type = norm_type[len('spectral'):]
        if subnorm_type == 'none' or len(subnorm_type) == 0:
            return layer
        if getattr(layer, 'bias', None) is not None:     # the norm that follows cancels the bias
            delattr(layer, 'bias')
            layer.register_parameter('bias', None)
        if subnorm_type == 'batch':
            norm_layer = cnn.BatchNorm2d(get_out_channel(layer), affine=True)
        elif subnorm_type == 'syncbatch':
            norm_layer = cnn.SynchronizedBatchNorm2d(get_out_channel(layer), affine=True)
        elif subnorm_type == 'instance':
            norm_layer = cnn.InstanceNorm2d(get_out_channel(layer), affine=False)
        else:
            raise ValueError('normalization layer %s is not recognized' % subnorm_type)
        return cnn.FusedSequential(layer, norm_layer)

    return add_norm_layer
