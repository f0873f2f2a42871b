"""get_nonspade_norm_layer (reference models/modules/spade_architecture/normalization.py:17-50): the SPADE discriminator's layer wrapper.

`norm_type` = ['spectral'] + {'', 'none', 'batch', 'syncbatch', 'instance'}: the optional prefix puts spectral normalisation on the conv,
the rest names the parameter-free norm that follows it.  A wrapped layer is `FusedSequential(conv, norm)` -- the reference's
`nn.Sequential`, so checkpoints keep their `0.*` / `1.*` keys -- and loses its bias, which the norm would cancel anyway."""
from . import nn as cnn

_FOLLOWING_NORM = {
    'batch': lambda width: cnn.BatchNorm2d(width, affine=True),
    'syncbatch': lambda width: cnn.SynchronizedBatchNorm2d(width, affine=True),
    'instance': lambda width: cnn.InstanceNorm2d(width, affine=False),
}


def get_nonspade_norm_layer(opt, norm_type='instance'):
    spectral = norm_type.startswith('spectral')
    kind = norm_type[len('spectral'):] if spectral else norm_type

    def wrap(layer):
        if spectral:
            layer = cnn.spectral_norm(layer)
        if kind in ('', 'none'):
            return layer
        if getattr(layer, 'bias', None) is not None:
            del layer.bias
            layer.register_parameter('bias', None)
        if kind not in _FOLLOWING_NORM:
            raise ValueError('normalization layer %s is not recognized' % kind)
        width = layer.out_channels if hasattr(layer, 'out_channels') else layer.weight.size(0)
        return cnn.FusedSequential(layer, _FOLLOWING_NORM[kind](width))

    return wrap
