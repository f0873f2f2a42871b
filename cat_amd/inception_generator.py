"""InceptionGenerator (CycleGAN / pix2pix teacher and student), reference
models/modules/inception_architecture/inception_generator.py:11-145.

The three containers keep the reference's names and positions -- `down_sampling.{1,2,4,5,7,8}`, `features.{0..n-1}`,
`up_sampling.{0,1,3,4,7}` -- because those indices ARE the checkpoint keys and the names the distillation hooks tap.  Layout:

    down_sampling : pad3 . conv7x7(in -> w0) . norm . relu | conv3x3 s2 (w0 -> w1) . norm . relu | conv3x3 s2 (w1 -> w2) . norm . relu
    features      : n_blocks x InvertedResidualChannels(w2)
    up_sampling   : convT3x3 s2 (w2 -> w1) . norm . relu | convT3x3 s2 (w1 -> w0) . norm . relu | pad3 . conv7x7(w0 -> out, bias) . tanh

with (w0, w1, w2) = (ngf, 2 ngf, 4 ngf).  Convs carry a bias exactly when the norm is an InstanceNorm (reference :30-33)."""
import functools

from torch import nn

from . import nn as cnn
from .inception_modules import InvertedResidualChannels, _get_named_block_list, get_active_fn


class BaseNetwork(nn.Module):
    """reference models/networks.py:15-21"""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser


def _is_instance_norm(norm_layer):
    cls = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    return issubclass(cls, nn.InstanceNorm2d)


class InceptionGenerator(BaseNetwork):
    def __init__(self, input_nc, output_nc, ngf, channels, channels_reduction_factor, kernel_sizes, padding_type='reflect',
                 norm_layer=cnn.InstanceNorm2d, norm_momentum=0.1, norm_epsilon=1e-5, dropout_rate=0, active_fn='nn.ReLU',
                 n_blocks=9):
        if n_blocks < 0:
            raise AssertionError('n_blocks must be >= 0')
        if len(set(kernel_sizes)) != len(kernel_sizes):
            raise AssertionError('no duplicate in kernel sizes is allowed.')
        super().__init__()
        bias = _is_instance_norm(norm_layer)
        widths = [ngf, 2 * ngf, 4 * ngf]

        def normed(conv, width):
            return [conv, norm_layer(width), cnn.ReLU(True)]

        stem = [cnn.ReflectionPad2d(3)] + normed(cnn.Conv2d(input_nc, widths[0], kernel_size=7, padding=0, bias=bias), widths[0])
        for narrow, wide in zip(widths[:-1], widths[1:]):
            stem += normed(cnn.Conv2d(narrow, wide, kernel_size=3, stride=2, padding=1, bias=bias), wide)

        block_args = dict(res_channels=channels, dw_channels=channels, channels_reduction_factor=channels_reduction_factor,
                          res_kernel_sizes=kernel_sizes, dw_kernel_sizes=kernel_sizes, padding_type=padding_type, use_bias=bias,
                          norm_layer=norm_layer, norm_kwargs={'momentum': norm_momentum, 'eps': norm_epsilon},
                          dropout_rate=dropout_rate, active_fn=get_active_fn(active_fn))
        trunk = [InvertedResidualChannels(widths[-1], **block_args) for _ in range(n_blocks)]

        head = []
        for wide, narrow in zip(widths[:0:-1], widths[-2::-1]):
            head += normed(cnn.ConvTranspose2d(wide, narrow, kernel_size=3, stride=2, padding=1, output_padding=1, bias=bias), narrow)
        head += [cnn.ReflectionPad2d(3), cnn.Conv2d(widths[0], output_nc, kernel_size=7, padding=0), cnn.Tanh()]

        self.down_sampling = cnn.FusedSequential(*stem)
        self.features = cnn.FusedSequential(*trunk)
        self.up_sampling = cnn.FusedSequential(*head)

    def forward(self, input):
        if self.training:      # the fused training-mode blocks refresh their packed operands once per optimizer step: all of them in one launch
            from . import fused_block
            fused_block.prepare_many(list(self.features))
        return self.up_sampling(self.features(self.down_sampling(input)))

    def get_named_block_list(self):
        return _get_named_block_list(self)
