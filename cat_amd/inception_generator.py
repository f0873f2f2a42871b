"""InceptionGenerator (CycleGAN / pix2pix teacher and student), reference
models/modules/inception_architecture/inception_generator.py:11-145: same constructor, same Sequential indices
(down_sampling.{1,2,4,5,7,8}, features.{0..8}, up_sampling.{0,1,3,4,7}) hence identical state_dict keys and hook names."""
import functools

from torch import nn

from . import nn as cnn
from .inception_modules import InvertedResidualChannels, _get_named_block_list, get_active_fn


class BaseNetwork(nn.Module):
    """reference models/networks.py:15-21"""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser


class InceptionGenerator(BaseNetwork):
    def __init__(self, input_nc, output_nc, ngf, channels, channels_reduction_factor, kernel_sizes, padding_type='reflect',
                 norm_layer=cnn.InstanceNorm2d, norm_momentum=0.1, norm_epsilon=1e-5, dropout_rate=0, active_fn='nn.ReLU',
                 n_blocks=9):
        assert n_blocks >= 0
        assert len(kernel_sizes) == len(set(kernel_sizes)), 'no duplicate in kernel sizes is allowed.'
        super(InceptionGenerator, self).__init__()
        if type(norm_layer) == functools.partial:
            use_bias = issubclass(norm_layer.func, nn.InstanceNorm2d)
        else:
            use_bias = issubclass(norm_layer, nn.InstanceNorm2d)
        norm_kwargs = {'momentum': norm_momentum, 'eps': norm_epsilon}
        active_fn = get_active_fn(active_fn)

        down_sampling = [cnn.ReflectionPad2d(3), cnn.Conv2d(input_nc, ngf, kernel_size=7, padding=0, bias=use_bias),
                         norm_layer(ngf), cnn.ReLU(True)]
        n_downsampling = 2
        for i in range(n_downsampling):
            mult = 2 ** i
            down_sampling += [cnn.Conv2d(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=1, bias=use_bias),
                              norm_layer(ngf * mult * 2), cnn.ReLU(True)]
        mult = 2 ** n_downsampling

        features = []
        for i in range(n_blocks):   # the reference builds them in three identical loops (n//3, n//3, rest)
            features += [InvertedResidualChannels(ngf * mult, res_channels=channels, dw_channels=channels,
                                                  channels_reduction_factor=channels_reduction_factor,
                                                  res_kernel_sizes=kernel_sizes, dw_kernel_sizes=kernel_sizes,
                                                  padding_type=padding_type, use_bias=use_bias, norm_layer=norm_layer,
                                                  norm_kwargs=norm_kwargs, dropout_rate=dropout_rate, active_fn=active_fn)]

        up_sampling = []
        for i in range(n_downsampling):
            mult = 2 ** (n_downsampling - i)
            up_sampling += [cnn.ConvTranspose2d(ngf * mult, int(ngf * mult / 2), kernel_size=3, stride=2, padding=1,
                                                output_padding=1, bias=use_bias),
                            norm_layer(int(ngf * mult / 2)), cnn.ReLU(True)]
        up_sampling += [cnn.ReflectionPad2d(3)]
        up_sampling += [cnn.Conv2d(ngf, output_nc, kernel_size=7, padding=0)]
        up_sampling += [cnn.Tanh()]
        self.down_sampling = cnn.FusedSequential(*down_sampling)
        self.features = cnn.FusedSequential(*features)
        self.up_sampling = cnn.FusedSequential(*up_sampling)

    def forward(self, input):
        res = self.down_sampling(input)
        res = self.features(res)
        res = self.up_sampling(res)
        return res

    def get_named_block_list(self):
        return _get_named_block_list(self)
