"""Factories with the reference's signatures: models/networks.py:29-64 (get_norm_layer), :106-164 (init_weights /
init_net), :167-266 (define_G / define_D).  One process drives one GPU, so `gpu_ids` only selects the device
(data parallelism is cat_amd.parallel, not nn.DataParallel)."""
import functools

import torch
from torch.nn import init
from torch.optim import lr_scheduler

from . import nn as cnn


def get_norm_layer(norm_type='instance', affine=True, track_running_stats=True):
    if norm_type == 'batch':
        return functools.partial(cnn.BatchNorm2d, affine=affine, track_running_stats=track_running_stats)
    if norm_type == 'instance':
        return functools.partial(cnn.InstanceNorm2d, affine=affine, track_running_stats=track_running_stats)
    if norm_type == 'none':
        return lambda x: cnn.Identity()
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def get_scheduler(optimizer, opt):
    """reference networks.py:67-103 ('linear' is what every distillation script uses)."""
    if opt.lr_policy == 'linear':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + 1 - opt.nepochs) / float(opt.nepochs_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.niter, eta_min=0)
    raise NotImplementedError('learning rate policy [%s] is not implemented' % opt.lr_policy)


_WEIGHT_INIT = {
    'normal': lambda w, gain: init.normal_(w, 0.0, gain),
    'xavier': lambda w, gain: init.xavier_normal_(w, gain=gain),
    'kaiming': lambda w, gain: init.kaiming_normal_(w, a=0, mode='fan_in'),
    'orthogonal': lambda w, gain: init.orthogonal_(w, gain=gain),
}


def init_weights(net, init_type='normal', init_gain=0.02, verbose=False):
    """reference networks.py:106-147: conv / linear weights by `init_type`, their biases 0; BatchNorm2d weights N(1, gain), biases 0.
    Modules are matched by class NAME (so SynchronizedBatchNorm2d counts as BatchNorm2d, InstanceNorm2d keeps its defaults), in
    `net.apply` order -- the order in which the random stream is consumed."""
    fill = _WEIGHT_INIT.get(init_type)

    def visit(m):
        kind = type(m).__name__
        if not hasattr(m, 'weight'):                         # containers (ConvBNReLU ...) are matched by name too, but own no weight
            return
        if 'Conv' in kind or 'Linear' in kind:
            if fill is None:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            fill(m.weight.data, init_gain)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif 'BatchNorm2d' in kind and m.weight is not None:
            init.normal_(m.weight.data, 1.0, init_gain)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)

    if verbose:
        print('initialize network with %s' % init_type)
    net.apply(visit)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(torch.device('cuda', gpu_ids[0]))
    init_weights(net, init_type, init_gain=init_gain)
    return net


def define_G(input_nc, output_nc, ngf, netG, norm='batch', dropout_rate=0, init_type='normal', init_gain=0.02, gpu_ids=[],
             opt=None):
    norm_layer = get_norm_layer(norm_type=norm, affine=getattr(opt, 'norm_affine', False),
                                track_running_stats=getattr(opt, 'norm_track_running_stats', False))
    if netG == 'inception_9blocks':
        from .inception_generator import InceptionGenerator
        net = InceptionGenerator(input_nc, output_nc, ngf=ngf, channels=opt.channels,
                                 channels_reduction_factor=opt.channels_reduction_factor, kernel_sizes=opt.kernel_sizes,
                                 norm_layer=norm_layer, norm_momentum=opt.norm_momentum, norm_epsilon=opt.norm_epsilon,
                                 dropout_rate=dropout_rate, active_fn=opt.active_fn, n_blocks=9)
    elif netG == 'inception_spade':
        from .inception_spade_generator import InceptionSPADEGenerator
        net = InceptionSPADEGenerator(opt)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[], opt=None):
    norm_layer = get_norm_layer(norm_type=norm, affine=getattr(opt, 'norm_affine_D', False),
                                track_running_stats=getattr(opt, 'norm_track_running_stats', False))
    if netD == 'multi_scale':
        from .discriminators import MultiscaleDiscriminator
        net = MultiscaleDiscriminator(opt)
    elif netD == 'n_layers':
        from .discriminators import NLayerDiscriminator
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer, active_fn=opt.active_fn_D)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)
