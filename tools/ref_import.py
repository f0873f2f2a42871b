"""Import snap-research/CAT (read-only at /root/reference) on CPU under torch 2.10.

Only used in THIS container by tools/make_golden.py and tools/time_reference_cpu.py to
generate golden vectors / reference timings.  Nothing under tests/, bench.py or the
product package imports this module: /root/reference does not exist on the GPU box.

Recipe follows SURVEY.md §8(c):
  1. import stdlib `profile`/`cProfile` BEFORE /root/reference is on sys.path
     (reference/profile.py shadows the stdlib module torch._dynamo pulls in);
  2. stub torchvision / cv2 / tensorboardX (absent offline, only needed by eval/logging code);
  3. neutralise torch.cuda.synchronize (utils/common.py:316,665 call it unconditionally);
  4. model_profiling defaults to use_cuda=True (utils/model_profiling.py:277) -> rebind.
"""
import cProfile  # noqa: F401  (step 1)
import profile  # noqa: F401
import functools
import sys
import types
from argparse import Namespace

import torch

REF = '/root/reference'


class _Permissive(types.ModuleType):
    """Stub module: any attribute that is not set explicitly resolves to a dummy class (import-time only)."""

    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        return type(item, (), {'__init__': lambda self, *a, **k: None, '__call__': lambda self, *a, **k: None})


def _stub(name, **attrs):
    m = _Permissive(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if getattr(install, 'done', False):
        return
    import torch.optim  # force-load anything that may import `profile`
    torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))])

    class _Dummy(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tv = _stub('torchvision')
    tvm = _stub('torchvision.models', inception_v3=lambda *a, **k: None, vgg19=lambda *a, **k: None)
    tvi = _stub('torchvision.models.inception', InceptionA=_Dummy, InceptionC=_Dummy, InceptionE=_Dummy,
                BasicConv2d=_Dummy)
    tvm.inception = tvi
    tv.models = tvm
    tvt = _stub('torchvision.transforms', Compose=object, functional=_stub('torchvision.transforms.functional'))
    tv.transforms = tvt
    tv.utils = _stub('torchvision.utils')
    _stub('cv2')
    _stub('tensorboardX', SummaryWriter=object)
    torch.cuda.synchronize = lambda *a, **k: None
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import utils.common as uc
    from utils.model_profiling import model_profiling
    uc.model_profiling = functools.partial(model_profiling, use_cuda=False)
    install.done = True


def make_opt(norm='instance', track=False, **kw):
    """The hot-path-relevant flags with the values the launch scripts use
    (scripts/cycle_gan/horse2zebra/train_inception_student_2p6B.sh,
     scripts/pix2pix/map2sat/train_inception_student_4p6B.sh)."""
    opt = Namespace(
        input_nc=3, output_nc=3, teacher_ngf=64, student_ngf=20, pretrained_ngf=64,
        teacher_netG='inception_9blocks', student_netG='inception_9blocks', pretrained_netG='inception_9blocks',
        norm=norm, norm_affine=True, norm_affine_D=True, norm_track_running_stats=track,
        norm_momentum=0.1, norm_epsilon=1e-5, channels=None, channels_reduction_factor=6,
        kernel_sizes=[1, 3, 5], active_fn='nn.ReLU', active_fn_D='nn.LeakyReLU',
        teacher_dropout_rate=0, student_dropout_rate=0, init_type='normal', init_gain=0.02,
        gpu_ids=[], ndf=128, netD='n_layers', n_layers_D=3, gan_mode='hinge',
        dataset_mode='aligned', direction='AtoB', lambda_distill=1.0, lambda_recon=100.0, lambda_gan=1.0,
        recon_loss_type='l1', distill_G_loss_type='ka', lr=2e-4, beta1=0.5, lr_policy='linear',
        nepochs=5, nepochs_decay=15, prune_cin_lb=16, target_flops=2.6e9,
        data_height=256, data_width=256, data_channel=3, prune_logging_verbose=False, isTrain=True,
        distiller='inception')
    opt.__dict__.update(kw)
    return opt
