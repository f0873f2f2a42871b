"""Shared test helpers: rebuild the golden-vector networks from seeds (oracle/detfill.py) and load fixtures."""
import json
import os
from argparse import Namespace

import numpy as np
import torch

from oracle import detfill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEED_T, SEED_S, SEED_D, SEED_A, SEED_X = 11, 21, 41, 51, 31   # tools/make_golden.py


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sd_from_shapes(js):
    sd = {}
    for k, shape in json.loads(str(js)):
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros(shape, dtype=torch.long)
        else:
            sd[k] = torch.zeros(shape, dtype=torch.float32)
    return sd


def make_opt(norm='instance', track=False, **kw):
    opt = Namespace(
        input_nc=3, output_nc=3, teacher_ngf=64, student_ngf=20, pretrained_ngf=64,
        teacher_netG='inception_9blocks', student_netG='inception_9blocks', pretrained_netG='inception_9blocks',
        norm=norm, norm_affine=True, norm_affine_D=True, norm_track_running_stats=track,
        norm_momentum=0.1, norm_epsilon=1e-5, channels=None, channels_reduction_factor=6,
        kernel_sizes=[1, 3, 5], active_fn='nn.ReLU', active_fn_D='nn.LeakyReLU',
        teacher_dropout_rate=0, student_dropout_rate=0, init_type='normal', init_gain=0.02,
        gpu_ids=[0], ndf=128, netD='n_layers', n_layers_D=3, gan_mode='hinge',
        dataset_mode='aligned', direction='AtoB', lambda_distill=1.0, lambda_recon=100.0, lambda_gan=1.0,
        recon_loss_type='l1', distill_G_loss_type='ka', lr=2e-4, beta1=0.5, lr_policy='linear',
        nepochs=5, nepochs_decay=15, prune_cin_lb=16, target_flops=2.6e9,
        data_height=256, data_width=256, data_channel=3, prune_logging_verbose=False, isTrain=True,
        distiller='inception', log_dir='/tmp/cat_amd_logs')
    opt.__dict__.update(kw)
    return opt


def teacher_sd(opt):
    """Canonical teacher (ngf 64) state_dict: shapes from the product's own define_G, values from detfill."""
    from cat_amd import networks
    T = networks.define_G(3, 3, 64, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
    return detfill.fill_state_dict(T.state_dict(), SEED_T, gamma_abs_normal=True)


def disc_sd(opt, d_in):
    from cat_amd import networks
    D = networks.define_D(d_in, opt.ndf, 'n_layers', 3, opt.norm, 'normal', 0.02, [], opt=opt)
    return detfill.fill_state_dict(D.state_dict(), SEED_D)


def sub(t, cmax=8, step=8):
    return t.detach().cpu()[:, :cmax, ::step, ::step].contiguous().numpy()


def cfg_for(norm):
    return {'norm': norm, 'eps': 1e-5, 'momentum': 0.1}


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
