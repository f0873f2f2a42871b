"""Seeded synthetic inputs, weights and option sets for bench.py / smoke runs (there is no dataset or checkpoint on the GPU box).

numpy's PCG64 + ziggurat normals are bit-reproducible across machines (torch's vectorised normal_() is not), so the same seeds
give the same teacher / batches everywhere.  The fills follow SURVEY §8(d): images are N(0,1).tanh() in (-1,1); the canonical
teacher has N(0, 1/fan_in) conv weights and |N(0,1)| norm scales (a trained teacher's scales are non-uniform, which is what makes
pruning non-degenerate).  oracle/detfill.py is the test infrastructure's own copy of the same fills (tests/test_host.py pins the two
to each other), so nothing on the measured path imports oracle/."""
from argparse import Namespace

import numpy as np
import torch

SEED_TEACHER = 11         # == tests/helpers.SEED_T: bench and the golden fixtures share the canonical teacher
SEED_TEACHER_SPADE = 111


def _rng(seed):
    return np.random.default_rng(seed)


def normal(shape, seed, scale=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy((_rng(seed).standard_normal(n, dtype=np.float32) * np.float32(scale)).reshape(shape))


def images(shape, seed):
    """N(0,1).tanh() images in (-1,1)."""
    return torch.tanh(normal(shape, seed))


def label_maps(batch, h, w, seed, n_labels=35, n_instances=1000, block=16):
    """Cityscapes-like label / instance-id maps: random ids per block x block cell, so regions and instance edges exist."""
    rng = _rng(seed)
    up = lambda a: np.repeat(np.repeat(a, block, 2), block, 3)
    lab = up(rng.integers(0, n_labels, (batch, 1, h // block, w // block))).astype(np.int32)
    ins = up(rng.integers(0, n_instances, (batch, 1, h // block, w // block))).astype(np.int32)
    return torch.from_numpy(lab), torch.from_numpy(ins)


def fill_state_dict(sd, seed, gamma_abs_normal=False):
    """Overwrite every tensor of a state_dict in key order.
    conv / conv-transpose weights ~ N(0, 1/fan_in); biases ~ 0.1 N(0,1); norm weight ~ 1 + 0.2 N(0,1) (or |N(0,1)| for the
    canonical teacher); running_mean ~ 0.1 N; running_var ~ U(0.5, 1.5)."""
    rng = _rng(seed)
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        n = int(np.prod(shape)) if len(shape) else 1
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros_like(v)
            continue
        z = rng.standard_normal(n, dtype=np.float32)
        if k.endswith('running_var'):
            t = 0.5 + rng.random(n, dtype=np.float32)
        elif k.endswith('running_mean'):
            t = 0.1 * z
        elif v.dim() >= 2:
            t = z / np.float32(np.sqrt(int(np.prod(shape[1:]))))
        elif k.endswith('.weight'):
            t = np.abs(z) if gamma_abs_normal else 1.0 + 0.2 * z
        else:
            t = 0.1 * z
        out[k] = torch.from_numpy(np.asarray(t, dtype=np.float32).reshape(shape)).clone()
    return out


def default_options(norm='instance', track=False, **kw):
    """The option set of the reference's inception-distiller launch scripts (distill_options.py / base_inception_distiller.py
    defaults + scripts/*/distill.sh), as the Namespace `create_distiller(opt)` consumes."""
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
