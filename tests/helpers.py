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
    """The launch-script option set (one definition: cat_amd.synthetic.default_options, which bench.py uses as well)."""
    from cat_amd import synthetic
    return synthetic.default_options(norm=norm, track=track, **kw)


def teacher_sd(opt):
    """Canonical teacher (ngf 64) state_dict: shapes from the product's own define_G, values from detfill."""
    from cat_amd import networks
    T = networks.define_G(3, 3, 64, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
    return detfill.fill_state_dict(T.state_dict(), SEED_T, gamma_abs_normal=True)


def disc_sd(opt, d_in):
    from cat_amd import networks
    D = networks.define_D(d_in, opt.ndf, 'n_layers', 3, opt.norm, 'normal', 0.02, [], opt=opt)
    return detfill.fill_state_dict(D.state_dict(), SEED_D)


def netA_sds(student_sd, teacher_ngf=64):
    """state_dicts of the four 1x1 adaptors netAs (Conv2d(student trunk -> 4 * teacher_ngf)) as tools/make_golden.py fills them."""
    trunk = student_sd['down_sampling.7.weight'].shape[0]
    out = []
    for i in range(4):
        shapes = {'weight': torch.zeros(4 * teacher_ngf, trunk, 1, 1), 'bias': torch.zeros(4 * teacher_ngf)}
        out.append(detfill.fill_state_dict(shapes, SEED_A + i))
    return out


def sub(t, cmax=8, step=8):
    return t.detach().cpu()[:, :cmax, ::step, ::step].contiguous().numpy()


def cfg_for(norm):
    return {'norm': norm, 'eps': 1e-5, 'momentum': 0.1}


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def build_distiller(opt, student_shapes, d_in=None):
    """cat_amd InceptionDistiller on cuda:0 with the golden-vector networks: canonical teacher, the pruned student whose
    shapes the fixture records (values from detfill), discriminator from detfill."""
    import torch
    from cat_amd import networks
    from cat_amd.distillers import create_distiller
    from cat_amd.distillers.inception_distiller import InceptionDistiller
    from cat_amd.optim import FusedAdam
    import itertools
    model = create_distiller(opt, verbose=False)
    model.netG_teacher.load_state_dict(teacher_sd(opt))
    shapes = sd_from_shapes(student_shapes)
    student = student_from_shapes(opt, shapes).to(model.device)
    student.load_state_dict(detfill.fill_state_dict(shapes, SEED_S))
    student.train()
    model.netG_student = student
    if d_in is None:
        d_in = 6 if opt.dataset_mode == 'aligned' else 3
    model.netD.load_state_dict(disc_sd(opt, d_in))
    model.netD.train()
    trunk = student.state_dict()['down_sampling.7.weight'].shape[0]
    from cat_amd import nn as cnn
    model.netAs = [cnn.Conv2d(trunk, a.out_channels, 1).to(model.device) for a in model.netAs]      # shrink_model rebuilds them likewise
    for a, sd in zip(model.netAs, netA_sds(student.state_dict())):
        a.load_state_dict(sd)
    gp = [a.parameters() for a in model.netAs]
    model.optimizer_G = FusedAdam([{'params': model.netG_student.parameters()}, {'params': itertools.chain(*gp)}], lr=opt.lr,
                                  betas=(opt.beta1, 0.999))
    model.optimizers = [model.optimizer_G, model.optimizer_D]
    model.setup(opt, verbose=False)
    return model


def student_from_shapes(opt, shapes):
    """Instantiate a cat_amd InceptionGenerator whose layer widths match a recorded state_dict (a pruned student)."""
    from cat_amd import networks
    c0 = shapes['down_sampling.1.weight'].shape[0]
    net = networks.define_G(3, 3, 64, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
    from cat_amd import prune
    from cat_amd import nn as cnn
    # rebuild every layer at the recorded width
    def conv_like(old, w, transposed=False):
        if transposed:
            return cnn.ConvTranspose2d(w.shape[0], w.shape[1], kernel_size=old.kernel_size, stride=old.stride, padding=old.padding,
                                       output_padding=old.output_padding, bias=old.bias is not None)
        return cnn.Conv2d(w.shape[1], w.shape[0], kernel_size=old.kernel_size, stride=old.stride, padding=old.padding,
                          bias=old.bias is not None)
    for idx in (1, 4, 7):
        net.down_sampling[idx] = conv_like(net.down_sampling[idx], shapes[f'down_sampling.{idx}.weight'])
        net.down_sampling[idx + 1] = prune._new_norm(opt, shapes[f'down_sampling.{idx}.weight'].shape[0])
    trunk = shapes['down_sampling.7.weight'].shape[0]
    for i, blk in enumerate(net.features):
        blk.input_dim = trunk
        res, dw = [], []
        for j in range(3):
            k = f'features.{i}.res_ops.{j}.1.0.weight'
            res.append(shapes[k].shape[0] if k in shapes else 0)
            k = f'features.{i}.dw_ops.{j}.0.0.weight'
            dw.append(shapes[k].shape[0] if k in shapes else 0)
        # dropped branches shift the indices: recover per-kernel-size widths from the conv kernel sizes
        res_c, dw_c = [0, 0, 0], [0, 0, 0]
        ks = list(opt.kernel_sizes)
        for j in range(3):
            k = f'features.{i}.res_ops.{j}.1.0.weight'
            if k in shapes:
                res_c[ks.index(shapes[k].shape[-1])] = shapes[k].shape[0]
            k = f'features.{i}.dw_ops.{j}.2.0.weight'
            if k in shapes:
                dw_c[ks.index(shapes[k].shape[-1])] = shapes[k].shape[0]
        blk.res_channels, blk.dw_channels = res_c, dw_c
        blk.res_ops, blk.dw_ops, blk.pw_bn = blk._build()
    for idx in (0, 3):
        net.up_sampling[idx] = conv_like(net.up_sampling[idx], shapes[f'up_sampling.{idx}.weight'], transposed=True)
        net.up_sampling[idx + 1] = prune._new_norm(opt, shapes[f'up_sampling.{idx}.weight'].shape[1])
    net.up_sampling[7] = conv_like(net.up_sampling[7], shapes['up_sampling.7.weight'])
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    want = {k: tuple(v.shape) for k, v in shapes.items()}
    assert got == want, 'student rebuild does not match the recorded state_dict'
    return net
