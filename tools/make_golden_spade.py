"""Generate tests/golden/spade_*.npz by RUNNING THE REFERENCE's GauGAN / SPADE distillation path (snap-research/CAT imported
from /root/reference on CPU).  Build container only:  python tools/make_golden_spade.py

Recipe: SPADEDistillerModules is built for real (define_G / define_D / netAs / GANLoss / VGGLoss);
torchvision is absent offline, so `torchvision.models.vgg19` is stubbed with a seeded, NARROW network of identical layer
topology (the reference's VGG19 class slices `.features` by index only, models/modules/loss.py:151-186); Adam is
constructed with betas (0.0, 0.9) by hand because torch 2.10 rejects the int 0 the reference passes
(base_spade_distiller_modules.py:96-101); backward_G / backward_D / optimize_parameters are the three statements of
models/spade_model.py:189-203 and distillers/base_spade_distiller.py:226-234.  No reference source is stored, only
seeds, shapes and outputs."""
import copy
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

ref_import.install()
import torch  # noqa: E402
from torch import nn  # noqa: E402

from oracle import detfill  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_num_threads(8)
SEED_T, SEED_S, SEED_D, SEED_V, SEED_X = 111, 121, 141, 161, 131
VGG_WIDTH_DIV = 8


def narrow_vgg19_features():
    """torchvision.models.vgg19().features topology (cfg 'E'), channel widths divided by VGG_WIDTH_DIV."""
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v // VGG_WIDTH_DIV, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v // VGG_WIDTH_DIV
    feats = nn.Sequential(*layers)
    feats.load_state_dict(detfill.fill_state_dict(feats.state_dict(), SEED_V))
    return feats


class _VGGStub:
    def __init__(self):
        self.features = narrow_vgg19_features()


sys.modules['torchvision.models'].vgg19 = lambda *a, **k: _VGGStub()
sys.modules['torchvision'].models.vgg19 = sys.modules['torchvision.models'].vgg19

from models.modules.spade_modules.spade_distiller_modules import SPADEDistillerModules  # noqa: E402  (reference)
from models.spade_model import SPADEModel  # noqa: E402
from models import networks  # noqa: E402
from utils.model_profiling import model_profiling  # noqa: E402
from utils import common as uc  # noqa: E402


def spade_opt(**kw):
    """Flags of scripts/gaugan/cityscapes/train_inception_student_5p6B.sh + the defaults of base_spade_distiller.py:27-138,
    spade_distiller.py:24-84 and discriminators.py:185-203, at a fixture-sized geometry."""
    opt = ref_import.make_opt()
    opt.__dict__.update(dict(
        input_nc=5, output_nc=3, semantic_nc=6, contain_dontcare_label=False, no_instance=False,
        teacher_ngf=8, student_ngf=6, pretrained_ngf=8, teacher_netG='inception_spade', student_netG='inception_spade',
        pretrained_netG='inception_spade', teacher_norm_G='spadesyncbatch3x3', student_norm_G='spadesyncbatch3x3',
        pretrained_norm_G='spadesyncbatch3x3', norm_G='spadesyncbatch3x3', num_upsampling_layers='more', crop_size=256,
        aspect_ratio=2.0, netD='multi_scale', ndf=8, n_layers_D=4, num_D=2, norm_D='spectralinstance', norm='instance',
        init_type='xavier', init_gain=0.02, gan_mode='hinge', lambda_gan=1.0, lambda_feat=10.0, lambda_vgg=10.0,
        lambda_distill=0.5, distill_G_loss_type='ka', no_TTUR=False, lr=2e-4, beta1=0.5, beta2=0.999, distiller='spade',
        channels=None, channels_reduction_factor=6, kernel_sizes=[1, 3, 5], active_fn='nn.ReLU', isTrain=True, gpu_ids=[],
        restore_pretrained_G_path=None, restore_teacher_G_path=None, restore_student_G_path=None, restore_D_path=None,
        restore_A_path=None))
    opt.__dict__.update(kw)
    return opt


def shapes_json(sd):
    return json.dumps([[k, list(v.shape)] for k, v in sd.items()])


def sub(t, cmax=6, step=4):
    return t.detach()[:, :cmax, ::step, ::step].contiguous().numpy()


def synth_inputs(n, h, w, nlabel, seed):
    """SURVEY §8d inputs: labels / instance ids constant over 16x16 blocks (so regions and edges exist), image ~ tanh(N(0,1))."""
    rng = np.random.default_rng(seed)
    bh, bw = max(h // 16, 1), max(w // 16, 1)
    lab = rng.integers(0, nlabel, size=(n, 1, bh, bw))
    ins = rng.integers(0, 1000, size=(n, 1, bh, bw))
    up = lambda a: np.repeat(np.repeat(a, h // bh, axis=2), w // bw, axis=3)
    return up(lab).astype(np.int64), up(ins).astype(np.int32), detfill.images((n, 3, h, w), seed + 1)


def checks(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def build(opt):
    m = SPADEDistillerModules(opt)
    m.netG_teacher.load_state_dict(detfill.fill_state_dict(m.netG_teacher.state_dict(), SEED_T))
    m.netG_student.load_state_dict(detfill.fill_state_dict(m.netG_student.state_dict(), SEED_S))
    m.netD.load_state_dict(detfill.fill_state_dict(m.netD.state_dict(), SEED_D))
    m.netG_teacher.eval()
    return m


def main():
    opt = spade_opt()
    h, w = int(opt.crop_size / opt.aspect_ratio), opt.crop_size
    n = 2
    lab, ins, img = synth_inputs(n, h, w, opt.input_nc, SEED_X)
    out = dict(label=lab.astype(np.int16), instance=ins, image_seed=SEED_X + 1, h=h, w=w, n=n,
               opt=json.dumps({k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, bool, list, type(None)))}))

    # ---- A19: preprocess_input / get_edges ------------------------------------------------------------------------
    sm = SPADEModel.__new__(SPADEModel)
    sm.opt, sm.device = opt, torch.device('cpu')
    sem, real_B = sm.preprocess_input({'label': torch.from_numpy(lab).float(), 'instance': torch.from_numpy(ins), 'image': img})
    out['sem_checks'] = checks(sem)
    out['sem_edge'] = sem[:, -1].numpy().astype(np.uint8)

    m = build(opt)
    out['T_shapes'], out['S_shapes'], out['D_shapes'] = [shapes_json(x.state_dict()) for x in (m.netG_teacher, m.netG_student, m.netD)]
    out['V_shapes'] = shapes_json(m.criterionVGG.vgg.state_dict())
    vmap = {}   # reference VGG19 keys ('slice1.0.weight') -> torchvision feature index keys ('0.weight')
    for k in m.criterionVGG.vgg.state_dict():
        vmap[k] = k.split('.', 1)[1]
    out['V_keymap'] = json.dumps(vmap)

    # ---- A14-A17: generator forwards --------------------------------------------------------------------------------
    layers = m.mapping_layers
    with torch.no_grad():
        Tfake, Tacts = m.netG_teacher(sem, mapping_layers=layers)
    snap = copy.deepcopy(m.netG_student.state_dict())
    with torch.no_grad():
        Sfake, Sacts = m.netG_student(sem, mapping_layers=layers)
    out['Tfake_sub'], out['Tfake_checks'] = sub(Tfake), checks(Tfake)
    out['Sfake_sub'], out['Sfake_checks'] = sub(Sfake), checks(Sfake)
    for name in layers:
        out[f'Tact_{name}'] = checks(Tacts[name])
        out[f'Sact_{name}'] = checks(Sacts[name])
    sd1 = m.netG_student.state_dict()
    out['S_rm_head'] = sd1['head_0.spade.param_free_norm.running_mean'].numpy().copy()
    out['S_rv_up3'] = sd1['up_3.res_ops.1.0.norm.running_var'].numpy().copy()
    m.netG_student.load_state_dict(snap)      # the forward above advanced the running statistics

    # model_profiling (n_macs) of both generators -- used by the shrink fixtures
    for tag, net in (('T', m.netG_teacher), ('S', m.netG_student)):
        net_c = copy.deepcopy(net)
        model_profiling(net_c, h, w, channel=opt.semantic_nc, num_forwards=0, verbose=False, use_cuda=False)
        out[f'{tag}_n_macs'] = np.int64(net_c.n_macs)

    # ---- A18: discriminator on a fixed 2N batch (training mode: one power iteration) -------------------------------
    d_snap = copy.deepcopy(m.netD.state_dict())
    with torch.no_grad():
        fr = torch.cat([torch.cat([sem, Sfake], 1), torch.cat([sem, real_B], 1)], 0)
        dout = m.netD(fr)
    for i, scale in enumerate(dout):
        for j, t in enumerate(scale):
            out[f'D_{i}_{j}'] = checks(t)
    out['D_u_after'] = m.netD.state_dict()['discriminator_0.model2.0.0.weight_u'].numpy().copy()
    m.netD.load_state_dict(d_snap)

    # ---- A13: one optimize_parameters ---------------------------------------------------------------------------------
    G_params = list(m.netG_student.parameters())
    for netA in m.netAs:
        G_params += list(netA.parameters())
    opt_G = torch.optim.Adam(G_params, lr=opt.lr / 2, betas=(0.0, 0.9))
    opt_D = torch.optim.Adam(list(m.netD.parameters()), lr=opt.lr * 2, betas=(0.0, 0.9))
    m.train()
    m.netG_teacher.eval()
    for p in m.netD.parameters():
        p.requires_grad_(False)
    opt_G.zero_grad()
    losses = m(sem, real_B, mode='G_loss')
    losses['loss_G'].mean().backward()
    g_losses = {k: float(v.detach().mean()) for k, v in losses.items()}
    gS = {k: v.grad.clone() for k, v in m.netG_student.named_parameters() if v.grad is not None}
    opt_G.step()
    for p in m.netD.parameters():
        p.requires_grad_(True)
    opt_D.zero_grad()
    losses = m(sem, real_B, mode='D_loss')
    losses['loss_D'].mean().backward()
    d_losses = {k: float(v.detach().mean()) for k, v in losses.items()}
    gD = {k: v.grad.clone() for k, v in m.netD.named_parameters() if v.grad is not None}
    opt_D.step()
    out['losses'] = json.dumps({**g_losses, **d_losses})
    out['S_gmax'] = np.float64(max(float(v.abs().max()) for v in gS.values()))
    out['D_gmax'] = np.float64(max(float(v.abs().max()) for v in gD.values()))
    sdS, sdD = m.netG_student.state_dict(), m.netD.state_dict()
    probe_S = ['fc.weight', 'fc_norm.weight', 'head_0.spade.res_ops.1.0.conv.weight', 'head_0.spade.dw_ops.2.2.bias',
               'G_middle_1.res_ops.2.1.conv.weight', 'up_1.dw_ops.1.1.conv.weight', 'up_1.shortcut.0.weight', 'up_1.shortcut.1.conv.weight',
               'up_3.spade.dw_ops.0.1.norm.bias', 'conv_img.weight', 'conv_img.bias']
    probe_D = ['discriminator_0.model0.0.weight', 'discriminator_0.model2.0.0.weight_orig', 'discriminator_0.model4.0.bias',
               'discriminator_1.model1.0.0.weight_orig', 'discriminator_1.model3.0.0.weight_orig', 'discriminator_1.model4.0.weight']
    out['probe_S'], out['probe_D'] = json.dumps(probe_S), json.dumps(probe_D)
    for k in probe_S:
        out['S_after/' + k] = sdS[k].numpy().reshape(-1)[:256].copy()
        out['S_grad/' + k] = gS[k].numpy().reshape(-1)[:256].copy()
        out['S_gnorm/' + k] = np.float64(gS[k].double().norm().item())
    for k in probe_D:
        out['D_after/' + k] = sdD[k].numpy().reshape(-1)[:256].copy()
        out['D_grad/' + k] = gD[k].numpy().reshape(-1)[:256].copy()
        out['D_gnorm/' + k] = np.float64(gD[k].double().norm().item())
    out['D_u_step'] = sdD['discriminator_1.model2.0.0.weight_u'].numpy().copy()
    out['S_rv_step'] = sdS['G_middle_0.spade.param_free_norm.running_var'].numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'spade_step.npz'), **out)
    print('spade_step.npz', {k: round(v, 5) for k, v in {**g_losses, **d_losses}.items()})
    print('n_macs T/S', int(out['T_n_macs']), int(out['S_n_macs']))


if __name__ == '__main__' and os.environ.get('GOLDEN_STEP', '1') == '1':
    main()


def shrink_golden():
    """shrink_spade_model (utils/common.py:710-869) run for real on a seeded teacher whose norm scales are |N(0,1)|."""
    from argparse import Namespace
    uc.Adam = lambda params, lr, betas: torch.optim.Adam(params, lr=lr, betas=(float(betas[0]), float(betas[1])))   # torch 2.10 rejects int 0
    out = {}
    for tag, target, lb in (('a', 0.8e9, 1), ('b', 0.35e9, 2)):
        opt = spade_opt(target_flops=target, prune_cin_lb=lb, data_height=128, data_width=256, data_channel=6, lr_policy='linear')
        m = build(opt)
        m.netG_teacher.load_state_dict(detfill.fill_state_dict(m.netG_teacher.state_dict(), SEED_T + 1, gamma_abs_normal=True))
        model = Namespace(modules_on_one_gpu=m, device=torch.device('cpu'), isTrain=True,
                          optimizer_D=torch.optim.Adam(m.netD.parameters(), lr=1e-4, betas=(0.0, 0.9)))
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            uc.shrink_spade_model(model, target, opt)
        line = [l for l in buf.getvalue().splitlines() if l.startswith('scale threshold')][0]
        thr = float(line.split('scale threshold: ')[1].split(',')[0])
        S = m.netG_student
        cfg = {'fc': S.fc.out_channels, 'blocks': {}}
        for name, blk in S.get_named_block_list().items():
            cfg['blocks'][name] = dict(input_dim=blk.input_dim, output_dim=blk.output_dim, res=blk.res_channels, dw=blk.dw_channels,
                                       spade_res=blk.spade.res_channels, spade_dw=blk.spade.dw_channels,
                                       shortcut=blk.shortcut is not None)
        out[f'{tag}_target'], out[f'{tag}_lb'] = target, lb
        out[f'{tag}_thr'] = np.float32(thr)
        out[f'{tag}_n_macs'] = np.int64(S.n_macs)
        out[f'{tag}_cfg'] = json.dumps(cfg)
        out[f'{tag}_netA'] = np.array([a.in_channels for a in m.netAs] + [a.out_channels for a in m.netAs])
        out[f'{tag}_S_shapes'] = shapes_json(S.state_dict())
        out['T_shapes'] = shapes_json(m.netG_teacher.state_dict())
        out['opt'] = json.dumps({k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, bool, list, type(None)))})
        print('shrink', tag, thr, int(S.n_macs), cfg['fc'], cfg['blocks']['up_1'])
    np.savez_compressed(os.path.join(OUT, 'spade_shrink.npz'), **out)


def weight_transfer_golden():
    """load_pretrained_weight('inception_spade', ...) (utils/weight_transfer.py:137-212, 267-288) run for real: ngf 8 -> 6, seeded weights.
    Recorded: every student tensor's shape AFTER the transfer (the reference replaces the gamma|beta convs by their gamma rows) and exact
    fingerprints of all tensors; five tensors in full."""
    from utils.weight_transfer import load_pretrained_weight
    opt = spade_opt(data_height=128, data_width=256, data_channel=6)
    def G(ngf):
        o = copy.deepcopy(opt)
        o.ngf, o.norm_G = ngf, 'spadesyncbatch3x3'
        return networks.define_G(opt.input_nc, 3, ngf, 'inception_spade', 'instance', 0, 'xavier', 0.02, [], opt=o)
    A, B = G(8), G(6)
    A.load_state_dict(detfill.fill_state_dict(A.state_dict(), 601, gamma_abs_normal=True))
    B.load_state_dict(detfill.fill_state_dict(B.state_dict(), 602))
    out = {'A_shapes': shapes_json(A.state_dict()), 'B_shapes_before': shapes_json(B.state_dict())}
    load_pretrained_weight('inception_spade', 'inception_spade', A, B, 8, 6)
    sd = B.state_dict()
    out['B_shapes_after'] = shapes_json(sd)
    fp = []
    for k, v in sd.items():
        d = v.double().reshape(-1)
        w = torch.arange(1, d.numel() + 1, dtype=torch.float64)
        fp.append([float(d.sum()), float((d * w).sum()), float((d * d).sum())])
    out['B_fingerprints'] = np.array(fp, dtype=np.float64)
    for k in ('fc.weight', 'fc_norm.weight', 'head_0.spade.res_ops.1.0.conv.weight', 'head_0.spade.res_ops.1.1.weight', 'up_1.shortcut.1.conv.weight',
              'up_1.res_ops.1.0.conv.weight', 'conv_img.weight'):
        out['B:' + k] = sd[k].numpy().copy()
    out['opt'] = json.dumps({k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, bool, list, type(None)))})
    np.savez_compressed(os.path.join(OUT, 'spade_weight_transfer.npz'), **out)
    changed = sum(1 for (k, a), (_, b) in zip(json.loads(out['B_shapes_before']), json.loads(out['B_shapes_after'])) if a != b)
    print('spade_weight_transfer.npz', len(sd), 'tensors,', changed, 'changed shape')


if __name__ == '__main__' and os.environ.get('GOLDEN_TRANSFER', '1') == '1':
    weight_transfer_golden()


if __name__ == '__main__' and os.environ.get('GOLDEN_SHRINK', '1') == '1':
    shrink_golden()


def spadeinstance_golden():
    """norm_G = 'spadeinstance3x3' (inception_modules.py:407-423: the block's hidden norms, its shortcut norm and the SPADE layers' param-free
    norm become nn.InstanceNorm2d; the gamma|beta nets keep SynchronizedBatchNorm2d, :598): one train-mode generator forward + the gradient of
    sum(out * g) w.r.t. four parameters, ngf 6, 2 x 128 x 256, seeded weights.  No launch script uses the option; the vectors pin the oracle
    (and, once built, the HIP path) for it."""
    opt = spade_opt(data_height=128, data_width=256, data_channel=6)
    o = copy.deepcopy(opt)
    o.ngf, o.norm_G = 6, 'spadeinstance3x3'
    G = networks.define_G(opt.input_nc, 3, 6, 'inception_spade', 'instance', 0, 'xavier', 0.02, [], opt=o)
    G.load_state_dict(detfill.fill_state_dict(G.state_dict(), 701, gamma_abs_normal=True))
    G.train()
    h, w, n = 128, 256, 2
    lab, ins, img = synth_inputs(n, h, w, opt.input_nc, 711)
    sm = SPADEModel.__new__(SPADEModel)
    sm.opt, sm.device = opt, torch.device('cpu')
    sem, _ = sm.preprocess_input({'label': torch.from_numpy(lab).float(), 'instance': torch.from_numpy(ins), 'image': img})
    out = {'shapes': shapes_json(G.state_dict()), 'label': lab.astype(np.int16), 'instance': ins, 'h': h, 'w': w, 'n': n}
    y = G(sem)
    g = detfill.normal(tuple(y.shape), 712)
    (y * g).sum().backward()
    out['y_sub'] = y.detach()[:, :, ::2, ::2].contiguous().numpy()
    out['y_checks'] = checks(y)
    params = dict(G.named_parameters())
    for k in ('fc.weight', 'head_0.res_ops.1.0.norm.weight', 'up_2.spade.dw_ops.1.0.conv.weight', 'up_3.shortcut.0.weight', 'up_3.dw_ops.0.2.conv.weight'):
        out['grad:' + k] = params[k].grad.numpy().copy()
    out['state_after'] = json.dumps(sorted(k for k in G.state_dict() if 'running' in k))      # only the SyncBN layers carry statistics
    for k in ('fc_norm.running_mean', 'up_3.spade.res_ops.0.0.norm.running_var'):
        out['buf:' + k] = G.state_dict()[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'spade_instance_fwd.npz'), **out)
    print('spade_instance_fwd.npz', tuple(y.shape), len(params), 'parameters')


if __name__ == '__main__' and os.environ.get('GOLDEN_INSTANCE', '1') == '1':
    spadeinstance_golden()
