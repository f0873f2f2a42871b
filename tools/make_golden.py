"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (snap-research/CAT imported from /root/reference on CPU).

Run in the build container only:  python tools/make_golden.py
The fixtures hold inputs (or the seeds that rebuild them via oracle/detfill.py) and the reference's outputs; they pin
the CPU oracle (oracle/ref_cpu.py) and, through it, the HIP path.  No reference source is stored."""
import itertools
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
from models import networks  # noqa: E402  (reference)
from models.modules.loss import GANLoss  # noqa: E402
from utils import common as uc  # noqa: E402
from utils.model_profiling import model_profiling  # noqa: E402
from distillers.inception_distiller import InceptionDistiller  # noqa: E402
from torch.optim import Adam  # noqa: E402

from oracle import detfill  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

SEED_T, SEED_S, SEED_D, SEED_A, SEED_X = 11, 21, 41, 51, 31


def sub(t, cmax=8, step=8):
    """small deterministic sub-sample of an NCHW tensor"""
    return t.detach()[:, :cmax, ::step, ::step].contiguous().numpy()


def shapes_json(sd):
    return json.dumps([[k, list(v.shape)] for k, v in sd.items()])


def teacher(opt):
    T = networks.define_G(3, 3, 64, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
    T.load_state_dict(detfill.fill_state_dict(T.state_dict(), SEED_T, gamma_abs_normal=True))
    T.eval()
    return T


def ref_distiller(opt, T, D):
    """InceptionDistiller without its dataset / FID constructor (SURVEY §8c recipe step 5)."""
    m = InceptionDistiller.__new__(InceptionDistiller)
    m.opt, m.gpu_ids, m.isTrain, m.device = opt, [], True, torch.device('cpu')
    m.loss_names = ['G_gan', 'G_distill', 'G_recon', 'D_fake', 'D_real']
    m.netG_teacher, m.netD = T, D
    m.netG_student = networks.define_G(3, 3, opt.student_ngf, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
    m.criterionGAN = GANLoss(opt.gan_mode)
    m.criterionRecon = torch.nn.L1Loss()
    m.mapping_layers = ['down_sampling.9'] + ['features.%d' % i for i in range(2, 11, 3)]
    m.netAs, m.Tacts, m.Sacts = [], {}, {}
    gp = []
    for i, n in enumerate(m.mapping_layers):
        netA = nn.Conv2d(opt.student_ngf * 4, opt.teacher_ngf * 4, 1)
        gp.append(netA.parameters())
        m.netAs.append(netA)
        m.loss_names.append('G_distill%d' % i)
    m.optimizer_G = Adam([{'params': m.netG_student.parameters()}, {'params': itertools.chain(*gp)}], lr=opt.lr, betas=(opt.beta1, 0.999))
    m.optimizer_D = Adam(D.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
    m.optimizers = [m.optimizer_G, m.optimizer_D]
    m.add_mapping_hook()
    return m


def gamma_dump(T):
    g = {'down': [T.down_sampling[i].weight.detach().clone() for i in (2, 5, 8)], 'up': [T.up_sampling[i].weight.detach().clone() for i in (1, 4)],
         'blocks': []}
    for blk in T.features:
        g['blocks'].append(([bn.weight.detach().clone() for bn in blk.get_first_res_bn()],
                            [bn.weight.detach().clone() for bn in blk.get_first_dw_bn()]))
    return g


def run_config(tag, norm, track, target, dataset_mode, gan_mode, ndf, lam_recon, lam_distill, size, nbatch, distill='ka', keep=('shrink', 'forward', 'step')):
    opt = ref_import.make_opt(norm=norm, track=track, target_flops=target, dataset_mode=dataset_mode, gan_mode=gan_mode, ndf=ndf,
                              lambda_recon=lam_recon, lambda_distill=lam_distill, distill_G_loss_type=distill)
    T = teacher(opt)
    d_in = 6 if dataset_mode == 'aligned' else 3
    D = networks.define_D(d_in, ndf, 'n_layers', 3, norm, 'normal', 0.02, [], opt=opt)
    D.load_state_dict(detfill.fill_state_dict(D.state_dict(), SEED_D))
    m = ref_distiller(opt, T, D)

    # ---- shrink (utils/common.py:315-707) --------------------------------------------------------------
    model_profiling(T, 256, 256, use_cuda=False, num_forwards=0, verbose=False)
    t_macs = [T.n_macs, T.down_sampling.n_macs, T.features.n_macs, T.up_sampling.n_macs]
    gam = gamma_dump(T)
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        uc.shrink_model(m, target, opt)
    line = [l for l in buf.getvalue().splitlines() if l.startswith('scale threshold')][0]
    thr = np.float32(line.split('scale threshold: ')[1].split(',')[0])
    S = m.netG_student
    s_macs = [S.n_macs, S.down_sampling.n_macs, S.features.n_macs, S.up_sampling.n_macs]
    cfg = {'down': [S.down_sampling[i].num_features for i in (2, 5, 8)], 'up': [S.up_sampling[i].num_features for i in (1, 4)],
           'blocks': [[b.res_channels, b.dw_channels] for b in S.features]}
    sh = {'thr': thr, 'target': np.float64(target), 't_macs': np.array(t_macs, dtype=np.int64), 's_macs': np.array(s_macs, dtype=np.int64),
          'cfg': json.dumps(cfg), 'student_shapes': shapes_json(S.state_dict()), 'prune_cin_lb': np.int64(opt.prune_cin_lb),
          'norm': norm, 'track': np.bool_(track)}
    for i, g in enumerate(gam['down']):
        sh[f'g_down{i}'] = g.numpy()
    for i, g in enumerate(gam['up']):
        sh[f'g_up{i}'] = g.numpy()
    for b, (res, dw) in enumerate(gam['blocks']):
        for j, g in enumerate(res):
            sh[f'g_b{b}_res{j}'] = g.numpy()
        for j, g in enumerate(dw):
            sh[f'g_b{b}_dw{j}'] = g.numpy()
    # copied (masked) weights: a few tensors prove the mask / index selection bit-exactly
    ssd = S.state_dict()
    for k in ['down_sampling.1.weight', 'down_sampling.4.weight', 'down_sampling.8.weight', 'features.0.res_ops.1.1.0.weight',
              'features.0.res_ops.1.4.weight', 'features.3.dw_ops.2.2.0.weight', 'features.8.dw_ops.0.4.weight', 'up_sampling.0.weight',
              'up_sampling.3.weight', 'up_sampling.7.weight']:
        if k in ssd:
            sh['copied:' + k] = ssd[k].numpy().copy()
    if 'shrink' in keep:
        np.savez_compressed(os.path.join(OUT, f'shrink_{tag}.npz'), **sh)
    print(tag, 'shrink: thr', thr, 'student macs', s_macs[0], 'down', cfg['down'], 'up', cfg['up'], 'block0', cfg['blocks'][0])

    # ---- student forward (trainer.py:106-107 re-initialises the pruned student; here: deterministic fill) --------
    S.load_state_dict(detfill.fill_state_dict(S.state_dict(), SEED_S))
    for i, a in enumerate(m.netAs):
        a.load_state_dict(detfill.fill_state_dict(a.state_dict(), SEED_A + i))
    S.train()
    x = detfill.images((1, 3, 256, 256), SEED_X)
    m.Sacts.clear()
    s_before = {k: v.clone() for k, v in S.state_dict().items()}
    with torch.no_grad():
        y = S(x)
    S.load_state_dict(s_before)     # a train-mode forward moves BatchNorm running stats; the step fixture starts fresh
    fw = {'out': sub(y, 3, 8), 'student_shapes': shapes_json(S.state_dict())}
    for k, v in m.Sacts.items():
        fw['act:' + k.replace('cpu', '')] = sub(v, 8, 8)
    m.Tacts.clear()
    xt = detfill.images((1, 3, size, size), SEED_X + 1)
    with torch.no_grad():
        yt = T(xt)
    fw['teacher_out'] = sub(yt, 3, 4)
    for k, v in m.Tacts.items():
        fw['tact:' + k.replace('cpu', '')] = sub(v, 8, 4)
    # discriminator forward (train mode)
    xd = detfill.images((nbatch, d_in, size, size), SEED_X + 2)
    D.train()
    d_before = {k: v.clone() for k, v in D.state_dict().items()}
    with torch.no_grad():
        fw['disc_out'] = D(xd).numpy()
    D.load_state_dict(d_before)
    if 'forward' in keep:
        np.savez_compressed(os.path.join(OUT, f'forward_{tag}.npz'), **fw)

    # ---- two full optimize_parameters steps (inception_distiller.py:179-188) ----------------------------------
    st = {}
    probe_S = ['down_sampling.1.weight', 'down_sampling.2.weight', 'features.0.res_ops.2.1.0.weight', 'features.4.dw_ops.1.2.0.weight',
               'features.8.pw_bn.bias', 'up_sampling.0.weight', 'up_sampling.7.weight', 'up_sampling.7.bias']
    probe_D = ['model.0.weight', 'model.0.bias', 'model.2.weight', 'model.3.weight', 'model.8.weight', 'model.11.weight']
    for step in range(2):
        A = detfill.images((nbatch, 3, size, size), SEED_X + 10 + step)
        B = detfill.images((nbatch, 3, size, size), SEED_X + 20 + step)
        m.set_input({'A': A, 'B': B, 'A_paths': [], 'B_paths': []})
        m.optimize_parameters(step)
        for k, v in m.get_current_losses().items():
            st[f'loss{step}:{k}'] = np.float64(v)
        st[f'Sfake{step}'] = sub(m.Sfake_B, 3, 4)
        ssd, dsd = S.state_dict(), D.state_dict()
        for k in probe_S:
            st[f'S{step}:{k}'] = ssd[k].reshape(-1)[:64].numpy().copy()
        for k in probe_D:
            if k in dsd:
                st[f'D{step}:{k}'] = dsd[k].reshape(-1)[:64].numpy().copy()
        for k in dsd:
            if k.endswith('running_mean') or k.endswith('running_var'):
                st[f'D{step}:{k}'] = dsd[k].reshape(-1)[:16].numpy().copy()
        for k in ssd:
            if k in ('down_sampling.2.running_mean', 'down_sampling.2.running_var', 'features.0.pw_bn.running_var'):
                st[f'S{step}:{k}'] = ssd[k].reshape(-1)[:16].numpy().copy()
        for i, a in enumerate(m.netAs):        # the 1x1 adaptors only train with distill_G_loss_type='mse'
            st[f'A{step}:{i}.weight'] = a.weight.detach().reshape(-1)[:64].numpy().copy()
            st[f'A{step}:{i}.bias'] = a.bias.detach().reshape(-1)[:64].numpy().copy()
    st['meta'] = json.dumps(dict(norm=norm, track=track, dataset_mode=dataset_mode, gan_mode=gan_mode, ndf=ndf, lambda_recon=lam_recon,
                                 lambda_distill=lam_distill, lambda_gan=1.0, size=size, nbatch=nbatch, lr=opt.lr, beta1=opt.beta1,
                                 target=target, distill=distill))
    st['student_shapes'] = shapes_json(S.state_dict())
    np.savez_compressed(os.path.join(OUT, f'step_{tag}.npz'), **st)
    print(tag, 'step losses', {k: float(v) for k, v in st.items() if k.startswith('loss1')})


def small_ops():
    out = {}
    for n in (2, 3, 16):
        X = detfill.normal((n, 7, 6, 5), 100 + n).requires_grad_(True)
        Y = detfill.normal((n, 11, 6, 5), 200 + n)
        v = uc.KA(X, Y)
        v.backward()
        out[f'ka{n}'] = np.float64(v.item())
        out[f'ka{n}_grad'] = X.grad.numpy().copy()
    pred = detfill.normal((2, 1, 6, 6), 300, 1.5)
    for mode in ('hinge', 'lsgan', 'vanilla', 'wgangp'):
        crit = GANLoss(mode)
        for real in (True, False):
            p = pred.clone().requires_grad_(True)
            l = crit(p, real, for_discriminator=True)
            l.backward()
            out[f'gan_{mode}_D_{int(real)}'] = np.float64(l.item())
            out[f'gan_{mode}_D_{int(real)}_grad'] = p.grad.numpy().copy()
        p = pred.clone().requires_grad_(True)
        l = crit(p, True, for_discriminator=False)
        l.backward()
        out[f'gan_{mode}_G'] = np.float64(l.item())
        out[f'gan_{mode}_G_grad'] = p.grad.numpy().copy()
    # multiscale list form (loss.py:71-82)
    preds = [[detfill.normal((2, 4, 5, 5), 310), detfill.normal((2, 1, 5, 5), 311)], [detfill.normal((2, 4, 3, 3), 312), detfill.normal((2, 1, 3, 3), 313)]]
    crit = GANLoss('hinge')
    out['gan_hinge_list_D_1'] = crit(preds, True, True).numpy().copy()
    out['gan_hinge_list_D_0'] = crit(preds, False, True).numpy().copy()
    out['gan_hinge_list_G'] = crit(preds, True, False).numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'small_ops.npz'), **out)
    print('small ops done')


def teacher_training_steps():
    """Two optimize_parameters steps of Pix2PixModel and CycleGANModel (models/pix2pix_model.py:198-207, cycle_gan_model.py:292-303),
    constructed without their dataset / FID members like the distillers (SURVEY §8c recipe step 5)."""
    import random
    from models.pix2pix_model import Pix2PixModel
    from models.cycle_gan_model import CycleGANModel
    from utils.image_pool import ImagePool
    out = {}
    size, n = 64, 2
    # ---- pix2pix: BatchNorm (tracked), hinge, aligned -------------------------------------------------------------------------
    opt = ref_import.make_opt(norm='batch', track=True, ndf=32, gan_mode='hinge', lambda_recon=10.0, ngf=16, netG='inception_9blocks',
                              dropout_rate=0, direction='AtoB', lambda_comp_cost=0)
    m = Pix2PixModel.__new__(Pix2PixModel)
    m.opt, m.gpu_ids, m.isTrain, m.device = opt, [], True, torch.device('cpu')
    m.netG = networks.define_G(3, 3, opt.ngf, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
    m.netD = networks.define_D(6, opt.ndf, 'n_layers', 3, opt.norm, 'normal', 0.02, [], opt=opt)
    m.netG.load_state_dict(detfill.fill_state_dict(m.netG.state_dict(), 301))
    m.netD.load_state_dict(detfill.fill_state_dict(m.netD.state_dict(), 302))
    m.criterionGAN, m.criterionRecon = GANLoss(opt.gan_mode), torch.nn.L1Loss()
    m.optimizer_G = Adam(m.netG.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
    m.optimizer_D = Adam(m.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
    out['p2p_G_shapes'], out['p2p_D_shapes'] = shapes_json(m.netG.state_dict()), shapes_json(m.netD.state_dict())
    probes_G = ['down_sampling.1.weight', 'features.4.res_ops.2.1.0.weight', 'features.7.dw_ops.1.2.1.weight', 'up_sampling.7.weight']
    probes_D = ['model.0.weight', 'model.5.weight', 'model.11.weight']
    for step in range(2):
        A, B = detfill.images((n, 3, size, size), 310 + step), detfill.images((n, 3, size, size), 320 + step)
        m.set_input({'A': A, 'B': B, 'A_paths': [], 'B_paths': []})
        m.optimize_parameters(step)
        for k in ('G_gan', 'G_recon', 'D_real', 'D_fake'):
            out[f'p2p_loss{step}:{k}'] = np.float64(getattr(m, 'loss_' + k).item())
        out[f'p2p_fake{step}'] = sub(m.fake_B, 3, 4)
        for k in probes_G:
            out[f'p2p_G{step}:{k}'] = m.netG.state_dict()[k].reshape(-1)[:64].numpy().copy()
        for k in probes_D:
            out[f'p2p_D{step}:{k}'] = m.netD.state_dict()[k].reshape(-1)[:64].numpy().copy()
        out[f'p2p_G{step}:down_sampling.2.running_var'] = m.netG.state_dict()['down_sampling.2.running_var'].numpy().copy()
    out['p2p_meta'] = json.dumps(dict(norm='batch', track=True, ndf=32, ngf=16, gan_mode='hinge', lambda_recon=10.0, lambda_gan=1.0, lr=opt.lr,
                                      beta1=opt.beta1, size=size, nbatch=n))
    # ---- cycle_gan: InstanceNorm, lsgan, pool of 2 images (so the random swap path runs in step 2 and 3) ------------------------------
    opt = ref_import.make_opt(norm='instance', track=False, ndf=16, gan_mode='lsgan', ngf=8, netG='inception_9blocks', dropout_rate=0,
                              direction='AtoB', dataset_mode='unaligned', lambda_A=10.0, lambda_B=10.0, lambda_identity=0.5, pool_size=2)
    c = CycleGANModel.__new__(CycleGANModel)
    c.opt, c.gpu_ids, c.isTrain, c.device = opt, [], True, torch.device('cpu')
    for i, name in enumerate(('G_A', 'G_B')):
        net = networks.define_G(3, 3, opt.ngf, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
        net.load_state_dict(detfill.fill_state_dict(net.state_dict(), 401 + i))
        setattr(c, 'net' + name, net)
    for i, name in enumerate(('D_A', 'D_B')):
        net = networks.define_D(3, opt.ndf, 'n_layers', 3, opt.norm, 'normal', 0.02, [], opt=opt)
        net.load_state_dict(detfill.fill_state_dict(net.state_dict(), 411 + i))
        setattr(c, 'net' + name, net)
    c.fake_A_pool, c.fake_B_pool = ImagePool(opt.pool_size), ImagePool(opt.pool_size)
    c.criterionGAN, c.criterionCycle, c.criterionIdt = GANLoss(opt.gan_mode), torch.nn.L1Loss(), torch.nn.L1Loss()
    c.optimizer_G = Adam(itertools.chain(c.netG_A.parameters(), c.netG_B.parameters()), lr=opt.lr, betas=(opt.beta1, 0.999))
    c.optimizer_D = Adam(itertools.chain(c.netD_A.parameters(), c.netD_B.parameters()), lr=opt.lr, betas=(opt.beta1, 0.999))
    out['cyc_G_shapes'], out['cyc_D_shapes'] = shapes_json(c.netG_A.state_dict()), shapes_json(c.netD_A.state_dict())
    random.seed(1234)
    names = ['D_A', 'G_A', 'G_cycle_A', 'G_idt_A', 'D_B', 'G_B', 'G_cycle_B', 'G_idt_B']
    for step in range(3):
        A, B = detfill.images((1, 3, size, size), 420 + step), detfill.images((1, 3, size, size), 430 + step)
        c.set_input({'A': A, 'B': B})
        c.optimize_parameters(step)
        for k in names:
            out[f'cyc_loss{step}:{k}'] = np.float64(float(getattr(c, 'loss_' + k)))
        out[f'cyc_fakeB{step}'] = sub(c.fake_B, 3, 4)
        for nm, key in (('G_A', 'features.3.res_ops.1.1.0.weight'), ('G_B', 'up_sampling.7.weight'), ('D_A', 'model.0.weight'), ('D_B', 'model.8.weight')):
            out[f'cyc_{nm}{step}:{key}'] = getattr(c, 'net' + nm).state_dict()[key].reshape(-1)[:64].numpy().copy()
    out['cyc_meta'] = json.dumps(dict(norm='instance', ndf=16, ngf=8, gan_mode='lsgan', lambda_A=10.0, lambda_B=10.0, lambda_identity=0.5,
                                      pool_size=2, lr=opt.lr, beta1=opt.beta1, size=size, seed=1234))
    np.savez_compressed(os.path.join(OUT, 'train_steps.npz'), **out)
    print('train_steps.npz', {k: float(v) for k, v in out.items() if 'loss1' in k})


def weight_transfer_golden():
    """load_pretrained_weight (utils/weight_transfer.py:240-266) ngf 32 -> ngf 20, inception_9blocks, seeded pretrained weights."""
    from utils.weight_transfer import load_pretrained_weight
    opt = ref_import.make_opt(norm='instance', track=False)
    A = networks.define_G(3, 3, 32, 'inception_9blocks', 'instance', 0, 'normal', 0.02, [], opt=opt)
    B = networks.define_G(3, 3, 20, 'inception_9blocks', 'instance', 0, 'normal', 0.02, [], opt=opt)
    A.load_state_dict(detfill.fill_state_dict(A.state_dict(), 501))
    B.load_state_dict(detfill.fill_state_dict(B.state_dict(), 502))
    load_pretrained_weight('inception_9blocks', 'inception_9blocks', A, B, 32, 20)
    out = {'A_shapes': shapes_json(A.state_dict()), 'B_shapes': shapes_json(B.state_dict())}
    sd = B.state_dict()
    # pure index selection (no arithmetic): exact fingerprints of every tensor pin it bit for bit without shipping 3 MB of weights
    fp = []
    for k, v in sd.items():
        d = v.double().reshape(-1)
        w = torch.arange(1, d.numel() + 1, dtype=torch.float64)
        fp.append([float(d.sum()), float((d * w).sum()), float((d * d).sum())])
    out['B_fingerprints'] = np.array(fp, dtype=np.float64)
    for k in ('down_sampling.1.weight', 'features.2.res_ops.1.1.0.weight', 'features.5.dw_ops.2.2.0.weight', 'up_sampling.3.weight', 'up_sampling.7.weight'):
        out['B:' + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'weight_transfer.npz'), **out)
    print('weight_transfer.npz', len(sd), 'tensors')


def eval_utils_golden():
    """utils/util.py tensor2im / tensor2label / labelcolormap on seeded tensors: pins cat_amd/distillers/evaluation.py (image dumps of
    evaluate_model)."""
    from utils import util
    img = torch.tanh(detfill.normal((3, 12, 10), 77) * 1.5)
    img[0, 0, 0], img[1, 0, 1] = 1.0, -1.0
    lab = torch.zeros(37, 6, 7)
    idx = torch.from_numpy(np.random.default_rng(78).integers(0, 37, (6, 7)))
    lab.scatter_(0, idx[None], 1.0)
    out = {'img': img.numpy(), 'img_u8': util.tensor2im(img), 'gray_u8': util.tensor2im(img[:1]), 'lab': lab.numpy(),
           'lab_u8': util.tensor2label(lab, 37), 'cmap37': util.labelcolormap(37), 'cmap20': util.labelcolormap(20)}
    np.savez_compressed(os.path.join(OUT, 'eval_utils.npz'), **out)
    print('eval_utils.npz', {k: v.shape for k, v in out.items()})


def mse_step():
    # distill_G_loss_type='mse' (the flag's default, inception_distiller.py:113-132): MSE(netA(Sact), Tact) through the 1x1 adaptors
    run_config('mse', 'instance', False, 2.6e9, 'unaligned', 'lsgan', 64, 5.0, 1.0, 64, 2, distill='mse', keep=('step',))


def main_configs():
    small_ops()
    # C3-like: CycleGAN student (InstanceNorm affine, lsgan, unaligned, ndf 64, lambda_recon 5), SURVEY §8d
    run_config('in', 'instance', False, 2.6e9, 'unaligned', 'lsgan', 64, 5.0, 1.0, 64, 2)
    # C2-like: pix2pix student (BatchNorm + running stats, hinge, aligned, ndf 128, lambda_recon 100, lambda_distill 1.3)
    run_config('bn', 'batch', True, 4.6e9, 'aligned', 'hinge', 128, 100.0, 1.3, 64, 2)


PARTS = {'small': small_ops, 'eval': eval_utils_golden, 'transfer': weight_transfer_golden, 'train': teacher_training_steps, 'mse': mse_step,
         'main': main_configs}


if __name__ == '__main__':
    # `python tools/make_golden.py` regenerates EVERY inception-path fixture (round 4; the four extra parts used to need GOLDEN_ONLY=...
    # invocations).  Each part runs in its own interpreter: the reference's modules keep global state (option singletons, the RNG streams
    # the fixtures were recorded under), so a part must see a fresh import.  GOLDEN_ONLY=<part> still runs a single part.
    only = os.environ.get('GOLDEN_ONLY')
    if only:
        PARTS[only]()
        sys.exit(0)
    import subprocess
    for part in ('main', 'mse', 'eval', 'transfer', 'train'):
        print('== make_golden part:', part, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, GOLDEN_ONLY=part), check=True)
    print('== tools/make_golden_spade.py', flush=True)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'make_golden_spade.py')], check=True)
