"""ORACLE — test infrastructure only (never imported by the product package `cat_amd`).

A CPU restatement, in plain functional PyTorch (fp32, NCHW, stock ATen ops), of the part of snap-research/CAT that
the MI355X build accelerates: the InceptionDistiller training step.  Every function cites the reference file:line
it follows.  The restatement is PINNED against the real reference code: tools/make_golden.py imports
/root/reference in the build container, runs the reference classes on seeded inputs and stores inputs/outputs under
tests/golden/; tests/test_oracle_golden.py replays those vectors through this file (CPU, `-m "not gpu"`).
The reference itself has no tests or golden vectors for this path, so those fixtures are the pin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- networks
def _norm(sd, prefix, x, cfg, training, act=None, slope=0.2):
    """get_norm_layer (models/networks.py:29-64): InstanceNorm2d / BatchNorm2d, eps 1e-5, momentum 0.1."""
    w, b = sd.get(prefix + '.weight'), sd.get(prefix + '.bias')
    if cfg['norm'] == 'instance':
        y = F.instance_norm(x, None, None, w, b, True, cfg.get('momentum', 0.1), cfg.get('eps', 1e-5))
    elif cfg['norm'] == 'batch':
        rm, rv = sd.get(prefix + '.running_mean'), sd.get(prefix + '.running_var')
        use_batch = training or rm is None
        if use_batch and rm is not None and prefix + '.num_batches_tracked' in sd:
            sd[prefix + '.num_batches_tracked'] += 1
        y = F.batch_norm(x, rm if (rm is not None) else None, rv, w, b, use_batch, cfg.get('momentum', 0.1), cfg.get('eps', 1e-5))
    else:
        raise NotImplementedError(cfg['norm'])
    if act == 'relu':
        y = F.relu(y)
    elif act == 'lrelu':
        y = F.leaky_relu(y, slope)
    return y


def _rpad(x, p):
    return F.pad(x, (p, p, p, p), mode='reflect') if p > 0 else x


def inverted_residual_channels(sd, p, x, cfg, training):
    """InvertedResidualChannels.forward, models/modules/inception_modules.py:124-180,230-236."""
    outs = []
    j = 0
    while f'{p}.res_ops.{j}.1.0.weight' in sd:
        q = f'{p}.res_ops.{j}'
        w1 = sd[q + '.1.0.weight']
        k = w1.shape[-1]
        h = F.conv2d(_rpad(x, (k - 1) // 2), w1, sd.get(q + '.1.0.bias'))
        h = _norm(sd, q + '.1.1', h, cfg, training, 'relu')
        h = F.conv2d(_rpad(h, (k - 1) // 2), sd[q + '.4.weight'], sd.get(q + '.4.bias'))
        outs.append(h)
        j += 1
    j = 0
    while f'{p}.dw_ops.{j}.0.0.weight' in sd:
        q = f'{p}.dw_ops.{j}'
        h = F.conv2d(x, sd[q + '.0.0.weight'], sd.get(q + '.0.0.bias'))
        h = _norm(sd, q + '.0.1', h, cfg, training, 'relu')
        wd = sd[q + '.2.0.weight']
        k = wd.shape[-1]
        h = F.conv2d(_rpad(h, (k - 1) // 2), wd, sd.get(q + '.2.0.bias'), groups=wd.shape[0])
        h = _norm(sd, q + '.2.1', h, cfg, training, 'relu')
        h = F.conv2d(h, sd[q + '.4.weight'], sd.get(q + '.4.bias'))
        outs.append(h)
        j += 1
    if not outs:
        return x
    tmp = outs[0]
    for o in outs[1:]:
        tmp = tmp + o
    tmp = _norm(sd, p + '.pw_bn', tmp, cfg, training)
    return x + tmp


MAPPING_LAYERS = ['down_sampling.9', 'features.2', 'features.5', 'features.8']  # base_inception_distiller.py:183-190


def inception_generator(sd, x, cfg, training=True, n_blocks=9):
    """InceptionGenerator.forward, models/modules/inception_architecture/inception_generator.py:37-142.
    Returns (output, {mapping layer name: activation})."""
    acts = OrderedDict()
    h = F.conv2d(_rpad(x, 3), sd['down_sampling.1.weight'], sd.get('down_sampling.1.bias'))
    h = _norm(sd, 'down_sampling.2', h, cfg, training, 'relu')
    h = F.conv2d(h, sd['down_sampling.4.weight'], sd.get('down_sampling.4.bias'), stride=2, padding=1)
    h = _norm(sd, 'down_sampling.5', h, cfg, training, 'relu')
    h = F.conv2d(h, sd['down_sampling.7.weight'], sd.get('down_sampling.7.bias'), stride=2, padding=1)
    h = _norm(sd, 'down_sampling.8', h, cfg, training, 'relu')
    acts['down_sampling.9'] = h
    for i in range(n_blocks):
        h = inverted_residual_channels(sd, f'features.{i}', h, cfg, training)
        if f'features.{i}' in MAPPING_LAYERS:
            acts[f'features.{i}'] = h
    h = F.conv_transpose2d(h, sd['up_sampling.0.weight'], sd.get('up_sampling.0.bias'), stride=2, padding=1, output_padding=1)
    h = _norm(sd, 'up_sampling.1', h, cfg, training, 'relu')
    h = F.conv_transpose2d(h, sd['up_sampling.3.weight'], sd.get('up_sampling.3.bias'), stride=2, padding=1, output_padding=1)
    h = _norm(sd, 'up_sampling.4', h, cfg, training, 'relu')
    h = F.conv2d(_rpad(h, 3), sd['up_sampling.7.weight'], sd.get('up_sampling.7.bias'))
    return torch.tanh(h), acts


def nlayer_discriminator(sd, x, cfg, training=True, n_layers=3):
    """NLayerDiscriminator.forward, models/modules/discriminators.py:14-79 (4x4 convs, LeakyReLU 0.2)."""
    h = F.leaky_relu(F.conv2d(x, sd['model.0.weight'], sd.get('model.0.bias'), stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers):
        h = F.conv2d(h, sd[f'model.{idx}.weight'], sd.get(f'model.{idx}.bias'), stride=2, padding=1)
        h = _norm(sd, f'model.{idx + 1}', h, cfg, training, 'lrelu')
        idx += 3
    h = F.conv2d(h, sd[f'model.{idx}.weight'], sd.get(f'model.{idx}.bias'), stride=1, padding=1)
    h = _norm(sd, f'model.{idx + 1}', h, cfg, training, 'lrelu')
    idx += 3
    return F.conv2d(h, sd[f'model.{idx}.weight'], sd.get(f'model.{idx}.bias'), stride=1, padding=1)


# ---------------------------------------------------------------------------------------------- losses
def ka(X, Y):
    """utils/common.py:38-46."""
    X_ = X.reshape(X.size(0), -1)
    Y_ = Y.reshape(Y.size(0), -1)
    assert X_.shape[0] == Y_.shape[0]
    X_vec = X_ @ X_.T
    Y_vec = Y_ @ Y_.T
    return (X_vec * Y_vec).sum() / ((X_vec ** 2).sum() * (Y_vec ** 2).sum()) ** 0.5


def gan_loss(gan_mode, prediction, target_is_real, for_discriminator=True):
    """GANLoss.__call__, models/modules/loss.py:52-99 (lsgan, vanilla, wgangp, hinge incl. the multiscale list form)."""
    if gan_mode == 'lsgan':
        target = torch.tensor(1.0 if target_is_real else 0.0, dtype=prediction.dtype).expand_as(prediction)
        return F.mse_loss(prediction, target)
    if gan_mode == 'vanilla':       # nn.BCEWithLogitsLoss, loss.py:33-34
        target = torch.tensor(1.0 if target_is_real else 0.0, dtype=prediction.dtype).expand_as(prediction)
        return F.binary_cross_entropy_with_logits(prediction, target)
    if gan_mode == 'wgangp':        # loss.py:66-70
        return -prediction.mean() if target_is_real else prediction.mean()
    if gan_mode == 'hinge':
        if isinstance(prediction, list):
            loss = 0
            for pred_i in prediction:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                loss_tensor = gan_loss(gan_mode, pred_i, target_is_real, for_discriminator)
                bs = 1 if loss_tensor.dim() == 0 else loss_tensor.size(0)
                loss = loss + torch.mean(loss_tensor.view(bs, -1), dim=1)
            return loss / len(prediction)
        zero = torch.zeros_like(prediction)
        if for_discriminator:
            if target_is_real:
                return -torch.mean(torch.min(prediction - 1, zero))
            return -torch.mean(torch.min(-prediction - 1, zero))
        assert target_is_real
        return -torch.mean(prediction)
    raise NotImplementedError(gan_mode)


# ---------------------------------------------------------------------------------------------- optimizer
def adam_step(params, grads, state, lr, beta1, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (single-tensor form) as constructed at distillers/base_inception_distiller.py:205-214.
    Parameters whose gradient is None are skipped, as in torch."""
    for name, p in params.items():
        g = grads.get(name)
        if g is None:
            continue
        st = state.setdefault(name, {'step': 0, 'm': torch.zeros_like(p), 'v': torch.zeros_like(p)})
        st['step'] += 1
        t = st['step']
        st['m'].lerp_(g, 1 - beta1)
        st['v'].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
        denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(eps)
        p.data.addcdiv_(st['m'], denom, value=-lr / bc1)


# ---------------------------------------------------------------------------------------------- the step
class DistillState:
    """Everything `InceptionDistiller.optimize_parameters` reads and writes (inception_distiller.py:179-188)."""

    def __init__(self, teacher_sd, student_sd, d_sd, cfg, netA_sds=None):
        self.cfg = dict(cfg)
        self.T = {k: v.clone() for k, v in teacher_sd.items()}
        self.S = {k: v.clone() for k, v in student_sd.items()}
        self.D = {k: v.clone() for k, v in d_sd.items()}
        # the 1x1 adaptors netAs (base_inception_distiller.py:192-204): second param group of optimizer_G, used by 'mse' distillation
        self.A = {} if netA_sds is None else {f'{i}.{k}': v.clone() for i, sd in enumerate(netA_sds) for k, v in sd.items()}
        self.adam_G, self.adam_D = {}, {}
        self.losses = OrderedDict()

    @staticmethod
    def _is_param(k):
        return not (k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))

    def params(self, sd):
        return {k: v for k, v in sd.items() if self._is_param(k)}


def distill_step(st, real_A, real_B, n_shards=1):
    """One InceptionDistiller.optimize_parameters (distillers/inception_distiller.py:100-104,159-188 and
    base_inception_distiller.py:293-312): teacher+student forward, D step, G step (gan + recon + KA distill), two Adam
    updates.  n_shards > 1 restates nn.DataParallel's semantics for the loss: recon / GAN terms are means over
    the gathered batch, KA is computed per shard and summed (inception_distiller.py:136-148); norm statistics are
    per-shard."""
    cfg = st.cfg
    aligned = cfg['dataset_mode'] == 'aligned'
    gan_mode = cfg['gan_mode']
    for v in st.params(st.S).values():
        v.requires_grad_(True)
        v.grad = None
    shards_A = real_A.chunk(n_shards, 0)

    # forward() — inception_distiller.py:100-104
    with torch.no_grad():
        t_out = [inception_generator(st.T, a, cfg['T'], training=False) for a in shards_A]
    s_out = [inception_generator(st.S, a, cfg['S'], training=True) for a in shards_A]
    Tfake_B = torch.cat([o[0] for o in t_out], 0)
    Sfake_B = torch.cat([o[0] for o in s_out], 0)

    # backward_D — base_inception_distiller.py:293-312
    for v in st.params(st.D).values():
        v.requires_grad_(True)
        v.grad = None
    if aligned:
        fake = torch.cat((real_A, Sfake_B), 1).detach()
        real = torch.cat((real_A, real_B), 1).detach()
    else:
        fake, real = Sfake_B.detach(), real_B.detach()

    def netD(x):
        return torch.cat([nlayer_discriminator(st.D, c, cfg['D'], training=True) for c in x.chunk(n_shards, 0)], 0)

    loss_D_fake = gan_loss(gan_mode, netD(fake), False, True)
    loss_D_real = gan_loss(gan_mode, netD(real), True, True)
    loss_D = (loss_D_fake + loss_D_real) * 0.5
    loss_D.backward()
    pD = st.params(st.D)
    st.grads_D = {k: v.grad.clone() for k, v in pD.items() if v.grad is not None}
    adam_step(pD, {k: v.grad for k, v in pD.items()}, st.adam_D, cfg['lr'], cfg['beta1'])
    for v in pD.values():
        v.requires_grad_(False)
        v.grad = None

    # backward_G — inception_distiller.py:159-177
    if aligned:
        loss_G_recon = F.l1_loss(Sfake_B, real_B) * cfg['lambda_recon']
        fake = torch.cat((real_A, Sfake_B), 1)
    else:
        loss_G_recon = F.l1_loss(Sfake_B, Tfake_B) * cfg['lambda_recon']
        fake = Sfake_B
    loss_G_gan = gan_loss(gan_mode, netD(fake), True, for_discriminator=False) * cfg['lambda_gan']
    distill = []
    mse = cfg.get('distill_G_loss_type', 'ka') == 'mse'
    if mse:
        for v in st.A.values():
            v.requires_grad_(True)
            v.grad = None
    for i, name in enumerate(MAPPING_LAYERS):
        if mse:     # inception_distiller.py:113-132: per-device MSE(netA(Sact), Tact), summed over devices
            li = sum(F.mse_loss(F.conv2d(s[1][name], st.A[f'{i}.weight'], st.A[f'{i}.bias']), t[1][name]) for s, t in zip(s_out, t_out))
        else:
            li = sum(-ka(s[1][name], t[1][name]) for s, t in zip(s_out, t_out))
        distill.append(li)
    loss_G_distill = sum(distill) * cfg['lambda_distill']
    loss_G = loss_G_gan + loss_G_recon + loss_G_distill
    loss_G.backward()
    pS = st.params(st.S)
    adam_step(pS, {k: v.grad for k, v in pS.items()}, st.adam_G, cfg['lr'], cfg['beta1'])
    for v in pS.values():
        v.requires_grad_(False)
    if mse:
        pA = {'A.' + k: v for k, v in st.A.items()}
        adam_step(pA, {k: v.grad for k, v in pA.items()}, st.adam_G, cfg['lr'], cfg['beta1'])
        for v in st.A.values():
            v.requires_grad_(False)

    st.losses = OrderedDict(G_gan=float(loss_G_gan), G_distill=float(loss_G_distill), G_recon=float(loss_G_recon),
                            D_fake=float(loss_D_fake), D_real=float(loss_D_real))
    for i, d in enumerate(distill):
        st.losses['G_distill%d' % i] = float(d)
    st.Sfake_B, st.Tfake_B = Sfake_B.detach(), Tfake_B
    st.grads_S = {k: v.grad.clone() for k, v in pS.items() if v.grad is not None}
    return st.losses


# ---------------------------------------------------------------------------------------------- pruning
def conv_macs(cin, cout, k, ho, wo, groups=1, n=1):
    """module_profiling, utils/model_profiling.py:87-97 (ConvTranspose2d is counted by OUTPUT size, too)."""
    return (cin * cout * k * k * ho * wo // groups) * n


def generator_macs(cfg, height=256, width=256, input_nc=3, output_nc=3, norm_counts=True):
    """n_macs of an InceptionGenerator as utils/model_profiling.py:65-135 counts it, in closed form.
    cfg = dict(down=[c0,c1,c2], blocks=[(res_channels, dw_channels)]*9, up=[u0,u1], kernel_sizes=[...]).
    norm_counts: norm layers add C*H*W each unless they track running stats (model_profiling.py:106-135)."""
    c0, c1, c2 = cfg['down']
    ks = cfg['kernel_sizes']
    h, w = height, width
    nm = (lambda c, hh, ww: c * hh * ww) if norm_counts else (lambda c, hh, ww: 0)
    ds = conv_macs(input_nc, c0, 7, h, w) + nm(c0, h, w)
    ds += conv_macs(c0, c1, 3, h // 2, w // 2) + nm(c1, h // 2, w // 2)
    ds += conv_macs(c1, c2, 3, h // 4, w // 4) + nm(c2, h // 4, w // 4)
    hh, ww = h // 4, w // 4
    ft = 0
    for res, dw in cfg['blocks']:
        nbranch = 0
        for m, k in zip(res, ks):
            if m == 0:
                continue
            nbranch += 1
            ft += conv_macs(c2, m, k, hh, ww) + nm(m, hh, ww) + conv_macs(m, c2, k, hh, ww)
        for m, k in zip(dw, ks):
            if m == 0:
                continue
            nbranch += 1
            ft += conv_macs(c2, m, 1, hh, ww) + nm(m, hh, ww) + conv_macs(m, m, k, hh, ww, groups=m) + nm(m, hh, ww)
            ft += conv_macs(m, c2, 1, hh, ww)
        # an empty block returns x before any sub-module runs, so its pw_bn hook never fires; but the profiler's
        # InvertedResidualChannels rule then reads a stale/absent n_macs -- the reference never produces such blocks
        # for the budgets in its scripts, and shrink rebuilds pw_bn (fresh module, n_macs unset -> add_sub skips it).
        if nbranch > 0:
            ft += nm(c2, hh, ww)
    u0, u1 = cfg['up']
    us = conv_macs(c2, u0, 3, h // 2, w // 2) + nm(u0, h // 2, w // 2)
    us += conv_macs(u0, u1, 3, h, w) + nm(u1, h, w)
    us += conv_macs(u1, output_nc, 7, h, w)
    return dict(total=ds + ft + us, down_sampling=ds, features=ft, up_sampling=us)


def shrink_search(gammas, target_flops, prune_cin_lb=1, kernel_sizes=(1, 3, 5), height=256, width=256, norm_counts=True,
                  prune_cin_ub=float('inf'), prune_ft_cin_lb=1):
    """The threshold binary search of shrink_model, utils/common.py:341-443, on plain |gamma| vectors.
    gammas = dict(down=[g0,g1,g2], blocks=[([res gammas...], [dw gammas...])]*9, up=[g0,g1]) (1-D float32 tensors).
    All comparisons are done in float32 exactly as the reference does (tensor > tensor-scalar)."""
    allw = torch.cat([g.abs() for g in gammas['down']] + [g.abs() for blk in gammas['blocks'] for br in blk for g in br] +
                     [g.abs() for g in gammas['up']])
    lb, ub = allw.min(), allw.max()
    searched = float('inf')
    thr = None
    while (abs(ub - lb) > 1e-3 * lb) or (searched > target_flops):
        thr = (lb + ub) / 2
        cfg = shrink_config(gammas, thr, prune_cin_lb, kernel_sizes, prune_cin_ub, prune_ft_cin_lb)
        searched = generator_macs(cfg, height, width, norm_counts=norm_counts)['total']
        if searched > target_flops:
            lb = thr
        else:
            ub = thr
    return thr, searched, cfg


def shrink_config(gammas, thr, prune_cin_lb=1, kernel_sizes=(1, 3, 5), prune_cin_ub=float('inf'), prune_ft_cin_lb=1):
    """Channel counts kept at threshold `thr` (utils/common.py:349-388): count(|gamma| > thr) with the floors."""
    down = []
    for i, g in enumerate(gammas['down']):
        c = int((g.abs() > thr).sum().item())
        c = max(c, prune_cin_lb)
        if i == 0:
            c = min(c, prune_cin_ub)
        if i == len(gammas['down']) - 1:
            c = max(c, prune_ft_cin_lb)
        down.append(c)
    blocks = []
    for res, dw in gammas['blocks']:
        blocks.append(([int((g.abs() > thr).sum().item()) for g in res], [int((g.abs() > thr).sum().item()) for g in dw]))
    up = [max(int((g.abs() > thr).sum().item()), prune_cin_lb) for g in gammas['up']]
    return dict(down=down, blocks=blocks, up=up, kernel_sizes=list(kernel_sizes))


def shrink_masks(gammas, thr, prune_cin_lb=1):
    """Boolean keep-masks of the final weight-copy pass (utils/common.py:449-476, 521-541, 596-607): |gamma| > thr, or the
    top-`prune_cin_lb` channels (>= the lb-th largest) when fewer survive."""
    def mask_lb(g):
        m = g.abs() > thr
        if int(m.sum()) < prune_cin_lb:
            private = torch.sort(g.abs().view(-1), descending=True)[0][prune_cin_lb - 1]
            m = g.abs() >= private
        return m
    return dict(down=[mask_lb(g) for g in gammas['down']],
                blocks=[([g.abs() > thr for g in res], [g.abs() > thr for g in dw]) for res, dw in gammas['blocks']],
                up=[mask_lb(g) for g in gammas['up']])


# ---------------------------------------------------------------------------------------------- teacher-training steps
class TrainState:
    """Networks + Adam state of Pix2PixModel / CycleGANModel: dict name -> state_dict."""

    def __init__(self, nets, cfg):
        self.cfg = dict(cfg)
        self.nets = {n: {k: v.clone() for k, v in sd.items()} for n, sd in nets.items()}
        self.adam = {}
        self.losses = OrderedDict()

    @staticmethod
    def params(sd):
        return {k: v for k, v in sd.items() if not (k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))}

    def grad_on(self, names, on):
        for n in names:
            for v in self.params(self.nets[n]).values():
                v.requires_grad_(on)
                v.grad = None

    def adam_step(self, names, key):
        for n in names:
            p = {f'{n}.{k}': v for k, v in self.params(self.nets[n]).items()}
            adam_step(p, {k: v.grad for k, v in p.items()}, self.adam.setdefault(key, {}), self.cfg['lr'], self.cfg['beta1'])


def pix2pix_step(st, real_A, real_B):
    """Pix2PixModel.optimize_parameters (models/pix2pix_model.py:156-207): forward, D step 0.5*(fake+real), G step GAN + lambda*recon."""
    cfg = st.cfg
    G, D = st.nets['G'], st.nets['D']
    st.grad_on(['G'], True)
    fake_B, _ = inception_generator(G, real_A, cfg['G'], training=True)
    st.grad_on(['D'], True)
    loss_D_fake = gan_loss(cfg['gan_mode'], nlayer_discriminator(D, torch.cat((real_A, fake_B), 1).detach(), cfg['D'], True), False, True)
    loss_D_real = gan_loss(cfg['gan_mode'], nlayer_discriminator(D, torch.cat((real_A, real_B), 1).detach(), cfg['D'], True), True, True)
    ((loss_D_fake + loss_D_real) * 0.5).backward()
    st.adam_step(['D'], 'D')
    st.grad_on(['D'], False)
    loss_G_gan = gan_loss(cfg['gan_mode'], nlayer_discriminator(D, torch.cat((real_A, fake_B), 1), cfg['D'], True), True, False) * cfg['lambda_gan']
    recon = F.l1_loss if cfg.get('recon_loss_type', 'l1') == 'l1' else F.mse_loss
    loss_G_recon = recon(fake_B, real_B) * cfg['lambda_recon']
    (loss_G_gan + loss_G_recon).backward()
    st.grads_G = {k: v.grad.clone() for k, v in st.params(G).items() if v.grad is not None}
    st.adam_step(['G'], 'G')
    for v in st.params(G).values():
        v.requires_grad_(False)
    st.fake_B = fake_B.detach()
    st.losses = OrderedDict(G_gan=float(loss_G_gan), G_recon=float(loss_G_recon), D_real=float(loss_D_real), D_fake=float(loss_D_fake))
    return st.losses


class ImagePoolRef:
    """utils/image_pool.py:5-53 with an injected random source (same draw sequence as `random`)."""

    def __init__(self, pool_size, rng):
        self.pool_size, self.rng, self.num_imgs, self.images = pool_size, rng, 0, []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for image in images:
            image = image.detach().unsqueeze(0)
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
                out.append(image)
            elif self.rng.uniform(0, 1) > 0.5:
                i = self.rng.randint(0, self.pool_size - 1)
                tmp = self.images[i].clone()
                self.images[i] = image
                out.append(tmp)
            else:
                out.append(image)
        return torch.cat(out, 0)


def cyclegan_step(st, real_A, real_B, pools):
    """CycleGANModel.optimize_parameters (models/cycle_gan_model.py:226-303): forward (4 generator passes), G step (identity, GAN,
    cycle), D_A / D_B steps on pooled fakes.  pools = (fake_A_pool, fake_B_pool)."""
    cfg = st.cfg
    gc, dc, mode = cfg['G'], cfg['D'], cfg['gan_mode']
    GA, GB, DA, DB = (st.nets[n] for n in ('G_A', 'G_B', 'D_A', 'D_B'))
    lam_idt, lam_A, lam_B = cfg['lambda_identity'], cfg['lambda_A'], cfg['lambda_B']
    gen = lambda sd, x: inception_generator(sd, x, gc, training=True)[0]
    dis = lambda sd, x: nlayer_discriminator(sd, x, dc, training=True)
    st.grad_on(['G_A', 'G_B'], True)
    st.grad_on(['D_A', 'D_B'], False)
    fake_B = gen(GA, real_A)
    rec_A = gen(GB, fake_B)
    fake_A = gen(GB, real_B)
    rec_B = gen(GA, fake_A)
    if lam_idt > 0:
        loss_idt_A = F.l1_loss(gen(GA, real_B), real_B) * lam_B * lam_idt
        loss_idt_B = F.l1_loss(gen(GB, real_A), real_A) * lam_A * lam_idt
    else:
        loss_idt_A = loss_idt_B = torch.zeros(())
    loss_G_A = gan_loss(mode, dis(DA, fake_B), True)
    loss_G_B = gan_loss(mode, dis(DB, fake_A), True)
    loss_cycle_A = F.l1_loss(rec_A, real_A) * lam_A
    loss_cycle_B = F.l1_loss(rec_B, real_B) * lam_B
    (loss_G_A + loss_G_B + loss_cycle_A + loss_cycle_B + loss_idt_A + loss_idt_B).backward()
    st.adam_step(['G_A', 'G_B'], 'G')
    st.grad_on(['G_A', 'G_B'], False)
    st.grad_on(['D_A', 'D_B'], True)

    def d_basic(sd, real, fake):
        l_real = gan_loss(mode, dis(sd, real), True)
        l_fake = gan_loss(mode, dis(sd, fake.detach()), False)
        loss = (l_real + l_fake) * 0.5
        loss.backward()
        return loss
    loss_D_A = d_basic(DA, real_B, pools[1].query(fake_B))
    loss_D_B = d_basic(DB, real_A, pools[0].query(fake_A))
    st.adam_step(['D_A', 'D_B'], 'D')
    st.grad_on(['D_A', 'D_B'], False)
    st.fake_A, st.fake_B = fake_A.detach(), fake_B.detach()
    st.losses = OrderedDict(D_A=float(loss_D_A), G_A=float(loss_G_A), G_cycle_A=float(loss_cycle_A), G_idt_A=float(loss_idt_A),
                            D_B=float(loss_D_B), G_B=float(loss_G_B), G_cycle_B=float(loss_cycle_B), G_idt_B=float(loss_idt_B))
    return st.losses
