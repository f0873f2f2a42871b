"""ORACLE — test infrastructure only (never imported by the product package `cat_amd`).

CPU restatement (functional PyTorch, fp32, NCHW, stock ATen ops) of the GauGAN / SPADE distillation step of
snap-research/CAT: SPADEDistiller.optimize_parameters (distillers/base_spade_distiller.py:226-234) and everything below it
(SURVEY.md §8a rows A13-A19).  Every function cites the reference file:line it follows.  PINNED against the real reference:
tools/make_golden.py imports /root/reference in the build container, runs the reference classes on seeded inputs and
stores inputs/outputs in tests/golden/spade_*.npz; tests/test_oracle_golden.py replays them through this file.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .ref_cpu import adam_step, gan_loss, ka

EPS = 1e-5
MOMENTUM = 0.1
MAPPING_LAYERS = ['head_0', 'G_middle_1', 'up_1']          # base_spade_distiller_modules.py:74


# ---------------------------------------------------------------------------------------------- normalisation
def sync_bn(sd, p, x, training, synced=False, act=None):
    """SynchronizedBatchNorm2d.forward (models/modules/sync_batchnorm/batchnorm.py:68-140).
    synced=False: the single-device / eval path, F.batch_norm (:69-72).
    synced=True : the multi-replica training path on the WHOLE (gathered) batch: sum / square-sum -> mean, biased variance
                  clamped at eps (not +eps, :140), running stats with the unbiased variance (:132-137)."""
    w, b = sd.get(p + '.weight'), sd.get(p + '.bias')
    rm, rv = sd.get(p + '.running_mean'), sd.get(p + '.running_var')
    if rm is None:
        # norm_G = 'spadeinstance...' (inception_modules.py:414-415): the layer is nn.InstanceNorm2d (no running statistics in the
        # state dict, per-sample statistics in train and eval mode alike); its affine form carries weight / bias
        y = F.instance_norm(x, weight=w, bias=b, eps=EPS)
        return F.relu(y) if act == 'relu' else y
    if not (training and synced):
        y = F.batch_norm(x, rm, rv, w, b, training, MOMENTUM, EPS)
    else:
        n, c = x.shape[:2]
        xv = x.reshape(n, c, -1)
        size = xv.size(0) * xv.size(2)
        sum_ = xv.sum(dim=0).sum(dim=-1)
        ssum = (xv ** 2).sum(dim=0).sum(dim=-1)
        mean = sum_ / size
        sumvar = ssum - sum_ * mean
        unbias_var = sumvar / (size - 1)
        bias_var = sumvar / size
        with torch.no_grad():
            rm.copy_((1 - MOMENTUM) * rm + MOMENTUM * mean.detach())
            rv.copy_((1 - MOMENTUM) * rv + MOMENTUM * unbias_var.detach())
        inv_std = bias_var.clamp(EPS) ** -0.5
        if w is not None:
            y = (xv - mean.view(1, -1, 1)) * (inv_std * w).view(1, -1, 1) + b.view(1, -1, 1)
        else:
            y = (xv - mean.view(1, -1, 1)) * inv_std.view(1, -1, 1)
        y = y.view(x.shape)
    return F.relu(y) if act == 'relu' else y


def conv_sync_bn_relu(sd, p, x, training, synced):
    """ConvSyncBNReLU.forward (models/modules/inception_modules.py:280-313): conv (pad (k-1)//2) -> norm -> ReLU."""
    w = sd[p + '.conv.weight']
    k = w.shape[-1]
    groups = w.shape[0] if (w.shape[1] == 1 and x.shape[1] == w.shape[0] and w.shape[0] > 1) else 1
    h = F.conv2d(x, w, sd.get(p + '.conv.bias'), padding=(k - 1) // 2, groups=groups)
    return sync_bn(sd, p + '.norm', h, training, synced, act='relu')


def _conv_same(sd, p, x):
    w = sd[p + '.weight']
    return F.conv2d(x, w, sd.get(p + '.bias'), padding=(w.shape[-1] - 1) // 2)


def inception_spade(sd, p, x, segmap, training, synced):
    """InceptionSPADE.forward (inception_modules.py:746-762)."""
    normalized = sync_bn(sd, p + '.param_free_norm', x, training, synced)
    seg = F.interpolate(segmap, size=x.shape[2:], mode='nearest')
    outs = []
    j = 0
    while f'{p}.res_ops.{j}.0.conv.weight' in sd:
        q = f'{p}.res_ops.{j}'
        outs.append(_conv_same(sd, q + '.1', conv_sync_bn_relu(sd, q + '.0', seg, training, synced)))
        j += 1
    j = 0
    while f'{p}.dw_ops.{j}.0.conv.weight' in sd:
        q = f'{p}.dw_ops.{j}'
        h = conv_sync_bn_relu(sd, q + '.0', seg, training, synced)
        h = conv_sync_bn_relu(sd, q + '.1', h, training, synced)
        outs.append(_conv_same(sd, q + '.2', h))
        j += 1
    if not outs:
        return normalized
    tmp = outs[0]
    for o in outs[1:]:
        tmp = tmp + o
    c = x.shape[1]
    return normalized * (1 + tmp[:, :c]) + tmp[:, c:]


def spade_inverted_residual_channels(sd, p, x, seg, training, synced):
    """SPADEInvertedResidualChannels.forward (inception_modules.py:549-562); branch layout from _build (:428-505)."""
    def shortcut(t):
        if p + '.shortcut.1.conv.weight' in sd:
            return _conv_same(sd, p + '.shortcut.1.conv', sync_bn(sd, p + '.shortcut.0', t, training, synced))
        return t
    has_res = f'{p}.res_ops.0.0.conv.weight' in sd
    has_dw = f'{p}.dw_ops.0.0.conv.weight' in sd
    if not has_res and not has_dw:
        return shortcut(x)
    tmp = F.relu(inception_spade(sd, p + '.spade', x, seg, training, synced))
    outs = []
    j = 0
    while f'{p}.res_ops.{j}.0.conv.weight' in sd:
        q = f'{p}.res_ops.{j}'
        outs.append(_conv_same(sd, q + '.1.conv', conv_sync_bn_relu(sd, q + '.0', tmp, training, synced)))
        j += 1
    j = 0
    while f'{p}.dw_ops.{j}.0.conv.weight' in sd:
        q = f'{p}.dw_ops.{j}'
        h = conv_sync_bn_relu(sd, q + '.0', tmp, training, synced)
        h = conv_sync_bn_relu(sd, q + '.1', h, training, synced)
        outs.append(_conv_same(sd, q + '.2.conv', h))
        j += 1
    out = outs[0]
    for o in outs[1:]:
        out = out + o
    return out + shortcut(x)


def latent_size(crop_size, aspect_ratio, num_upsampling_layers='more'):
    """compute_latent_vector_size (inception_spade_generator.py:47-61)."""
    n_up = {'normal': 5, 'more': 6, 'most': 7}[num_upsampling_layers]
    sw = crop_size // (2 ** n_up)
    return sw, round(sw / aspect_ratio)


def inception_spade_generator(sd, seg, cfg, training=True, synced=False, mapping_layers=()):
    """InceptionSPADEGenerator.forward (models/modules/inception_architecture/inception_spade_generator.py:63-124).
    cfg: dict(crop_size, aspect_ratio, num_upsampling_layers).  Returns (image, {layer: activation})."""
    nul = cfg.get('num_upsampling_layers', 'more')
    sw, sh = latent_size(cfg['crop_size'], cfg['aspect_ratio'], nul)
    acts = OrderedDict()
    up = lambda t: F.interpolate(t, scale_factor=2, mode='nearest')
    x = F.interpolate(seg, size=(sh, sw))
    x = F.conv2d(x, sd['fc.weight'], sd['fc.bias'], padding=1)
    x = sync_bn(sd, 'fc_norm', x, training, synced)
    x = spade_inverted_residual_channels(sd, 'head_0', x, seg, training, synced)
    acts['head_0'] = x
    x = up(x)
    x = spade_inverted_residual_channels(sd, 'G_middle_0', x, seg, training, synced)
    acts['G_middle_0'] = x
    if nul in ('more', 'most'):
        x = up(x)
    x = spade_inverted_residual_channels(sd, 'G_middle_1', x, seg, training, synced)
    acts['G_middle_1'] = x
    for name in ('up_0', 'up_1', 'up_2', 'up_3') + (('up_4',) if nul == 'most' else ()):
        x = up(x)
        x = spade_inverted_residual_channels(sd, name, x, seg, training, synced)
        acts[name] = x
    x = F.conv2d(F.leaky_relu(x, 2e-1), sd['conv_img.weight'], sd['conv_img.bias'], padding=1)
    x = torch.tanh(x)
    return x, OrderedDict((k, v) for k, v in acts.items() if k in mapping_layers)


# ---------------------------------------------------------------------------------------------- discriminator
def spectral_norm_weight(sd, p, training, eps=1e-12):
    """torch.nn.utils.spectral_norm's SpectralNorm.compute_weight (the hook get_nonspade_norm_layer installs,
    spade_architecture/normalization.py:27-29): one power iteration on the persistent u, v in training mode (in place,
    no grad), sigma = u^T W v, weight = weight_orig / sigma."""
    w = sd[p + '.weight_orig']
    u, v = sd[p + '.weight_u'], sd[p + '.weight_v']
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def spade_nlayer_discriminator(sd, p, x, training=True, n_layers_D=4):
    """SPADENLayerDiscriminator.forward (models/modules/discriminators.py:129-182) with norm_D='spectralinstance'
    (normalization.py:17-50): returns the n_layers_D + 1 intermediate outputs."""
    outs = []
    h = F.leaky_relu(F.conv2d(x, sd[f'{p}.model0.0.weight'], sd[f'{p}.model0.0.bias'], stride=2, padding=2), 0.2)
    outs.append(h)
    for n in range(1, n_layers_D):
        stride = 1 if n == n_layers_D - 1 else 2
        w = spectral_norm_weight(sd, f'{p}.model{n}.0.0', training)
        h = F.conv2d(h, w, None, stride=stride, padding=2)
        h = F.leaky_relu(F.instance_norm(h, eps=EPS), 0.2)
        outs.append(h)
    h = F.conv2d(h, sd[f'{p}.model{n_layers_D}.0.weight'], sd[f'{p}.model{n_layers_D}.0.bias'], stride=1, padding=2)
    outs.append(h)
    return outs


def multiscale_discriminator(sd, x, training=True, num_D=2, n_layers_D=4):
    """MultiscaleDiscriminator.forward (discriminators.py:185-226)."""
    result = []
    for i in range(num_D):
        result.append(spade_nlayer_discriminator(sd, f'discriminator_{i}', x, training, n_layers_D))
        x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
    return result


# ---------------------------------------------------------------------------------------------- losses / inputs
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
VGG_SLICES = [2, 7, 12, 21, 30]     # models/modules/loss.py:159-168: features[0:2], [2:7], [7:12], [12:21], [21:30]
VGG_WEIGHTS = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]


def vgg19_features(vsd, x):
    """VGG19.forward (models/modules/loss.py:151-186): torchvision vgg19().features[0:30], tapped after relu1_1, relu2_1,
    relu3_1, relu4_1, relu5_1.  vsd: {'<idx>.weight', '<idx>.bias'} with torchvision's layer indices."""
    outs = []
    idx = 0
    h = x
    for v in VGG_CFG:
        if idx >= VGG_SLICES[-1]:
            break
        if v == 'M':
            h = F.max_pool2d(h, 2, 2)
            idx += 1
        else:
            h = F.relu(F.conv2d(h, vsd[f'{idx}.weight'], vsd[f'{idx}.bias'], padding=1))
            idx += 2
        if idx in VGG_SLICES:
            outs.append(h)
    return outs


def vgg_loss(vsd, x, y):
    """VGGLoss.forward (models/modules/loss.py:189-203)."""
    xf = vgg19_features(vsd, x)
    with torch.no_grad():
        yf = vgg19_features(vsd, y)
    loss = 0
    for i in range(len(xf)):
        loss = loss + VGG_WEIGHTS[i] * F.l1_loss(xf[i], yf[i].detach())
    return loss


def get_edges(t):
    """SPADEModel.get_edges (models/spade_model.py:169-179)."""
    edge = torch.zeros(t.size(), dtype=torch.uint8)
    edge[:, :, :, 1:] = edge[:, :, :, 1:] | (t[:, :, :, 1:] != t[:, :, :, :-1]).byte()
    edge[:, :, :, :-1] = edge[:, :, :, :-1] | (t[:, :, :, 1:] != t[:, :, :, :-1]).byte()
    edge[:, :, 1:, :] = edge[:, :, 1:, :] | (t[:, :, 1:, :] != t[:, :, :-1, :]).byte()
    edge[:, :, :-1, :] = edge[:, :, :-1, :] | (t[:, :, 1:, :] != t[:, :, :-1, :]).byte()
    return edge.float()


def preprocess_input(label, instance, input_nc, contain_dontcare_label=False, no_instance=False):
    """SPADEModel.preprocess_input (models/spade_model.py:142-161)."""
    label_map = label.long()
    bs, _, h, w = label_map.size()
    nc = input_nc + 1 if contain_dontcare_label else input_nc
    sem = torch.zeros([bs, nc, h, w]).scatter_(1, label_map, 1.0)
    if not no_instance:
        sem = torch.cat((sem, get_edges(instance)), dim=1)
    return sem


def divide_pred(pred):
    """spade_model_modules.py:143-155."""
    fake, real = [], []
    for p in pred:
        fake.append([t[:t.size(0) // 2] for t in p])
        real.append([t[t.size(0) // 2:] for t in p])
    return fake, real


# ---------------------------------------------------------------------------------------------- the step
class SpadeState:
    """What SPADEDistiller.optimize_parameters reads and writes."""

    def __init__(self, teacher_sd, student_sd, d_sd, vgg_sd, cfg, netA_sds=None):
        self.cfg = dict(cfg)
        # 1x1 adaptors netAs (base_spade_distiller_modules.py:75-87), trained only by distill_G_loss_type='mse'
        self.A = {} if netA_sds is None else {f'{i}.{k}': v.clone() for i, sd in enumerate(netA_sds) for k, v in sd.items()}
        self.T = {k: v.clone() for k, v in teacher_sd.items()}
        self.S = {k: v.clone() for k, v in student_sd.items()}
        self.D = {k: v.clone() for k, v in d_sd.items()}
        self.V = None if vgg_sd is None else {k: v.clone() for k, v in vgg_sd.items()}
        self.adam_G, self.adam_D = {}, {}
        self.losses = OrderedDict()

    @staticmethod
    def _is_param(k):
        return not k.endswith(('running_mean', 'running_var', 'num_batches_tracked', 'weight_u', 'weight_v'))

    def params(self, sd):
        return OrderedDict((k, v) for k, v in sd.items() if self._is_param(k))


def _shard_mean(fn, n_shards, *tensors):
    """DataParallel replicas each return their loss; backward_G/D take `.mean()` over replicas (spade_model.py:191,200)."""
    chunks = [t.chunk(n_shards, 0) for t in tensors]
    return sum(fn(*[c[r] for c in chunks]) for r in range(n_shards)) / n_shards


def spade_step(st, sem, real_B, n_shards=1):
    """One SPADEDistiller.optimize_parameters (base_spade_distiller.py:226-234): G step (compute_G_loss,
    base_spade_distiller_modules.py:128-156; calc_distill_loss, spade_distiller_modules.py:17-31) then D step
    (compute_D_loss, :158-175), Adam each (TTUR: betas (0, 0.9), lr/2 and lr*2, :91-107).

    n_shards > 1 restates DataParallelWithCallback: SynchronizedBatchNorm statistics run over the whole batch with the
    clamp formula, every loss is computed per replica and averaged (spade_model.py:189-203)."""
    cfg = st.cfg
    synced = n_shards > 1
    gcfg = cfg['G']
    num_D, nl = cfg.get('num_D', 2), cfg.get('n_layers_D', 4)
    if cfg.get('no_TTUR', False):
        b1, b2, g_lr, d_lr = cfg['beta1'], cfg['beta2'], cfg['lr'], cfg['lr']
    else:
        b1, b2, g_lr, d_lr = 0.0, 0.9, cfg['lr'] / 2, cfg['lr'] * 2
    n = sem.shape[0]
    per = n // n_shards

    def discriminate(fake_B, training=True):
        """spade_model_modules.py:136-141, per replica: cat over the batch of [sem|fake] and [sem|real]."""
        fakes, reals = [], []
        # one power iteration per netD call; every replica starts from the same u, v (replicated buffers) and device 0's
        # result survives -> restart each shard from the same snapshot so the shared dict sees ONE iteration
        snap = {k: v.clone() for k, v in st.D.items() if k.endswith(('weight_u', 'weight_v'))}
        for r in range(n_shards):
            for k, v in snap.items():
                st.D[k].copy_(v)
            sl = slice(r * per, (r + 1) * per)
            fr = torch.cat([torch.cat([sem[sl], fake_B[sl]], 1), torch.cat([sem[sl], real_B[sl]], 1)], 0)
            pf, pr = divide_pred(multiscale_discriminator(st.D, fr, training, num_D, nl))
            fakes.append(pf)
            reals.append(pr)
        return fakes, reals

    # ---- G step -------------------------------------------------------------------------------------------------
    pS, pD = st.params(st.S), st.params(st.D)
    for v in pS.values():
        v.requires_grad_(True)
        v.grad = None
    for v in pD.values():
        v.requires_grad_(False)
    with torch.no_grad():
        Tfake_B, Tacts = inception_spade_generator(st.T, sem, gcfg, training=False, mapping_layers=MAPPING_LAYERS)
    Sfake_B, Sacts = inception_spade_generator(st.S, sem, gcfg, training=True, synced=synced, mapping_layers=MAPPING_LAYERS)
    distill = []
    mse = cfg.get('distill_G_loss_type', 'ka') == 'mse'
    pA = OrderedDict(('A.' + k, v) for k, v in st.A.items())
    for v in pA.values():
        v.requires_grad_(mse)
        v.grad = None
    for i, name in enumerate(MAPPING_LAYERS):
        if mse:     # spade_distiller_modules.py:23-25: F.mse_loss(netA(Sact), Tact)
            fn = lambda s, t: F.mse_loss(F.conv2d(s, st.A[f'{i}.weight'], st.A[f'{i}.bias']), t)
        else:
            fn = lambda s, t: -ka(s, t)
        distill.append(_shard_mean(fn, n_shards, Sacts[name], Tacts[name]))
    loss_G_distill = sum(distill) * cfg['lambda_distill']
    fakes, reals = discriminate(Sfake_B)
    loss_G_gan = sum(gan_loss('hinge', pf, True, for_discriminator=False) for pf in fakes) / n_shards * cfg['lambda_gan']
    loss_G_feat = 0
    for pf, pr in zip(fakes, reals):
        for i in range(num_D):
            for j in range(len(pf[i]) - 1):
                loss_G_feat = loss_G_feat + F.l1_loss(pf[i][j], pr[i][j].detach()) * cfg['lambda_feat'] / num_D / n_shards
    if st.V is not None and cfg['lambda_vgg'] > 0:
        loss_G_vgg = _shard_mean(lambda a, b: vgg_loss(st.V, a, b), n_shards, Sfake_B, real_B) * cfg['lambda_vgg']
    else:
        loss_G_vgg = torch.zeros(())
    loss_G = loss_G_gan.reshape(()) + loss_G_distill + loss_G_feat + loss_G_vgg
    loss_G.backward()
    adam_step(pS, {k: v.grad for k, v in pS.items()}, st.adam_G, g_lr, b1, b2)
    if mse:
        adam_step(pA, {k: v.grad for k, v in pA.items()}, st.adam_G, g_lr, b1, b2)
        for v in pA.values():
            v.requires_grad_(False)
    st.grads_S = {k: v.grad.clone() for k, v in pS.items() if v.grad is not None}
    for v in pS.values():
        v.requires_grad_(False)
        v.grad = None

    # ---- D step -------------------------------------------------------------------------------------------------
    for v in pD.values():
        v.requires_grad_(True)
        v.grad = None
    with torch.no_grad():
        fake_B, _ = inception_spade_generator(st.S, sem, gcfg, training=True, synced=synced)
    fakes, reals = discriminate(fake_B)
    loss_D_fake = sum(gan_loss('hinge', pf, False, True) for pf in fakes).reshape(()) / n_shards
    loss_D_real = sum(gan_loss('hinge', pr, True, True) for pr in reals).reshape(()) / n_shards
    (loss_D_fake + loss_D_real).backward()
    adam_step(pD, {k: v.grad for k, v in pD.items()}, st.adam_D, d_lr, b1, b2)
    st.grads_D = {k: v.grad.clone() for k, v in pD.items() if v.grad is not None}
    for v in pD.values():
        v.requires_grad_(False)
        v.grad = None

    st.losses = OrderedDict(G_gan=float(loss_G_gan), G_feat=float(loss_G_feat), G_vgg=float(loss_G_vgg),
                            G_distill=float(loss_G_distill), D_real=float(loss_D_real), D_fake=float(loss_D_fake))
    for i, d in enumerate(distill):
        st.losses['G_distill%d' % i] = float(d)
    st.Sfake_B, st.Tfake_B = Sfake_B.detach(), Tfake_B
    return st.losses
