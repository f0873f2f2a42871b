import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
import torch.nn.functional as F
import test_spade_gpu as T
from oracle import detfill, ref_spade_cpu as R
from cat_amd import ops
g, opt, lab, ins, img, sds, cfg = T.fixture()
sem = R.preprocess_input(lab, ins, opt.input_nc)
sd = sds['S']
G = T.make_G(opt, opt.student_ngf, sd, True)
gsem = ops.onehot_edges(lab.cuda(), ins.cuda(), opt.input_nc)
r = detfill.normal((2, 3, int(g['h']), int(g['w'])), 77)
cap = {}
orig_apply = ops.SpadeFn.apply
def patched(x, gb, *a):
    if x.shape[2] == int(g['h']):
        cap['x'], cap['gb'] = x, gb
        x.register_hook(lambda d: cap.__setitem__('dx', d))
        gb.register_hook(lambda d: cap.__setitem__('dgb', d))
    y = orig_apply(x, gb, *a)
    if x.shape[2] == int(g['h']):
        cap['y'] = y
        y.register_hook(lambda d: cap.__setitem__('dy', d))
    return y
ops.SpadeFn.apply = patched
y = G(gsem)
y.backward(T.nhwc(r))
# oracle
ocap = {}
orig_spade = R.inception_spade
def ospade(sd_, p, x, segmap, training, synced):
    if p != 'up_3.spade':
        return orig_spade(sd_, p, x, segmap, training, synced)
    normalized = R.sync_bn(sd_, p + '.param_free_norm', x, training, synced)
    seg = F.interpolate(segmap, size=x.shape[2:], mode='nearest')
    outs = []
    for j in range(3):
        q = f'{p}.res_ops.{j}'
        outs.append(R._conv_same(sd_, q + '.1', R.conv_sync_bn_relu(sd_, q + '.0', seg, training, synced)))
    for j in range(3):
        q = f'{p}.dw_ops.{j}'
        h = R.conv_sync_bn_relu(sd_, q + '.0', seg, training, synced)
        h = R.conv_sync_bn_relu(sd_, q + '.1', h, training, synced)
        outs.append(R._conv_same(sd_, q + '.2', h))
    tmp = outs[0]
    for o in outs[1:]:
        tmp = tmp + o
    tmp.retain_grad(); x.retain_grad()
    c = x.shape[1]
    out = normalized * (1 + tmp[:, :c]) + tmp[:, c:]
    out.retain_grad()
    ocap['x'], ocap['gb'], ocap['pre'] = x, tmp, out
    return out
R.inception_spade = ospade
ref_sd = {k: v.clone().requires_grad_(R.SpadeState._is_param(k)) for k, v in sd.items()}
yr, _ = R.inception_spade_generator(ref_sd, sem, cfg['G'], True, False)
(yr * r).sum().backward()
e = lambda a, b: float((a.detach().cpu() - b.detach()).abs().max() / b.detach().abs().max())
c = ocap['x'].shape[1]
print('x', e(cap['x'], ocap['x']), 'gb', e(cap['gb'], ocap['gb']), 'y', e(cap['y'], F.relu(ocap['pre'])))
# dy w.r.t. activated output: oracle d(pre) = dy*mask
mask = (ocap['pre'] > 0).float()
print('dx', e(cap['dx'], ocap['x'].grad))
dgb_o = ocap['gb'].grad
print('dgamma', e(cap['dgb'][:, :c], dgb_o[:, :c]), 'dbeta', e(cap['dgb'][:, c:], dgb_o[:, c:]))
d = (cap['dgb'][:, c:].cpu() - dgb_o[:, c:]).abs()
print('dbeta mismatch count', int((d > 1e-4 * dgb_o.abs().max()).sum()), 'of', d.numel())
gm = (cap['y'].cpu() > 0).float()
print('mask mismatches', int((gm != mask).sum()), 'y==0 frac', float((gm == 0).float().mean()))
print('dpre oracle vs dbeta oracle', e(ocap['pre'].grad * 1.0, dgb_o[:, c:]) if ocap['pre'].grad is not None else None)
print('sum dbeta gpu/oracle', cap['dgb'][:, c:].sum((0,2,3)).cpu()[:6], dgb_o[:, c:].sum((0,2,3))[:6])
