"""Pins oracle/ref_spade_cpu.py (the CPU restatement of the GauGAN / SPADE distillation step) against
tests/golden/spade_step.npz, produced by the REAL reference (tools/make_golden_spade.py).  CPU only."""
import json

import numpy as np
import torch

import helpers as H
from oracle import detfill
from oracle import ref_spade_cpu as R

SEED_T, SEED_S, SEED_D, SEED_V = 111, 121, 141, 161       # tools/make_golden_spade.py


def _checks(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def _close(a, b, tol=1e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-6)
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max(), scale)


def spade_fixture():
    g = H.load('spade_step.npz')
    opt = json.loads(str(g['opt']))
    T = detfill.fill_state_dict(H.sd_from_shapes(g['T_shapes']), SEED_T)
    S = detfill.fill_state_dict(H.sd_from_shapes(g['S_shapes']), SEED_S)
    D = detfill.fill_state_dict(H.sd_from_shapes(g['D_shapes']), SEED_D)
    Vref = detfill.fill_state_dict(spade_vgg_feature_shapes(g), SEED_V)
    lab = torch.from_numpy(g['label'].astype(np.int64))
    ins = torch.from_numpy(g['instance'])
    img = detfill.images((int(g['n']), 3, int(g['h']), int(g['w'])), int(g['image_seed']))
    cfg = dict(G=dict(crop_size=opt['crop_size'], aspect_ratio=opt['aspect_ratio'], num_upsampling_layers=opt['num_upsampling_layers']),
               num_D=opt['num_D'], n_layers_D=opt['n_layers_D'], lambda_gan=opt['lambda_gan'], lambda_feat=opt['lambda_feat'],
               lambda_vgg=opt['lambda_vgg'], lambda_distill=opt['lambda_distill'], lr=opt['lr'], beta1=opt['beta1'], beta2=opt['beta2'],
               no_TTUR=opt['no_TTUR'])
    return g, opt, T, S, D, Vref, lab, ins, img, cfg


def spade_vgg_feature_shapes(g):
    """The golden generator filled `features` (torchvision index keys) in ITS key order; rebuild that dict of shapes."""
    vmap = json.loads(str(g['V_keymap']))
    shapes = {}
    for k, shape in json.loads(str(g['V_shapes'])):
        shapes[vmap[k]] = torch.zeros(shape)
    order = sorted(shapes, key=lambda k: (int(k.split('.')[0]), 0 if k.endswith('weight') else 1))
    full = {}
    # layers past relu5_1 (features[30:]) exist in torchvision's net and were filled too; they follow in key order and do not
    # shift the draws of the earlier ones, so filling the first 30 indices reproduces the values.
    for k in order:
        full[k] = shapes[k]
    return full


def test_preprocess_input():
    g, opt, *_rest, lab, ins, img, cfg = spade_fixture()
    sem = R.preprocess_input(lab, ins, opt['input_nc'])
    assert sem.shape[1] == opt['semantic_nc']
    np.testing.assert_array_equal(sem[:, -1].numpy().astype(np.uint8), g['sem_edge'])
    np.testing.assert_allclose(_checks(sem), g['sem_checks'], rtol=0, atol=0)


def test_generator_forward_and_discriminator():
    g, opt, T, S, D, V, lab, ins, img, cfg = spade_fixture()
    sem = R.preprocess_input(lab, ins, opt['input_nc'])
    with torch.no_grad():
        Tfake, Tacts = R.inception_spade_generator(T, sem, cfg['G'], training=False, mapping_layers=R.MAPPING_LAYERS)
        S1 = {k: v.clone() for k, v in S.items()}
        Sfake, Sacts = R.inception_spade_generator(S1, sem, cfg['G'], training=True, mapping_layers=R.MAPPING_LAYERS)
    _close(H.sub(Tfake, 6, 4), g['Tfake_sub'])
    _close(H.sub(Sfake, 6, 4), g['Sfake_sub'])
    _close(_checks(Tfake), g['Tfake_checks'])
    _close(_checks(Sfake), g['Sfake_checks'])
    for name in R.MAPPING_LAYERS:
        _close(_checks(Tacts[name]), g[f'Tact_{name}'])
        _close(_checks(Sacts[name]), g[f'Sact_{name}'])
    _close(S1['head_0.spade.param_free_norm.running_mean'].numpy(), g['S_rm_head'])
    _close(S1['up_3.res_ops.1.0.norm.running_var'].numpy(), g['S_rv_up3'])
    with torch.no_grad():
        fr = torch.cat([torch.cat([sem, Sfake], 1), torch.cat([sem, img], 1)], 0)
        D1 = {k: v.clone() for k, v in D.items()}
        dout = R.multiscale_discriminator(D1, fr, True, cfg['num_D'], cfg['n_layers_D'])
    for i, scale in enumerate(dout):
        for j, t in enumerate(scale):
            _close(_checks(t), g[f'D_{i}_{j}'])
    _close(D1['discriminator_0.model2.0.0.weight_u'].numpy(), g['D_u_after'])


def test_spade_step():
    g, opt, T, S, D, V, lab, ins, img, cfg = spade_fixture()
    sem = R.preprocess_input(lab, ins, opt['input_nc'])
    st = R.SpadeState(T, S, D, V, cfg)
    losses = R.spade_step(st, sem, img)
    ref = json.loads(str(g['losses']))
    for k in ('G_gan', 'G_feat', 'G_vgg', 'G_distill', 'D_real', 'D_fake', 'G_distill0', 'G_distill1', 'G_distill2'):
        assert abs(losses[k] - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-2), (k, losses[k], ref[k])
    # Whole-step gradients pass through the piecewise-linear discriminator (LeakyReLU, L1, hinge): a 1e-5 difference in the
    # generated image flips a few units, so single gradient entries move by O(1e-3) of the gradient scale even between two
    # fp32 evaluation orders of the SAME code.  Hence: entries within 5e-3 of the global gradient maximum, tensor norms 1e-2.
    # (Sharp 1e-3 parity of every backward op is pinned with fixed inputs in test_generator/discriminator tests.)
    # The D step additionally sees a student that moved by +-lr PER WEIGHT (Adam's first step is sign(g); the sign of a
    # noise-level gradient is arbitrary), so its gradients agree a little less tightly.
    check_step_grads(g, 'S', st.grads_S, st.S, SEED_S, 5e-3)
    check_step_grads(g, 'D', st.grads_D, st.D, SEED_D, 2e-2)
    _close(st.D['discriminator_1.model2.0.0.weight_u'].numpy(), g['D_u_step'])
    _close(st.S['G_middle_0.spade.param_free_norm.running_var'].numpy(), g['S_rv_step'])


def check_step_grads(g, tag, grads, params, seed, tol):
    gmax = float(g[tag + '_gmax'])
    before = detfill.fill_state_dict(H.sd_from_shapes(g[tag + '_shapes']), seed)
    for k in json.loads(str(g['probe_' + tag])):
        got, ref = grads[k].detach().cpu().numpy().reshape(-1)[:256], g[f'{tag}_grad/' + k]
        assert np.abs(got - ref).max() <= tol * gmax, (k, np.abs(got - ref).max(), gmax)
        gn = float(grads[k].double().norm())
        assert abs(gn - float(g[f'{tag}_gnorm/' + k])) <= 4 * tol * max(float(g[f'{tag}_gnorm/' + k]), 1e-3 * gmax), k
        # Adam's first step moves every weight by ~lr * sign(g): compare the UPDATE on entries whose gradient is solid
        b = before[k].numpy().reshape(-1)[:256]
        d_ref, d_got = g[f'{tag}_after/' + k] - b, params[k].detach().cpu().numpy().reshape(-1)[:256] - b
        solid = np.abs(ref) > 10 * tol * gmax
        if solid.any():
            np.testing.assert_allclose(d_got[solid], d_ref[solid], rtol=5e-2, atol=1e-7)


def test_spadeinstance_generator_forward_and_gradients():
    """norm_G='spadeinstance3x3' (reference inception_modules.py:407-423; no launch script uses it): the block's hidden / shortcut norms and
    the SPADE layers' param-free norm are InstanceNorm2d -- recognised by the oracle from the state dict (no running statistics) -- while the
    gamma|beta nets keep SynchronizedBatchNorm2d.  Train-mode forward, five parameter gradients and the SyncBN statistics vs the reference."""
    g = H.load('spade_instance_fwd.npz')
    sd = detfill.fill_state_dict(H.sd_from_shapes(g['shapes']), 701, gamma_abs_normal=True)
    assert 'head_0.spade.param_free_norm.running_mean' not in sd and 'head_0.spade.res_ops.0.0.norm.running_mean' in sd
    assert sorted(k for k in sd if 'running' in k) == json.loads(str(g['state_after']))
    names = [k[5:] for k in g.files if k.startswith('grad:')]
    for k in names:
        sd[k].requires_grad_(True)
    lab, ins = torch.from_numpy(g['label'].astype(np.int64)), torch.from_numpy(g['instance'])
    sem = R.preprocess_input(lab, ins, 5)
    y, _ = R.inception_spade_generator(sd, sem, dict(crop_size=256, aspect_ratio=2.0, num_upsampling_layers='more'), training=True)
    _close(_checks(y), g['y_checks'], 1e-4)
    _close(y.detach()[:, :, ::2, ::2].numpy(), g['y_sub'], 1e-4)
    (y * detfill.normal(tuple(y.shape), 712)).sum().backward()
    for k in names:
        _close(sd[k].grad.numpy(), g['grad:' + k], 2e-3)
    for k in (f[4:] for f in g.files if f.startswith('buf:')):
        _close(sd[k].detach().numpy(), g['buf:' + k], 1e-5)
