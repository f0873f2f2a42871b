"""CPU: pin the oracle (oracle/ref_cpu.py) against golden vectors produced by the REAL reference code
(tools/make_golden.py ran snap-research/CAT on CPU).  Tolerances: the reference and the oracle execute the same ATen
kernels, so agreement is at fp32 round-off (1e-5 relative); integer / mask results are exact."""
import json

import numpy as np
import pytest
import torch

import helpers as H
from oracle import detfill, ref_cpu

TOL = 2e-5


def test_ka_and_gan_losses():
    g = H.load('small_ops.npz')
    for n in (2, 3, 16):
        X = detfill.normal((n, 7, 6, 5), 100 + n).requires_grad_(True)
        Y = detfill.normal((n, 11, 6, 5), 200 + n)
        v = ref_cpu.ka(X, Y)
        v.backward()
        assert abs(v.item() - float(g[f'ka{n}'])) < 1e-6
        assert H.rel_err(X.grad.numpy(), g[f'ka{n}_grad']) < TOL
    pred = detfill.normal((2, 1, 6, 6), 300, 1.5)
    for mode in ('hinge', 'lsgan', 'vanilla', 'wgangp'):
        for real in (True, False):
            p = pred.clone().requires_grad_(True)
            l = ref_cpu.gan_loss(mode, p, real, True)
            l.backward()
            assert abs(l.item() - float(g[f'gan_{mode}_D_{int(real)}'])) < 1e-6
            assert np.array_equal(p.grad.numpy(), g[f'gan_{mode}_D_{int(real)}_grad'])
        p = pred.clone().requires_grad_(True)
        l = ref_cpu.gan_loss(mode, p, True, False)
        l.backward()
        assert abs(l.item() - float(g[f'gan_{mode}_G'])) < 1e-6
    preds = [[detfill.normal((2, 4, 5, 5), 310), detfill.normal((2, 1, 5, 5), 311)],
             [detfill.normal((2, 4, 3, 3), 312), detfill.normal((2, 1, 3, 3), 313)]]
    for key, args in (('D_1', (True, True)), ('D_0', (False, True)), ('G', (True, False))):
        out = ref_cpu.gan_loss('hinge', preds, *args)
        assert H.rel_err(out.numpy(), g['gan_hinge_list_' + key]) < 1e-6


@pytest.mark.parametrize('tag,norm,track,d_in,size', [('in', 'instance', False, 3, 64), ('bn', 'batch', True, 6, 64)])
def test_forward(tag, norm, track, d_in, size):
    g = H.load(f'forward_{tag}.npz')
    opt = H.make_opt(norm=norm, track=track, ndf=64 if tag == 'in' else 128)
    cfg = H.cfg_for(norm)
    S = detfill.fill_state_dict(H.sd_from_shapes(g['student_shapes']), H.SEED_S)
    x = detfill.images((1, 3, 256, 256), H.SEED_X)
    with torch.no_grad():
        y, acts = ref_cpu.inception_generator(S, x, cfg, training=True)
    assert H.rel_err(H.sub(y, 3, 8), g['out']) < TOL
    for name, a in acts.items():
        assert H.rel_err(H.sub(a, 8, 8), g['act:' + name]) < TOL
    T = H.teacher_sd(opt)
    xt = detfill.images((1, 3, size, size), H.SEED_X + 1)
    with torch.no_grad():
        yt, tacts = ref_cpu.inception_generator(T, xt, cfg, training=False)
    # ~40 conv layers deep with |N(0,1)| norm scales: module-vs-functional ATen dispatch differences are amplified
    assert H.rel_err(H.sub(yt, 3, 4), g['teacher_out']) < 3e-4
    for name, a in tacts.items():
        assert H.rel_err(H.sub(a, 8, 4), g['tact:' + name]) < 3e-4
    D = H.disc_sd(opt, d_in)
    xd = detfill.images((2, d_in, size, size), H.SEED_X + 2)
    with torch.no_grad():
        yd = ref_cpu.nlayer_discriminator(D, xd, cfg, training=True)
    assert H.rel_err(yd.numpy(), g['disc_out']) < TOL


@pytest.mark.parametrize('tag', ['in', 'bn', 'mse'])
def test_two_distill_steps(tag):
    g = H.load(f'step_{tag}.npz')
    meta = json.loads(str(g['meta']))
    opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'])
    d_in = 6 if meta['dataset_mode'] == 'aligned' else 3
    ncfg = H.cfg_for(meta['norm'])
    cfg = dict(T=ncfg, S=ncfg, D=ncfg, dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'], lambda_recon=meta['lambda_recon'],
               lambda_distill=meta['lambda_distill'], lambda_gan=meta['lambda_gan'], lr=meta['lr'], beta1=meta['beta1'],
               distill_G_loss_type=meta.get('distill', 'ka'))
    S = detfill.fill_state_dict(H.sd_from_shapes(g['student_shapes']), H.SEED_S)
    st = ref_cpu.DistillState(H.teacher_sd(opt), S, H.disc_sd(opt, d_in), cfg, H.netA_sds(S))
    for step in range(2):
        A = detfill.images((meta['nbatch'], 3, meta['size'], meta['size']), H.SEED_X + 10 + step)
        B = detfill.images((meta['nbatch'], 3, meta['size'], meta['size']), H.SEED_X + 20 + step)
        losses = ref_cpu.distill_step(st, A, B)
        for k, v in losses.items():
            pref = 'Specific_loss/' if k[-1].isdigit() else ('D_loss/' if k.startswith('D_') else 'G_loss/')
            ref = float(g[f'loss{step}:{pref}{k}'])
            assert abs(v - ref) <= 2e-4 * max(1.0, abs(ref)), (step, k, v, ref)
        # step 0 runs on the initial weights (tight); step 1 runs on Adam-updated weights whose first updates are
        # +-lr * sign(grad): round-off sized gradient differences can flip a sign, so the bound is looser
        assert H.rel_err(H.sub(st.Sfake_B, 3, 4), g[f'Sfake{step}']) < (2e-5 if step == 0 else 1e-3)
        for key in g.files:
            if key.startswith(f'S{step}:') or key.startswith(f'D{step}:') or key.startswith(f'A{step}:'):
                name = key.split(':', 1)[1]
                sd = {'S': st.S, 'D': st.D, 'A': st.A}[key[0]]
                got = sd[name].detach().reshape(-1)[:len(g[key])].numpy()
                # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the UPDATE too
                diff = np.abs(got - g[key])
                scale = np.abs(g[key]).max()
                assert np.quantile(diff, 0.9) <= (2e-6 + 2e-5 * scale if step == 0 else 2e-5 + 2e-4 * scale), key
                assert diff.max() <= 2 * meta['lr'] * (step + 1) + 1e-6, key     # never further than a flipped Adam step


@pytest.mark.parametrize('tag', ['in', 'bn'])
def test_shrink_search(tag):
    g = H.load(f'shrink_{tag}.npz')
    gam = {'down': [torch.from_numpy(g[f'g_down{i}']) for i in range(3)], 'up': [torch.from_numpy(g[f'g_up{i}']) for i in range(2)],
           'blocks': [([torch.from_numpy(g[f'g_b{b}_res{j}']) for j in range(3)], [torch.from_numpy(g[f'g_b{b}_dw{j}']) for j in range(3)])
                      for b in range(9)]}
    track = bool(g['track'])
    thr, searched, cfg = ref_cpu.shrink_search(gam, float(g['target']), int(g['prune_cin_lb']), norm_counts=not track)
    ref_cfg = json.loads(str(g['cfg']))
    assert np.float32(thr.item()) == g['thr']           # bit-exact threshold
    assert searched == int(g['s_macs'][0])
    assert cfg['down'] == ref_cfg['down'] and cfg['up'] == ref_cfg['up']
    assert [[list(r), list(d)] for r, d in cfg['blocks']] == ref_cfg['blocks']
    full = ref_cpu.generator_macs(dict(down=[64, 128, 256], up=[128, 64], blocks=[([42] * 3, [42] * 3)] * 9, kernel_sizes=[1, 3, 5]),
                                  norm_counts=not track)
    assert [full['total'], full['down_sampling'], full['features'], full['up_sampling']] == list(g['t_macs'])
