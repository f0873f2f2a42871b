"""Teacher-training steps: Pix2PixModel / CycleGANModel.optimize_parameters.
CPU: the oracle restatements (oracle/ref_cpu.py pix2pix_step / cyclegan_step) against tests/golden/train_steps.npz, recorded from
the reference's own classes.  GPU: cat_amd.models on the HIP kernels against the same fixture."""
import json
import random

import numpy as np
import pytest
import torch

import helpers as H
from oracle import detfill, ref_cpu


def _cfgs(meta):
    ncfg = H.cfg_for(meta['norm'])
    return ncfg


def _check_probe(got, ref, lr, steps_done, tight):
    diff = np.abs(got - ref)
    scale = np.abs(ref).max()
    # Adam's first updates are +-lr * sign(g): most entries agree to round-off, none is further away than flipped steps
    assert np.quantile(diff, 0.75) <= (2e-5 + 2e-4 * scale if not tight else 2e-6 + 2e-5 * scale), diff
    assert diff.max() <= 2 * lr * steps_done + 1e-3 * scale, diff


def test_pix2pix_oracle_matches_reference():
    g = H.load('train_steps.npz')
    meta = json.loads(str(g['p2p_meta']))
    ncfg = _cfgs(meta)
    nets = {'G': detfill.fill_state_dict(H.sd_from_shapes(g['p2p_G_shapes']), 301), 'D': detfill.fill_state_dict(H.sd_from_shapes(g['p2p_D_shapes']), 302)}
    st = ref_cpu.TrainState(nets, dict(G=ncfg, D=ncfg, gan_mode=meta['gan_mode'], lambda_gan=1.0, lambda_recon=meta['lambda_recon'], lr=meta['lr'],
                                       beta1=meta['beta1']))
    for step in range(2):
        A, B = detfill.images((2, 3, 64, 64), 310 + step), detfill.images((2, 3, 64, 64), 320 + step)
        losses = ref_cpu.pix2pix_step(st, A, B)
        for k, v in losses.items():
            ref = float(g[f'p2p_loss{step}:{k}'])
            assert abs(v - ref) <= 3e-4 * max(1.0, abs(ref)), (step, k, v, ref)
        assert H.rel_err(H.sub(st.fake_B, 3, 4), g[f'p2p_fake{step}']) < (2e-5 if step == 0 else 2e-3)
        for key in g.files:
            if key.startswith(f'p2p_G{step}:') or key.startswith(f'p2p_D{step}:'):
                sd = st.nets[key[4]]
                name = key.split(':', 1)[1]
                _check_probe(sd[name].detach().reshape(-1)[:len(g[key])].numpy(), g[key], meta['lr'], step + 1, step == 0)


def test_cyclegan_oracle_matches_reference():
    g = H.load('train_steps.npz')
    meta = json.loads(str(g['cyc_meta']))
    ncfg = _cfgs(meta)
    gsh, dsh = H.sd_from_shapes(g['cyc_G_shapes']), H.sd_from_shapes(g['cyc_D_shapes'])
    nets = {'G_A': detfill.fill_state_dict(gsh, 401), 'G_B': detfill.fill_state_dict(gsh, 402), 'D_A': detfill.fill_state_dict(dsh, 411),
            'D_B': detfill.fill_state_dict(dsh, 412)}
    st = ref_cpu.TrainState(nets, dict(G=ncfg, D=ncfg, gan_mode=meta['gan_mode'], lambda_A=meta['lambda_A'], lambda_B=meta['lambda_B'],
                                       lambda_identity=meta['lambda_identity'], lr=meta['lr'], beta1=meta['beta1']))
    rng = random.Random(meta['seed'])
    pools = (ref_cpu.ImagePoolRef(meta['pool_size'], rng), ref_cpu.ImagePoolRef(meta['pool_size'], rng))
    for step in range(3):
        A, B = detfill.images((1, 3, 64, 64), 420 + step), detfill.images((1, 3, 64, 64), 430 + step)
        losses = ref_cpu.cyclegan_step(st, A, B, pools)
        for k, v in losses.items():
            ref = float(g[f'cyc_loss{step}:{k}'])
            assert abs(v - ref) <= (3e-4 if step == 0 else 5e-3) * max(1.0, abs(ref)), (step, k, v, ref)
        assert H.rel_err(H.sub(st.fake_B, 3, 4), g[f'cyc_fakeB{step}']) < (2e-5 if step == 0 else 5e-3)


def _opt_for(meta, **kw):
    return H.make_opt(norm=meta['norm'], track=meta.get('track', False), ndf=meta['ndf'], gan_mode=meta['gan_mode'], ngf=meta['ngf'],
                      netG='inception_9blocks', dropout_rate=0, direction='AtoB', lr=meta['lr'], beta1=meta['beta1'], **kw)


@pytest.mark.gpu
def test_pix2pix_step_gpu():
    from cat_amd import ops
    from cat_amd.models import create_model
    g = H.load('train_steps.npz')
    meta = json.loads(str(g['p2p_meta']))
    opt = _opt_for(meta, model='pix2pix', lambda_recon=meta['lambda_recon'], lambda_gan=1.0, recon_loss_type='l1', lambda_comp_cost=0.5, comp_cost='l1', l1_renorm=False)     # the comp-cost flag adds a zero term (pix2pix_model.py:186-195)
    m = create_model(opt, verbose=False)
    m.netG.load_state_dict(detfill.fill_state_dict(H.sd_from_shapes(g['p2p_G_shapes']), 301))
    m.netD.load_state_dict(detfill.fill_state_dict(H.sd_from_shapes(g['p2p_D_shapes']), 302))
    m.setup(opt, verbose=False)
    ops.STATS['conform_copies'] = 0
    for step in range(2):
        A, B = detfill.images((2, 3, 64, 64), 310 + step), detfill.images((2, 3, 64, 64), 320 + step)
        m.set_input({'A': A, 'B': B, 'A_paths': [], 'B_paths': []})
        m.optimize_parameters(step)
        losses = m.get_current_losses()
        for k in ('G_gan', 'G_recon', 'D_real', 'D_fake'):
            got = losses[('D_loss/' if k.startswith('D') else 'G_loss/') + k]
            ref = float(g[f'p2p_loss{step}:{k}'])
            assert abs(got - ref) <= 1e-3 * max(1.0, abs(ref)), (step, k, got, ref)
        assert losses['G_loss/G_comp_cost'] == 0.0
        assert H.rel_err(H.sub(m.fake_B, 3, 4), g[f'p2p_fake{step}']) < (1e-3 if step == 0 else 1e-2)
        for key in g.files:
            if key.startswith(f'p2p_G{step}:') or key.startswith(f'p2p_D{step}:'):
                net = m.netG if key[4] == 'G' else m.netD
                name = key.split(':', 1)[1]
                _check_probe(net.state_dict()[name].detach().cpu().reshape(-1)[:len(g[key])].numpy(), g[key], meta['lr'], step + 1, False)
    assert ops.STATS['conform_copies'] == 0


@pytest.mark.gpu
def test_cyclegan_step_gpu():
    from cat_amd import ops
    from cat_amd.models import create_model
    g = H.load('train_steps.npz')
    meta = json.loads(str(g['cyc_meta']))
    opt = _opt_for(meta, model='cycle_gan', dataset_mode='unaligned', lambda_A=meta['lambda_A'], lambda_B=meta['lambda_B'],
                   lambda_identity=meta['lambda_identity'], pool_size=meta['pool_size'])
    m = create_model(opt, verbose=False)
    gsh, dsh = H.sd_from_shapes(g['cyc_G_shapes']), H.sd_from_shapes(g['cyc_D_shapes'])
    m.netG_A.load_state_dict(detfill.fill_state_dict(gsh, 401))
    m.netG_B.load_state_dict(detfill.fill_state_dict(gsh, 402))
    m.netD_A.load_state_dict(detfill.fill_state_dict(dsh, 411))
    m.netD_B.load_state_dict(detfill.fill_state_dict(dsh, 412))
    m.setup(opt, verbose=False)
    random.seed(meta['seed'])
    ops.STATS['conform_copies'] = 0
    for step in range(3):
        A, B = detfill.images((1, 3, 64, 64), 420 + step), detfill.images((1, 3, 64, 64), 430 + step)
        m.set_input({'A': A, 'B': B})
        m.optimize_parameters(step)
        losses = m.get_current_losses()
        for k in ('D_A', 'G_A', 'G_cycle_A', 'G_idt_A', 'D_B', 'G_B', 'G_cycle_B', 'G_idt_B'):
            got = losses[('D_loss/' if k.startswith('D') else 'G_loss/') + k]
            ref = float(g[f'cyc_loss{step}:{k}'])
            assert abs(got - ref) <= (1e-3 if step == 0 else 1e-2) * max(1.0, abs(ref)), (step, k, got, ref)
        assert H.rel_err(H.sub(m.fake_B, 3, 4), g[f'cyc_fakeB{step}']) < (1e-3 if step == 0 else 2e-2)
    assert ops.STATS['conform_copies'] == 0
