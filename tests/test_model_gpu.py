"""GPU parity of the whole path against golden vectors produced by the REFERENCE itself (tools/make_golden.py) and
against the oracle on the same seeds: generator / discriminator forward, two full distillation steps (losses, updated
weights, BatchNorm running statistics).  Tolerance 1e-3 relative (north_star)."""
import json

import numpy as np
import pytest
import torch

import helpers as H
from oracle import detfill, ref_cpu

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope='module')
def dev():
    from cat_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _load_gen(opt, sd, dev, train):
    net = H.student_from_shapes(opt, sd) if 'features.0.res_ops.0.1.0.weight' in sd and sd['down_sampling.1.weight'].shape[0] != 64 else None
    if net is None:
        from cat_amd import networks
        net = networks.define_G(3, 3, 64, 'inception_9blocks', opt.norm, 0, 'normal', 0.02, [], opt=opt)
    net.load_state_dict(sd)
    net.to(dev)
    net.train(train)
    return net


@pytest.mark.parametrize('tag,norm,track,d_in', [('in', 'instance', False, 3), ('bn', 'batch', True, 6)])
def test_forward_matches_reference(dev, tag, norm, track, d_in):
    from cat_amd import networks, ops
    g = H.load(f'forward_{tag}.npz')
    opt = H.make_opt(norm=norm, track=track, ndf=64 if tag == 'in' else 128)
    # pruned student, train mode, 1x3x256x256 (BASELINE config 1)
    S = _load_gen(opt, detfill.fill_state_dict(H.sd_from_shapes(g['student_shapes']), H.SEED_S), dev, True)
    acts = {}
    for name, m in S.named_modules():
        if name in ref_cpu.MAPPING_LAYERS:
            m.register_forward_hook(lambda mod, i, o, name=name: acts.__setitem__(name, o))
    x = detfill.images((1, 3, 256, 256), H.SEED_X)
    with torch.no_grad():
        y = S(ops.to_nhwc(x.to(dev)))
    assert H.rel_err(H.sub(y, 3, 8), g['out']) < TOL
    for name in ref_cpu.MAPPING_LAYERS:
        assert H.rel_err(H.sub(acts[name], 8, 8), g['act:' + name]) < TOL, name
    # frozen teacher (eval)
    T = _load_gen(opt, H.teacher_sd(opt), dev, False)
    tacts = {}
    for name, m in T.named_modules():
        if name in ref_cpu.MAPPING_LAYERS:
            m.register_forward_hook(lambda mod, i, o, name=name: tacts.__setitem__(name, o))
    xt = detfill.images((1, 3, 64, 64), H.SEED_X + 1)
    with torch.no_grad():
        yt = T(ops.to_nhwc(xt.to(dev)))
    for name in ref_cpu.MAPPING_LAYERS:
        assert H.rel_err(H.sub(tacts[name], 8, 4), g['tact:' + name]) < TOL, name
    assert H.rel_err(H.sub(yt, 3, 4), g['teacher_out']) < 5 * TOL      # saturated tanh of a 40-layer random net
    # PatchGAN, train mode
    D = networks.define_D(d_in, opt.ndf, 'n_layers', 3, norm, 'normal', 0.02, [], opt=opt)
    D.load_state_dict(H.disc_sd(opt, d_in))
    D.to(dev).train()
    xd = detfill.images((2, d_in, 64, 64), H.SEED_X + 2)
    with torch.no_grad():
        yd = D(ops.to_nhwc(xd.to(dev)))
    assert H.rel_err(yd.cpu().numpy(), g['disc_out']) < TOL


@pytest.mark.parametrize('tag', ['in', 'bn', 'mse'])
def test_two_distill_steps_match_reference(dev, tag):
    from cat_amd import ops
    g = H.load(f'step_{tag}.npz')
    meta = json.loads(str(g['meta']))
    opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                     lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16,
                     distill_G_loss_type=meta.get('distill', 'ka'))
    model = H.build_distiller(opt, g['student_shapes'])
    ops.STATS['conform_copies'] = 0
    for step in range(2):
        A = detfill.images((meta['nbatch'], 3, meta['size'], meta['size']), H.SEED_X + 10 + step)
        B = detfill.images((meta['nbatch'], 3, meta['size'], meta['size']), H.SEED_X + 20 + step)
        model.set_input({'A': A, 'B': B, 'A_paths': [], 'B_paths': []})
        model.optimize_parameters(step)
        losses = model.get_current_losses()
        for k, v in losses.items():
            ref = float(g[f'loss{step}:{k}'])
            assert abs(v - ref) <= TOL * max(1.0, abs(ref)), (step, k, v, ref)
        # step 1 runs on weights that took one Adam step (+-lr*sign(grad), chaotic for near-zero gradients): see the
        # gradient-level test below for the tight comparison
        assert H.rel_err(H.sub(model.Sfake_B, 3, 4), g[f'Sfake{step}']) < (TOL if step == 0 else 1e-2)
        ssd, dsd = model.netG_student.state_dict(), model.netD.state_dict()
        for key in g.files:
            if key.startswith(f'S{step}:') or key.startswith(f'D{step}:') or key.startswith(f'A{step}:'):
                name = key.split(':', 1)[1]
                asd = {f'{i}.{k}': v for i, a in enumerate(model.netAs) for k, v in a.state_dict().items()}
                sd = {'S': ssd, 'D': dsd, 'A': asd}[key[0]]
                got = sd[name].detach().cpu().reshape(-1)[:len(g[key])].numpy()
                diff = np.abs(got - g[key])
                scale = np.abs(g[key]).max()
                # Adam's first updates are +-lr*sign(grad): most elements agree to round-off, none differs by more
                # than a flipped step
                assert np.quantile(diff, 0.75) <= 2e-5 + 2e-4 * scale, (key, diff)
                assert diff.max() <= 2 * meta['lr'] * (step + 1) + 1e-3 * scale, (key, diff)
    assert ops.STATS['conform_copies'] == 0, 'the hot path produced tensors outside the native NHWC layout'


def test_student_gradients_match_oracle(dev):
    """d(loss)/d(student weights) of one step, GPU vs oracle autograd (IN config, N=2, 64x64)."""
    from cat_amd import ops
    g = H.load('step_in.npz')
    meta = json.loads(str(g['meta']))
    opt = H.make_opt(norm='instance', track=False, ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                     lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16)
    model = H.build_distiller(opt, g['student_shapes'])
    A = detfill.images((2, 3, 64, 64), H.SEED_X + 10)
    B = detfill.images((2, 3, 64, 64), H.SEED_X + 20)
    model.set_input({'A': A, 'B': B})
    # run the step but capture the G gradients before Adam consumes them
    model.forward()
    model.set_requires_grad(model.netD, True)
    model.optimizer_D.zero_grad()
    model.backward_D()
    model.optimizer_D.step()
    model.set_requires_grad(model.netD, False)
    model.optimizer_G.zero_grad()
    model.backward_G(0)
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.netG_student.named_parameters()}
    ncfg = H.cfg_for('instance')
    cfg = dict(T=ncfg, S=ncfg, D=ncfg, dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'], lambda_recon=meta['lambda_recon'],
               lambda_distill=meta['lambda_distill'], lambda_gan=1.0, lr=meta['lr'], beta1=meta['beta1'])
    S = detfill.fill_state_dict(H.sd_from_shapes(g['student_shapes']), H.SEED_S)
    st = ref_cpu.DistillState(H.teacher_sd(opt), S, H.disc_sd(opt, 3), cfg)
    ref_cpu.distill_step(st, A, B)
    # network-level gradient criterion (activation-kink flips move single tensors by O(1e-3) of the gradient scale): see check_grads
    import test_spade_gpu as TS
    TS.check_grads(model.netG_student.named_parameters(), st.grads_S)


def test_frozen_teacher_block_fusion_equals_general_path(dev):
    """cat_amd.frozen: the eval / no-grad BatchNorm teacher block evaluated as two concatenated 1x1 GEMMs + slice-wise depthwise convs
    must equal the layer-by-layer path (and, through test_forward_matches_reference, the reference)."""
    from cat_amd import frozen
    opt = H.make_opt(norm='batch', track=True)
    sd = H.teacher_sd(opt)
    from cat_amd import networks
    T = networks.define_G(3, 3, 64, 'inception_9blocks', 'batch', 0, 'normal', 0.02, [0], opt=opt)
    T.load_state_dict(sd)
    T.eval()
    from cat_amd import ops
    x = ops.to_nhwc(detfill.images((2, 3, 64, 64), 991).to(dev))
    with torch.no_grad():
        assert frozen.applicable(T.features[0], T.down_sampling(x))
        y_fast = T(x)
        for blk in T.features:
            blk._cat_frozen_off = True
        assert not frozen.applicable(T.features[0], T.down_sampling(x))
        y_ref = T(x)
    assert H.rel_err(y_fast.cpu().numpy(), y_ref.cpu().numpy()) < 5e-4      # 9 blocks deep, different (algebraically equal) evaluation order
    T.train()
    assert not frozen.applicable(T.features[0], x)


def test_lazy_weight_rehousing_on_a_side_stream_reads_the_live_source(dev):
    """nn._to_channels_last_ on a stream other than the one the weight was allocated on (the frozen teacher's first forward on the data-parallel
    schedule's side stream): the old storage must stay out of the allocator's hands until the re-housing copy has executed.  Round 6: without
    record_stream the construction stream's next allocation recycled it first, and the teacher's late layers were copied from garbage in about
    half of the two-rank runs.  Here the side stream is kept busy so that the copy is still pending while the main stream recycles."""
    from cat_amd import nn as cnn
    conv = cnn.Conv2d(130, 512, 3).to(dev)            # 2.4 MB, Cin 130 -> padded to 132: re-housed at first use
    ref = conv.weight.detach().clone()
    big = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        acc = big
        for _ in range(30):                            # tens of milliseconds of queued work in front of the copy
            acc = (acc @ big) * 1e-3
        cnn._to_channels_last_(conv)
    junk = [torch.full_like(ref, float('nan')) for _ in range(6)]      # main stream: same-size allocations, written at once
    torch.cuda.synchronize()
    assert torch.equal(conv.weight.detach(), ref)
    del junk, acc
