"""hipGraph replay of the whole distillation step (cat_amd.graph.GraphedStep) must reproduce the eager launch sequence:
same kernels, same order, deterministic reductions -> identical losses and weights."""
import json

import numpy as np
import pytest
import torch

import helpers as H
from oracle import detfill

pytestmark = pytest.mark.gpu


def _max_param_diff(a, b):
    worst = 0.0
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb
        if va.dtype.is_floating_point:
            scale = max(float(vb.abs().max()), 1e-6)
            worst = max(worst, float((va - vb).abs().max()) / scale)
    return worst


@pytest.fixture
def branch_streams(request):
    """Run the test with the block branches / weight-gradient jobs on side HIP streams (True) or on one stream (False: the inception
    distillers' default since round 6); the process-wide flag is restored afterwards."""
    from cat_amd import ops
    was = ops.branch_streams_enabled()
    yield request.param
    ops.set_branch_streams(was)


@pytest.mark.parametrize('branch_streams', [False, True], indirect=True)
def test_graph_replay_equals_eager_inception(branch_streams):
    from cat_amd import ops
    from cat_amd.graph import GraphedStep
    g = H.load('step_bn.npz')
    meta = json.loads(str(g['meta']))

    def build():
        opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                         lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16)
        return H.build_distiller(opt, g['student_shapes'])
    n, s = meta['nbatch'], meta['size']
    batches = [{'A': detfill.images((n, 3, s, s), 500 + i).cuda(), 'B': detfill.images((n, 3, s, s), 600 + i).cuda(), 'A_paths': [], 'B_paths': []}
               for i in range(3)]
    eager, graphed = build(), build()
    assert not ops.branch_streams_enabled() or 'CAT_BRANCH_STREAMS' in __import__('os').environ      # the constructor's default: one stream
    ops.set_branch_streams(branch_streams)
    for i in range(3):                       # GraphedStep's warm-up: 3 eager steps on the example batch
        eager.set_input(batches[0])
        eager.optimize_parameters(i)
    step = GraphedStep(graphed, batches[0], warmup=3)
    for i in (1, 2, 1):
        eager.set_input(batches[i])
        eager.optimize_parameters(3 + i)
        step(batches[i])
        le, lg = eager.get_current_losses(), graphed.get_current_losses()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-5 * max(1.0, abs(le[k])), (k, le[k], lg[k])
    assert _max_param_diff(graphed.netG_student, eager.netG_student) < 1e-5
    assert _max_param_diff(graphed.netD, eager.netD) < 1e-5
    assert graphed.optimizer_G._flat[0]['step'] == eager.optimizer_G._flat[0]['step'] == 6
    assert float(graphed.optimizer_G._flat[0]['hyper'][5]) == 6.0
    # a learning-rate schedule (LambdaLR's linear decay, or update_learning_rate) between replays reaches the device-resident scalars
    for m in (eager, graphed):
        for o in m.optimizers:
            for grp in o.param_groups:
                grp['lr'] = grp['lr'] * 0.37
    eager.set_input(batches[2])
    eager.optimize_parameters(7)
    step(batches[2])
    assert abs(float(graphed.optimizer_G._flat[0]['hyper'][0]) - eager.optimizer_G.param_groups[0]['lr']) < 1e-9
    assert _max_param_diff(graphed.netG_student, eager.netG_student) < 1e-5
    assert _max_param_diff(graphed.netD, eager.netD) < 1e-5
    # a batch of another shape (the last batch of an epoch) falls back to eager launches instead of failing in copy_
    small = {k: (v[:1] if torch.is_tensor(v) else v) for k, v in batches[1].items()}
    step(small)
    assert graphed.Sfake_B.shape[0] == 1
    # ... and the replay that follows reports ITS losses / images again (the fallback rebinds loss_* / Sfake_B to eager tensors)
    small_losses = graphed.get_current_losses()
    eager.set_input(small)
    eager.optimize_parameters(8)
    eager.set_input(batches[1])
    eager.optimize_parameters(9)
    step(batches[1])
    assert graphed.Sfake_B.shape[0] == n
    le, lg = eager.get_current_losses(), graphed.get_current_losses()
    assert any(abs(lg[k] - small_losses[k]) > 1e-7 for k in lg), 'losses frozen at the eager fallback step'
    for k in le:
        assert abs(le[k] - lg[k]) <= 1e-5 * max(1.0, abs(le[k])), (k, le[k], lg[k])
    assert float((graphed.Sfake_B - eager.Sfake_B).abs().max()) < 1e-5


def test_graph_replay_equals_eager_spade():
    import test_spade_gpu as TS
    from cat_amd.graph import GraphedStep
    g, opt, lab, ins, img, sds, cfg = TS.fixture()
    opt.isTrain, opt.distiller, opt.log_dir = True, 'spade', '/tmp/cat_amd_logs'
    rng = np.random.default_rng(9)
    h, w, n = int(g['h']), int(g['w']), int(g['n'])
    batches = []
    for i in range(3):
        lab_i = np.repeat(np.repeat(rng.integers(0, opt.input_nc, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32)
        ins_i = np.repeat(np.repeat(rng.integers(0, 99, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32)
        batches.append({'label': torch.from_numpy(lab_i).cuda(), 'instance': torch.from_numpy(ins_i).cuda(),
                        'image': detfill.images((n, 3, h, w), 700 + i).cuda(), 'path': []})
    eager, graphed = TS.build_spade_distiller(opt, sds), TS.build_spade_distiller(opt, sds)
    for i in range(3):
        eager.set_input(batches[0])
        eager.optimize_parameters(i)
    step = GraphedStep(graphed, batches[0], warmup=3)
    for i in (1, 2):
        eager.set_input(batches[i])
        eager.optimize_parameters(3 + i)
        step(batches[i])
        le, lg = eager.get_current_losses(), graphed.get_current_losses()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-5 * max(1.0, abs(le[k])), (k, le[k], lg[k])
    me, mg = eager.modules_on_one_gpu, graphed.modules_on_one_gpu
    assert _max_param_diff(mg.netG_student, me.netG_student) < 1e-5
    assert _max_param_diff(mg.netD, me.netD) < 1e-5


@pytest.mark.parametrize('fused', [False, True])
def test_step_contains_only_library_kernels(fused):
    """Every device activity of one optimize_parameters is a libcat_hip kernel: no ATen elementwise / reduction kernel, no
    memcpy / memset node (what torch.profiler sees through roctracer, which is also what a captured hipGraph would hold)."""
    from torch.profiler import ProfilerActivity, profile
    from cat_amd import ops
    g = H.load('step_bn.npz')
    meta = json.loads(str(g['meta']))
    opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                     lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16)
    old = ops.set_tconv_min_tiles(1 if fused else 1 << 30)      # with / without the LDS-tile + fused-block kernels on this 64 x 64 batch
    try:
        model = H.build_distiller(opt, g['student_shapes'])
        n, s = meta['nbatch'], meta['size']
        b = {'A': detfill.images((n, 3, s, s), 1).cuda(), 'B': detfill.images((n, 3, s, s), 2).cuda(), 'A_paths': [], 'B_paths': []}
        for i in range(2):
            model.set_input(b)
            model.optimize_parameters(i)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            model.set_input(b)
            model.optimize_parameters(2)
            torch.cuda.synchronize()
    finally:
        ops.set_tconv_min_tiles(old)
    names = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            names[e.name] = names.get(e.name, 0) + 1
    assert len(names) > 30, names
    foreign = {k: v for k, v in names.items() if not ('(anonymous namespace)::' in k or 'cat_' in k)}
    assert not foreign, foreign
    if fused:
        assert any('tconv_kernel' in k for k in names) and any('tnorm_finalize' in k for k in names), sorted(names)


def test_segment_graphs_without_reducer_equal_eager():
    """GraphedDPStep on ONE GPU without a reducer: set_input | teacher (side stream) | student fwd + D bwd | Adam D + G bwd + Adam G as four
    hipGraphs must reproduce the eager step bit for bit (same kernels, the teacher merely overlaps the student / discriminator work)."""
    from cat_amd.graph import GraphedDPStep
    g = H.load('step_bn.npz')
    meta = json.loads(str(g['meta']))

    def build():
        opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                         lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16)
        return H.build_distiller(opt, g['student_shapes'])
    n, s = meta['nbatch'], meta['size']
    batches = [{'A': detfill.images((n, 3, s, s), 510 + i).cuda(), 'B': detfill.images((n, 3, s, s), 610 + i).cuda(), 'A_paths': [], 'B_paths': []}
               for i in range(3)]
    eager, graphed = build(), build()
    for i in range(2):
        eager.set_input(batches[0])
        eager.optimize_parameters(i)
    step = GraphedDPStep(graphed, batches[0], warmup=2)
    assert not step.dp
    for i in (1, 2, 1):
        eager.set_input(batches[i])
        eager.optimize_parameters(2 + i)
        step(batches[i])
        le, lg = eager.get_current_losses(), graphed.get_current_losses()
        for k in le:
            assert le[k] == lg[k], (k, le[k], lg[k])
    torch.cuda.synchronize()
    for a, b in ((graphed.netG_student, eager.netG_student), (graphed.netD, eager.netD)):
        for (ka, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert torch.equal(va, vb), ka
    assert graphed.optimizer_G._flat[0]['step'] == eager.optimizer_G._flat[0]['step'] == 5
