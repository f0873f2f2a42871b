"""cat_amd.export.to_reference_module: the stock-torch, NCHW, host-side twin of a cat_amd generator that the
reference's exporter (onnx_exporter.py:142-163: torch.onnx.export(model.netG_student.cpu(), ...)) can trace.

CPU tests: state_dict keys / shapes are the ones the REFERENCE's own networks produced (tests/golden/*.npz record them), the twin's
forward reproduces the reference's own forward (fixtures written by tools/make_golden*.py from the imported reference) and the oracle's,
and it traces on the host (torch.jit.trace stands in for torch.onnx.export where the `onnx` package is absent, as in this image).
GPU tests: twin forward == the HIP forward of the module it was built from, 1e-3."""
import json
from argparse import Namespace

import numpy as np
import pytest
import torch

import helpers as H
from oracle import detfill, ref_cpu
from oracle import ref_spade_cpu as R

TOL = 1e-3


def _student(tag, norm, track):
    g = H.load(f'forward_{tag}.npz')
    opt = H.make_opt(norm=norm, track=track)
    shapes = H.sd_from_shapes(g['student_shapes'])
    net = H.student_from_shapes(opt, shapes)
    sd = detfill.fill_state_dict(shapes, H.SEED_S)
    net.load_state_dict(sd)
    return g, opt, net, sd


@pytest.mark.parametrize('tag,norm,track', [('in', 'instance', False), ('bn', 'batch', True)])
def test_inception_twin_has_the_reference_state_dict_and_forward(tag, norm, track):
    from cat_amd import export
    g, opt, net, sd = _student(tag, norm, track)
    net.train()
    twin = export.to_reference_module(net)
    # the reference's own pruned student recorded these keys / shapes (tools/make_golden.py)
    want = [(k, tuple(s)) for k, s in json.loads(str(g['student_shapes']))]
    got = [(k, tuple(v.shape)) for k, v in twin.state_dict().items()]
    assert got == want
    for k, v in twin.state_dict().items():
        assert v.device.type == 'cpu' and v.is_contiguous() and torch.equal(v, sd[k]), k
    assert all(type(m).__module__.startswith(('torch.nn', 'cat_amd.export')) for m in twin.modules())
    assert all(m.training for m in twin.modules())
    # forward in training mode (batch statistics) = the REFERENCE's own run of this student on this input
    x = detfill.images((1, 3, 256, 256), H.SEED_X)
    with torch.no_grad():
        y = twin(x)
    assert H.rel_err(H.sub(y, 3, 8), g['out']) < 2e-5
    # and a checkpoint written from the twin loads back into a cat_amd generator (interchange both ways)
    net2 = H.student_from_shapes(opt, H.sd_from_shapes(g['student_shapes']))
    net2.load_state_dict(twin.state_dict())


def test_teacher_twin_matches_the_reference_forward_and_traces():
    from cat_amd import export, networks
    g = H.load('forward_bn.npz')
    opt = H.make_opt(norm='batch', track=True, ndf=128)
    T = networks.define_G(3, 3, 64, 'inception_9blocks', 'batch', 0, 'normal', 0.02, [], opt=opt)
    T.load_state_dict(H.teacher_sd(opt))
    T.eval()
    twin = export.to_reference_module(T)
    assert not any(m.training for m in twin.modules())
    xt = detfill.images((1, 3, 64, 64), H.SEED_X + 1)
    with torch.no_grad():
        yt = twin(xt)
        yo, _ = ref_cpu.inception_generator(H.teacher_sd(opt), xt, H.cfg_for('batch'), training=False)
    assert H.rel_err(H.sub(yt, 3, 4), g['teacher_out']) < 3e-4          # the reference's own define_G teacher forward
    assert H.rel_err(yt.numpy(), yo.numpy()) < 1e-5
    # what torch.onnx.export does first: trace the module on host tensors, with a dynamic batch
    traced = torch.jit.trace(twin, xt)
    x2 = detfill.images((2, 3, 64, 64), H.SEED_X + 2)
    with torch.no_grad():
        assert torch.allclose(traced(x2), twin(x2), atol=1e-6)


def test_onnx_export_when_the_package_is_present(tmp_path):
    onnx = pytest.importorskip('onnx')
    from cat_amd import export
    g, opt, net, sd = _student('in', 'instance', False)
    path = str(tmp_path / 'netG_student.onnx')
    export.export_onnx(net.eval(), torch.randn(1, 3, 64, 64), path)
    onnx.checker.check_model(onnx.load(path))


def _spade_fixture():
    g = H.load('spade_step.npz')
    o = json.loads(str(g['opt']))
    o['gpu_ids'] = []
    o['data_height'], o['data_width'], o['data_channel'] = int(g['h']), int(g['w']), o['semantic_nc']
    return g, Namespace(**o)


def _spade_G(opt, ngf, sd, norm_G='spadesyncbatch3x3'):
    from cat_amd import networks
    o = Namespace(**vars(opt))
    o.ngf, o.norm_G = ngf, norm_G
    G = networks.define_G(opt.input_nc, 3, ngf, 'inception_spade', 'instance', 0, 'xavier', 0.02, [], opt=o)
    G.load_state_dict(sd)
    return G


def test_spade_twin_has_the_reference_state_dict_and_forward():
    from cat_amd import export
    g, opt = _spade_fixture()
    for key, ngf, seed in (('T_shapes', opt.teacher_ngf, 111), ('S_shapes', opt.student_ngf, 121)):
        sd = detfill.fill_state_dict(H.sd_from_shapes(g[key]), seed, gamma_abs_normal=True)
        G = _spade_G(opt, ngf, sd).eval()
        twin = export.to_reference_module(G)
        want = [(k, tuple(s)) for k, s in json.loads(str(g[key]))]
        assert [(k, tuple(v.shape)) for k, v in twin.state_dict().items()] == want      # the reference's own generator recorded these
        lab = torch.from_numpy(g['label'].astype(np.int64))
        sem = R.preprocess_input(lab, torch.from_numpy(g['instance']), opt.input_nc)
        cfgG = dict(crop_size=opt.crop_size, aspect_ratio=opt.aspect_ratio, num_upsampling_layers=opt.num_upsampling_layers)
        with torch.no_grad():
            y = twin(sem)
            yo = R.inception_spade_generator(sd, sem, cfgG, training=False)
        yo = yo[0] if isinstance(yo, tuple) else yo
        assert H.rel_err(y.numpy(), yo.numpy()) < 1e-5
        traced = torch.jit.trace(twin, sem[:1])
        with torch.no_grad():
            assert torch.allclose(traced(sem), y, atol=1e-6)


def test_spadeinstance_twin_reproduces_the_reference_run():
    """norm_G = 'spadeinstance3x3': tests/golden/spade_instance_fwd.npz holds the REFERENCE's own training-mode forward."""
    from cat_amd import export
    g = H.load('spade_instance_fwd.npz')
    _, opt = _spade_fixture()
    sd = detfill.fill_state_dict(H.sd_from_shapes(g['shapes']), 701, gamma_abs_normal=True)          # tools/make_golden_spade.py::spadeinstance_golden
    G = _spade_G(opt, 6, sd, 'spadeinstance3x3').train()
    twin = export.to_reference_module(G)
    assert [(k, tuple(v.shape)) for k, v in twin.state_dict().items()] == [(k, tuple(s)) for k, s in json.loads(str(g['shapes']))]
    lab, ins = torch.from_numpy(g['label'].astype(np.int64)), torch.from_numpy(g['instance'])
    sem = R.preprocess_input(lab, ins, opt.input_nc)
    with torch.no_grad():
        y = twin(sem)          # training mode: batch statistics in the gamma|beta nets, running statistics move once
    assert H.rel_err(y[:, :, ::2, ::2].numpy(), g['y_sub']) < 2e-5
    for k in (f[4:] for f in g.files if f.startswith('buf:')):
        assert H.rel_err(twin.state_dict()[k].numpy(), g['buf:' + k]) < 2e-5, k


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('tag,norm,track,train', [('in', 'instance', False, True), ('bn', 'batch', True, True), ('bn', 'batch', True, False)])
def test_inception_twin_equals_the_hip_forward(tag, norm, track, train):
    from cat_amd import export, ops
    g, opt, net, sd = _student(tag, norm, track)
    net = net.cuda()
    net.train(train)
    twin = export.to_reference_module(net)          # built FROM the device module: padded channels-last weights back to dense OIHW
    for k, v in twin.state_dict().items():
        assert torch.equal(v, sd[k]), k
    x = detfill.images((2, 3, 256, 256), H.SEED_X)
    with torch.no_grad():
        y_hip = net(ops.to_nhwc(x.cuda())).float().cpu()
        y_twin = twin(x)
    assert H.rel_err(y_hip.numpy(), y_twin.numpy()) < TOL


@pytest.mark.gpu
def test_spade_twin_equals_the_hip_forward():
    from cat_amd import export, ops
    g, opt = _spade_fixture()
    sd = detfill.fill_state_dict(H.sd_from_shapes(g['S_shapes']), 121, gamma_abs_normal=True)
    G = _spade_G(opt, opt.student_ngf, sd).cuda().eval()
    twin = export.to_reference_module(G)
    lab = torch.from_numpy(g['label'].astype(np.int64))
    sem = R.preprocess_input(lab, torch.from_numpy(g['instance']), opt.input_nc)
    with torch.no_grad():
        y_hip = G(ops.to_nhwc(sem.cuda())).float().cpu()
        y_twin = twin(sem)
    assert H.rel_err(y_hip.numpy(), y_twin.numpy()) < TOL


def test_spade_block_sizes_match_the_forward_schedule():
    """InceptionSPADEGenerator._block_sizes (the resolutions the gamma|beta pre-pass prepares the segmentation pyramid at, round 5) against the
    resolutions the SPADE layers actually run at -- observed on the host twin of the same generator with forward pre-hooks."""
    from cat_amd import export
    g, opt = _spade_fixture()
    for ups in ('normal', 'more', 'most'):
        o = Namespace(**vars(opt))
        o.num_upsampling_layers = ups
        from cat_amd import networks
        o.ngf, o.norm_G = 4, 'spadesyncbatch3x3'
        G = networks.define_G(opt.input_nc, 3, 4, 'inception_spade', 'instance', 0, 'xavier', 0.02, [], opt=o).eval()
        twin = export.to_reference_module(G)
        seen = {}
        for name, _ in G._block_sizes():
            getattr(twin, name).spade.register_forward_pre_hook(lambda m, a, name=name: seen.__setitem__(name, tuple(a[0].shape[2:])))
        h = G.sh * (2 ** {'normal': 5, 'more': 6, 'most': 7}[ups])
        w = G.sw * (2 ** {'normal': 5, 'more': 6, 'most': 7}[ups])
        with torch.no_grad():
            y = twin(torch.zeros(1, opt.input_nc + (0 if getattr(opt, 'no_instance', False) else 1), h, w))
        want = {k: v for k, v in G._block_sizes() if len(getattr(G, k).res_ops) + len(getattr(G, k).dw_ops)}      # a block without branches skips its SPADE layer
        assert want == seen and len(seen) >= 6, (ups, G._block_sizes(), seen)
        assert tuple(y.shape[2:]) == (h, w)
