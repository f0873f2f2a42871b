"""CPU: the product's shrink_model / model_profiling (host logic, cat_amd/prune.py) against the reference's results
(tests/golden/shrink_*.npz): threshold, MAC counts, channel lists and the masked weight copy are BIT-EXACT."""
import json

import numpy as np
import pytest
import torch

import helpers as H
from oracle import detfill


@pytest.mark.parametrize('tag,norm,track,target', [('in', 'instance', False, 2.6e9), ('bn', 'batch', True, 4.6e9)])
def test_shrink_matches_reference_bit_exactly(tag, norm, track, target):
    from cat_amd import networks, prune
    g = H.load(f'shrink_{tag}.npz')
    opt = H.make_opt(norm=norm, track=track, target_flops=target)
    T = networks.define_G(3, 3, 64, 'inception_9blocks', norm, 0, 'normal', 0.02, [], opt=opt)
    T.load_state_dict(H.teacher_sd(opt))
    T.eval()
    macs, _ = prune.model_profiling(T, 256, 256)
    assert [macs, T.down_sampling.n_macs, T.features.n_macs, T.up_sampling.n_macs] == list(g['t_macs'])
    # gamma vectors of the rebuilt teacher equal the recorded ones (the fixture is self-consistent)
    assert np.array_equal(T.down_sampling[2].weight.detach().numpy(), g['g_down0'])
    names = prune.get_bn_to_prune(T)
    assert names[0] == 'features.0.res_ops.0.1.1.weight' and len(names) == 54
    thr, searched = prune.search_threshold(T, target, opt)
    assert np.float32(thr.item()) == g['thr']
    assert searched == int(g['s_macs'][0])
    import copy
    S = copy.deepcopy(T)
    trunk, masks = prune._apply_structure(S, T, thr, opt, copy_weights=True)
    ref_cfg = json.loads(str(g['cfg']))
    assert [S.down_sampling[i].num_features for i in (2, 5, 8)] == ref_cfg['down']
    assert [S.up_sampling[i].num_features for i in (1, 4)] == ref_cfg['up']
    assert [[b.res_channels, b.dw_channels] for b in S.features] == ref_cfg['blocks']
    assert trunk == ref_cfg['down'][2]
    macs, _ = prune.model_profiling(S, 256, 256)
    assert [macs, S.down_sampling.n_macs, S.features.n_macs, S.up_sampling.n_macs] == list(g['s_macs'])
    ssd = S.state_dict()
    assert [[k, list(v.shape)] for k, v in ssd.items()] == json.loads(str(g['student_shapes']))
    for key in g.files:
        if key.startswith('copied:'):
            assert np.array_equal(ssd[key[7:]].numpy(), g[key]), key      # index selection is bit-exact


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_shrink_spade_bit_exact(tag):
    """shrink_spade_model's architecture search (utils/common.py:710-812) against the reference's own run
    (tests/golden/spade_shrink.npz): threshold (float32), searched n_macs, every channel list and the resulting state_dict
    shapes are integers / bit patterns and must match exactly.  Host-side, runs on CPU (shape propagation, no forward)."""
    import json
    from argparse import Namespace
    from cat_amd import networks
    from cat_amd.prune import spade_search, model_profiling
    g = H.load('spade_shrink.npz')
    o = json.loads(str(g['opt']))
    o['target_flops'], o['prune_cin_lb'] = float(g[f'{tag}_target']), int(g[f'{tag}_lb'])
    opt = Namespace(**o)
    topt = Namespace(**o)
    topt.ngf, topt.norm_G = opt.teacher_ngf, opt.teacher_norm_G
    T = networks.define_G(opt.input_nc, 3, opt.teacher_ngf, 'inception_spade', 'instance', 0, 'xavier', 0.02, [], opt=topt)
    ref_shapes = json.loads(str(g['T_shapes']))
    assert [k for k, _ in ref_shapes] == list(T.state_dict().keys())
    T.load_state_dict(detfill.fill_state_dict(T.state_dict(), 112, gamma_abs_normal=True))
    thr, searched, S, ngf_stu = spade_search(T, opt.target_flops, opt)
    assert np.float32(float(thr)) == g[f'{tag}_thr'], (float(thr), float(g[f'{tag}_thr']))
    assert int(searched) == int(g[f'{tag}_n_macs'])
    cfg = json.loads(str(g[f'{tag}_cfg']))
    assert S.fc.out_channels == cfg['fc'] and ngf_stu * 16 == cfg['fc']
    for name, blk in S.get_named_block_list().items():
        ref = cfg['blocks'][name]
        got = dict(input_dim=blk.input_dim, output_dim=blk.output_dim, res=blk.res_channels, dw=blk.dw_channels,
                   spade_res=blk.spade.res_channels, spade_dw=blk.spade.dw_channels, shortcut=blk.shortcut is not None)
        assert got == ref, (name, got, ref)
    s_shapes = json.loads(str(g[f'{tag}_S_shapes']))
    sd = S.state_dict()
    assert [k for k, _ in s_shapes] == list(sd.keys())
    assert all(list(sd[k].shape) == shp for k, shp in s_shapes)
    assert model_profiling(S, opt.data_height, opt.data_width, channel=opt.data_channel)[0] == int(g[f'{tag}_n_macs'])


def test_load_pretrained_weight_bit_exact():
    """cat_amd.weight_transfer.load_pretrained_weight (ngf 32 -> 20) against the reference's run: the transfer is pure top-k index
    selection + copy, so every student tensor must equal the reference's exactly (fingerprints of all 472 tensors, 5 in full)."""
    import json
    from cat_amd import networks
    from cat_amd.weight_transfer import load_pretrained_weight
    g = H.load('weight_transfer.npz')
    opt = H.make_opt(norm='instance', track=False, gpu_ids=[])
    A = networks.define_G(3, 3, 32, 'inception_9blocks', 'instance', 0, 'normal', 0.02, [], opt=opt)
    B = networks.define_G(3, 3, 20, 'inception_9blocks', 'instance', 0, 'normal', 0.02, [], opt=opt)
    assert [k for k, _ in json.loads(str(g['A_shapes']))] == list(A.state_dict().keys())
    A.load_state_dict(detfill.fill_state_dict(A.state_dict(), 501))
    B.load_state_dict(detfill.fill_state_dict(B.state_dict(), 502))
    load_pretrained_weight('inception_9blocks', 'inception_9blocks', A, B, 32, 20)
    sd = B.state_dict()
    assert [k for k, _ in json.loads(str(g['B_shapes']))] == list(sd.keys())
    for (k, v), ref in zip(sd.items(), g['B_fingerprints']):
        d = v.double().reshape(-1)
        w = torch.arange(1, d.numel() + 1, dtype=torch.float64)
        got = np.array([float(d.sum()), float((d * w).sum()), float((d * d).sum())])
        assert np.array_equal(got, ref), (k, got, ref)
    for key in g.files:
        if key.startswith('B:'):
            np.testing.assert_array_equal(sd[key[2:]].numpy(), g[key])


def test_load_pretrained_weight_spade_bit_exact():
    """The `inception_spade` branch of load_pretrained_weight (reference utils/weight_transfer.py:137-212, 267-288) against the reference's
    own run (tests/golden/spade_weight_transfer.npz, ngf 8 -> 6): shapes after the transfer -- including the gamma|beta convolutions the
    reference cuts down to their gamma rows -- and exact fingerprints of all 1062 student tensors."""
    import json
    from argparse import Namespace
    from cat_amd import networks
    from cat_amd.weight_transfer import load_pretrained_weight
    g = H.load('spade_weight_transfer.npz')
    o = json.loads(str(g['opt']))

    def G(ngf):
        opt = Namespace(**o)
        opt.ngf, opt.norm_G, opt.gpu_ids = ngf, 'spadesyncbatch3x3', []
        return networks.define_G(opt.input_nc, 3, ngf, 'inception_spade', 'instance', 0, 'xavier', 0.02, [], opt=opt)
    A, B = G(8), G(6)
    assert [[k, list(v.shape)] for k, v in A.state_dict().items()] == json.loads(str(g['A_shapes']))
    assert [[k, list(v.shape)] for k, v in B.state_dict().items()] == json.loads(str(g['B_shapes_before']))
    A.load_state_dict(detfill.fill_state_dict(A.state_dict(), 601, gamma_abs_normal=True))
    B.load_state_dict(detfill.fill_state_dict(B.state_dict(), 602))
    load_pretrained_weight('inception_spade', 'inception_spade', A, B, 8, 6)
    sd = B.state_dict()
    after = json.loads(str(g['B_shapes_after']))
    assert [[k, list(v.shape)] for k, v in sd.items()] == after
    assert sum(1 for a, b in zip(json.loads(str(g['B_shapes_before'])), after) if a != b) == 84
    for (k, v), ref in zip(sd.items(), g['B_fingerprints']):
        d = v.double().reshape(-1)
        w = torch.arange(1, d.numel() + 1, dtype=torch.float64)
        got = np.array([float(d.sum()), float((d * w).sum()), float((d * d).sum())])
        assert np.array_equal(got, ref), (k, got, ref)
    for key in g.files:
        if key.startswith('B:'):
            np.testing.assert_array_equal(sd[key[2:]].numpy(), g[key])


def test_survey_canonical_students_are_reproduced():
    """SURVEY section 8(d)'s canonical synthetic nets, built with cat_amd's own define_G / shrink search on the host: teacher
    `define_G(3, 3, 64, 'inception_9blocks', ...)` after torch.manual_seed(233), every norm weight overwritten with |N(0, 1)| drawn in module order
    from torch.Generator().manual_seed(7), pruned to the launch scripts' budgets with prune_cin_lb = 16.  S_2.6 (InstanceNorm, 2.6e9): threshold,
    n_macs and every channel list are the ones the survey's probe of the REFERENCE printed; S_4.6 (BatchNorm, 4.6e9): the channel lists are
    (its threshold / n_macs read 0.9490039 / 4 591 284 224 here against 0.9516321 / 4 586 242 048 in the survey's text, with identical
    structure -- the survey's own conv-MAC figure for this student, 4 591.3 M, is the number found here).  bench.py draws the norm scales from
    the platform-independent PCG64 stream instead: same recipe, trunk 77 instead of 82."""
    import copy
    from cat_amd import networks, prune

    def canonical(norm, track, target):
        opt = H.make_opt(norm=norm, track=track, target_flops=target, prune_cin_lb=16)
        torch.manual_seed(233)
        T = networks.define_G(3, 3, 64, 'inception_9blocks', norm, 0, 'normal', 0.02, [], opt=opt)
        gen = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for m in T.modules():
                if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.InstanceNorm2d)) and getattr(m, 'weight', None) is not None:
                    m.weight.copy_(torch.randn(m.weight.shape, generator=gen).abs())
        T.eval()
        thr, searched = prune.search_threshold(T, target, opt)
        S = copy.deepcopy(T)
        prune._apply_structure(S, T, thr, opt, copy_weights=True)
        macs, _ = prune.model_profiling(S, 256, 256)
        return float(thr), searched, macs, S

    thr, searched, macs, S = canonical('instance', False, 2.6e9)
    assert '%.7f' % thr == '1.1518520' and searched == macs == 2566197248
    assert [S.down_sampling[i].num_features for i in (2, 5, 8)] == [16, 25, 56] and [S.up_sampling[i].num_features for i in (1, 4)] == [30, 16]
    want = [([11, 14, 10], [8, 12, 11]), ([11, 11, 11], [14, 7, 11]), ([6, 15, 10], [13, 11, 9]), ([11, 7, 11], [14, 12, 11]), ([4, 13, 15], [10, 16, 10]),
            ([11, 7, 9], [10, 10, 9]), ([7, 10, 7], [14, 9, 8]), ([7, 10, 8], [10, 9, 6]), ([7, 9, 9], [14, 10, 10])]
    assert [(b.res_channels, b.dw_channels) for b in S.features] == want
    thr, searched, macs, S = canonical('batch', True, 4.6e9)
    assert searched == macs <= 4.6e9
    assert [S.down_sampling[i].num_features for i in (2, 5, 8)] == [22, 37, 82] and [S.up_sampling[i].num_features for i in (1, 4)] == [38, 16]
    assert (S.features[0].res_channels, S.features[0].dw_channels) == ([17, 17, 14], [12, 13, 15])
