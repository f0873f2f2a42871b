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
