"""Is the eager GauGAN step deterministic run to run with the fused SPADE units (all / train form only / frozen form only / off)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import test_spade_gpu as TS
from oracle import detfill
from cat_amd import fused_spade as FS, ops

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
for mode in (os.environ.get('MODES', 'all,train,frozen,off').split(',')):
    mode = None if mode == 'all' else mode
    FS._ONLY = mode if mode in ('train', 'frozen') else None
    FS.set_enabled(mode != 'off')
    for streams in ((True,) if os.environ.get('MODES') else (True, False)):
        ops.set_branch_streams(streams)
        out = []
        for rep in range(int(os.environ.get('REPS', '3'))):
            m = TS.build_spade_distiller(opt, sds)
            trace = []
            for i, b in enumerate([0, 0, 1, 2]):
                m.set_input(batches[b]); m.optimize_parameters(i)
                torch.cuda.synchronize()
                l = m.get_current_losses()
                trace.append(round(l['G_loss/G_feat'], 5))
            out.append(tuple(trace))
        print('mode', mode, 'branch streams', streams, out)
