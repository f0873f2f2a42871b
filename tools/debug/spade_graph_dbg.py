"""Where do the eager and the graph-replayed GauGAN steps part ways?  Eager on the default stream vs eager on a side stream vs GraphedStep,
with the fused SPADE units on / off (CAT_FUSED_SPADE)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import test_spade_gpu as TS
from oracle import detfill
from cat_amd.graph import GraphedStep
from cat_amd import fused_spade as FS

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
order = [0, 0, 0, 1, 2]


def losses(m):
    return {k.split('/')[-1]: round(v, 6) for k, v in m.get_current_losses().items() if k.split('/')[-1] in ('G_gan', 'G_feat', 'G_vgg', 'D_fake')}


a = TS.build_spade_distiller(opt, sds)
for i, b in enumerate(order):
    a.set_input(batches[b]); a.optimize_parameters(i)
    torch.cuda.synchronize(); print('default stream step', i, losses(a), dict(FS.STATS))
b_ = TS.build_spade_distiller(opt, sds)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for i, b in enumerate(order):
        b_.set_input(batches[b]); b_.optimize_parameters(i)
        torch.cuda.synchronize(); print('side stream    step', i, losses(b_), dict(FS.STATS))
c = TS.build_spade_distiller(opt, sds)
step = GraphedStep(c, batches[0], warmup=3)
print('graph after warm-up   ', losses(c), dict(FS.STATS))
for i in (1, 2):
    step(batches[i]); torch.cuda.synchronize(); print('graph replay batch', i, losses(c))
