import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_spade_gpu as TS
from oracle import detfill
from cat_amd.graph import GraphedStep
g, opt, lab, ins, img, sds, cfg = TS.fixture()
opt.isTrain, opt.distiller, opt.log_dir = True, 'spade', '/tmp/cat_amd_logs'
rng = np.random.default_rng(9)
h, w, n = int(g['h']), int(g['w']), int(g['n'])
batches = []
for i in range(3):
    lab_i = np.repeat(np.repeat(rng.integers(0, opt.input_nc, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32)
    ins_i = np.repeat(np.repeat(rng.integers(0, 99, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32)
    batches.append({'label': torch.from_numpy(lab_i).cuda(), 'instance': torch.from_numpy(ins_i).cuda(), 'image': detfill.images((n, 3, h, w), 700 + i).cuda(), 'path': []})
def run_eager():
    m = TS.build_spade_distiller(opt, sds)
    out = []
    for i, b in enumerate([0, 0, 0, 1, 2]):
        m.set_input(batches[b]); m.optimize_parameters(i)
        out.append(dict(m.get_current_losses()))
    return out
a, b = run_eager(), run_eager()
mg = TS.build_spade_distiller(opt, sds)
st = GraphedStep(mg, batches[0], warmup=3)
gl = []
for bi in (1, 2):
    st(batches[bi]); gl.append(dict(mg.get_current_losses()))
for k in a[0]:
    print(f'{k:28s}', ' eagerA', [f'{x[k]:.7f}' for x in a[2:]], ' eagerB', [f'{x[k]:.7f}' for x in b[2:]], ' graph', [f'{x[k]:.7f}' for x in gl])
