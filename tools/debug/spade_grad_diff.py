"""Which gradient tensors differ run to run after the first backward of the eager GauGAN step?  (debug aid)"""
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
lab_i = np.repeat(np.repeat(rng.integers(0, opt.input_nc, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32)
ins_i = np.repeat(np.repeat(rng.integers(0, 99, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32)
batch = {'label': torch.from_numpy(lab_i).cuda(), 'instance': torch.from_numpy(ins_i).cuda(),
         'image': detfill.images((n, 3, h, w), 700).cuda(), 'path': []}
FS._ONLY = os.environ.get('ONLY') or None
ops.set_branch_streams(True)


def snap(m):
    net = m.modules_on_one_gpu
    out = {}
    for k, p in net.netG_student.named_parameters():
        out['S.' + k] = p.grad.detach().clone() if p.grad is not None else None
    return out


ref = None
for rep in range(int(os.environ.get('REPS', '6'))):
    m = TS.build_spade_distiller(opt, sds)
    m.set_input(batch)
    m.set_requires_grad(m.modules_on_one_gpu.netD, False)
    m.optimizer_G.zero_grad()
    m.backward_G()
    torch.cuda.synchronize()
    s = snap(m)
    s['Sfake'] = m.Sfake_B.detach().clone()
    s['Tfake'] = m.Tfake_B.detach().clone()
    if ref is None:
        ref = s
        continue
    bad = []
    for k, v in s.items():
        if v is None:
            continue
        d = (v - ref[k]).abs().max().item()
        if d != 0.0:
            bad.append((k, d, ref[k].abs().max().item(), int(((v - ref[k]) != 0).sum().item()), v.numel()))
    print('rep', rep, 'differing tensors:', len(bad))
    for b in bad[:40]:
        print('   %-70s maxdiff %.3e  (max %.3e)  %d of %d' % b)
        print('      ref', ref[b[0]].flatten().tolist())
        print('      now', s[b[0]].flatten().tolist())
    names = [k for k in s if 'up_3' in k and ('conv.bias' in k)]
    for k in names:
        print('      ', k, s[k].flatten().tolist())
