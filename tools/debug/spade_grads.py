import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
import test_spade_gpu as T
from oracle import detfill, ref_spade_cpu as R
from cat_amd import ops
g, opt, lab, ins, img, sds, cfg = T.fixture()
sem = R.preprocess_input(lab, ins, opt.input_nc)
sd = sds['S']
G = T.make_G(opt, opt.student_ngf, sd, True)
gsem = ops.onehot_edges(lab.cuda(), ins.cuda(), opt.input_nc)
r = detfill.normal((2, 3, int(g['h']), int(g['w'])), 77)
y = G(gsem)
y.backward(T.nhwc(r))
ref_sd = {k: v.clone().requires_grad_(R.SpadeState._is_param(k)) for k, v in sd.items()}
yr, _ = R.inception_spade_generator(ref_sd, sem, cfg['G'], True, False)
(yr * r).sum().backward()
gmax = max(float(v.grad.abs().max()) for v in ref_sd.values() if v.grad is not None)
filt = sys.argv[1] if len(sys.argv) > 1 else 'up_3'
for k, p in G.named_parameters():
    if filt in k:
        rg = ref_sd[k].grad
        own = float(rg.abs().max())
        err = float((p.grad.cpu().double() - rg.double()).abs().max())
        print(f'{k:46s} own/gmax {own/gmax:9.2e} err/own {err/max(own,1e-30):9.2e} err/gmax {err/gmax:9.2e}')
