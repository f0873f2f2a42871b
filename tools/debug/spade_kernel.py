import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from oracle import detfill
from cat_amd import ops, _lib as L
def run(n, c, h, w):
    x = detfill.normal((n, c, h, w), 30) * 1.5 + 0.3
    gb = detfill.normal((n, 2 * c, h, w), 31) * 0.7
    rm, rv = torch.zeros(c), torch.ones(c)
    gx, ggb = ops.to_nhwc(x.cuda()).requires_grad_(True), ops.to_nhwc(gb.cuda()).requires_grad_(True)
    y = ops.SpadeFn.apply(gx, ggb, rm.cuda(), rv.cuda(), 1e-5, 0.1, L.ACT_RELU, 0.0)
    dy = detfill.normal(tuple(y.shape), 32)
    y.backward(ops.to_nhwc(dy.cuda()))
    xr, gbr = x.clone().requires_grad_(True), gb.clone().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, rm, rv, None, None, True, 0.1, 1e-5) * (1 + gbr[:, :c]) + gbr[:, c:])
    yr.backward(dy)
    e = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    print((n, c, h, w), 'y', e(y.detach(), yr.detach()), 'dgamma', e(ggb.grad[:, :c], gbr.grad[:, :c]), 'dbeta', e(ggb.grad[:, c:], gbr.grad[:, c:]),
          'dx', e(gx.grad, xr.grad), 'sum dbeta', e(ggb.grad[:, c:].sum((0, 2, 3)), gbr.grad[:, c:].sum((0, 2, 3))))
run(2, 12, 128, 256)
run(2, 12, 16, 16)
run(2, 24, 64, 128)
run(2, 96, 2, 4)
