import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cat_amd import ops, _lib as L, nn as cnn
L.load()
for (n, c, h, w) in [(16, 25, 256, 256), (16, 23, 256, 256), (16, 77, 64, 64), (16, 128, 128, 128)]:
    bn = cnn.BatchNorm2d(c).cuda().train()
    x = ops.to_nhwc(torch.randn(n, c, h, w, device='cuda')).requires_grad_(True)
    for _ in range(3):
        y = bn(x, fuse_act=cnn.ReLU()); y.backward(y.detach())
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); y = bn(x, fuse_act=cnn.ReLU()); e[1].record(); y.backward(y.detach()); e[2].record(); torch.cuda.synchronize()
    mb = n * h * w * ops.cs_for(c) * 4 / 1e6
    print((n, c, h, w), 'tensor MB %.0f' % mb, 'fwd %.3f ms (%.2f TB/s)' % (e[0].elapsed_time(e[1]), 3 * mb / e[0].elapsed_time(e[1]) / 1e3),
          'bwd %.3f ms (%.2f TB/s)' % (e[1].elapsed_time(e[2]), 5 * mb / e[1].elapsed_time(e[2]) / 1e3))
