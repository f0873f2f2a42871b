"""Time the captured C2 student forward (train mode, batch 16 @ 256x256): prints ms per forward (median of 5 x 20 graph replays).
A/B tool for kernel knobs:  CAT_PK_SPLITN=4 python tools/debug/student_fwd_time.py"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from cat_amd import _lib as L, synthetic
L.load()
args = argparse.Namespace(workload='c2', batch=16, size=256, target_flops=4.6e9)
model, opt = bench.build_model(args, 0)
model.set_input({'A': synthetic.images((16, 3, 256, 256), 1).cuda(), 'B': synthetic.images((16, 3, 256, 256), 2).cuda(), 'A_paths': [], 'B_paths': []})
net, x = model.netG_student, model.real_A
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        net(x)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
print('student forward ms: median %.4f  (min %.4f max %.4f)  %s' % (sorted(ts)[2], min(ts), max(ts), ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('CAT_'))))
