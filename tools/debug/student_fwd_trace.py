"""Replay the captured C2 student forward a few times (for `rocprofv3 --kernel-trace`): the per-kernel start / end timestamps show
how the six branch streams of every block overlap.  tools/debug/trace_fold.py folds the CSV."""
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
    for _ in range(int(os.environ.get('REPS', '3'))):
        g.replay()
        torch.cuda.synchronize()
print('done')
