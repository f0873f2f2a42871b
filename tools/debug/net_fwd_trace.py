"""Replay the captured forward of one C2 network a few times (for `rocprofv3 --kernel-trace --output-format csv`); fold the
CSV with tools/debug/trace_fold.py.     NET=teacher|student|disc [SERIAL=1] [REPS=3] python tools/debug/net_fwd_trace.py"""
import argparse, os, sys
if os.environ.get('SERIAL', '1') == '1':
    os.environ.setdefault('CAT_BRANCH_STREAMS', '0')      # one stream: every kernel's duration is its own
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from cat_amd import _lib as L, synthetic
L.load()
batch = int(os.environ.get('BATCH', '16'))
args = argparse.Namespace(workload='c2', batch=batch, size=256, target_flops=4.6e9)
model, opt = bench.build_model(args, 0)
model.set_input({'A': synthetic.images((batch, 3, 256, 256), 1).cuda(), 'B': synthetic.images((batch, 3, 256, 256), 2).cuda(), 'A_paths': [], 'B_paths': []})
which = os.environ.get('NET', 'teacher')
if which == 'disc':
    net, x = model.netD, torch.cat((model.real_A, model.real_B), 1)
else:
    net, x = (model.netG_teacher if which == 'teacher' else model.netG_student), model.real_A
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        net(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get('REPS', '3'))
    for _ in range(reps):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
print('%s forward: %.3f ms per replay' % (which, e0.elapsed_time(e1)))
