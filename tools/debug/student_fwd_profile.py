"""Serial per-op timing of the C2 student forward alone (train-mode norms, no grad)."""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from cat_amd import _lib as L, ops
from oracle import detfill
L.load()
args = argparse.Namespace(workload='c2', batch=16, size=256, target_flops=4.6e9)
ops.set_branch_streams(False)
model, opt = bench.build_model(args, 0)
model.set_input({'A': detfill.images((16, 3, 256, 256), 1).cuda(), 'B': detfill.images((16, 3, 256, 256), 2).cuda(), 'A_paths': [], 'B_paths': []})
net, x = model.netG_student, model.real_A
with torch.no_grad():
    for _ in range(3): net(x)
torch.cuda.synchronize()
rec = collections.OrderedDict(); orig = L.call
def timed(name, *cargs):
    key = name
    if name.startswith(('cat_conv2d', 'cat_dwconv2d')):
        g = cargs[0]._obj; key = f'{name[4:]} {g.H}x{g.W} {g.Cin}->{g.Cout} k{g.kh} s{g.stride}'
    elif name.startswith('cat_norm'):
        g = cargs[0]._obj; key = f'{name[4:]} HW{g.HW} C{g.C}'
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *cargs); e1.record()
    rec.setdefault(key, []).append((e0, e1))
L.call = timed
with torch.no_grad(): net(x)
torch.cuda.synchronize(); L.call = orig
rows = sorted(((sum(a.elapsed_time(b) for a, b in v), k, len(v)) for k, v in rec.items()), reverse=True)
print('serial total %.2f ms over %d calls' % (sum(r[0] for r in rows), sum(r[2] for r in rows)))
byk = collections.Counter()
for ms, k, n in rows: byk[k.split(' ')[0]] += ms
print(dict((k, round(v, 2)) for k, v in byk.most_common()))
for ms, k, n in rows[:30]: print('%7.3f ms x%3d  %s' % (ms, n, k))
