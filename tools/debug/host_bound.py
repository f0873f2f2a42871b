"""Is the step host-bound?  Time to ISSUE a step (python + ctypes launches, no sync) vs time until the GPU is done."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from cat_amd import _lib as L
L.load()
wl = sys.argv[1] if len(sys.argv) > 1 else 'spade'
args = argparse.Namespace(workload=wl, batch=4 if wl == 'spade' else 16, size=256, target_flops=5.6e9 if wl == 'spade' else 4.6e9)
if wl == 'spade':
    model, opt = bench.build_spade_model(args, 0)
    batches = bench.spade_batches(args, 0, 2)
else:
    from oracle import detfill
    model, opt = bench.build_model(args, 0)
    batches = [{'A': detfill.images((16, 3, 256, 256), 1 + i).cuda(), 'B': detfill.images((16, 3, 256, 256), 9 + i).cuda(), 'A_paths': [], 'B_paths': []} for i in range(2)]
def step(i):
    model.set_input(batches[i % 2]); model.optimize_parameters(i)
for i in range(3): step(i)
torch.cuda.synchronize()
iss, tot = [], []
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(10 + i); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    iss.append(t1 - t0); tot.append(t2 - t0)
print(wl, 'issue ms', [round(1e3 * x, 1) for x in iss], 'total ms', [round(1e3 * x, 1) for x in tot])
n = {'c': 0}
orig = L.call
def cnt(name, *a):
    n['c'] += 1; return orig(name, *a)
L.call = cnt
step(99); torch.cuda.synchronize()
print('C-ABI calls per step', n['c'])
