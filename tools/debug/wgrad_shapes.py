"""List every cat_conv2d_wgrad / cat_tconv call of ONE eager C2 step with its geometry and its own duration (device-synchronised around
each call: durations are serial kernel times, not the overlapped step)."""
import argparse, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('CAT_BRANCH_STREAMS', '0')
import torch
import bench
from cat_amd import _lib as L, synthetic
L.load()
args = argparse.Namespace(workload='c2', batch=16, size=256, target_flops=4.6e9)
model, opt = bench.build_model(args, 0)
batch = {'A': synthetic.images((16, 3, 256, 256), 1).cuda(), 'B': synthetic.images((16, 3, 256, 256), 2).cuda(), 'A_paths': [], 'B_paths': []}
for i in range(3):
    model.set_input(batch); model.optimize_parameters(i)
torch.cuda.synchronize()
rows = collections.OrderedDict()
orig = L.call
WATCH = tuple(a for a in sys.argv[1:]) or ('cat_conv2d_wgrad',)
def call(name, *a):
    if name not in WATCH:
        return orig(name, *a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(name, *a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
    g = a[0]._obj
    key = (name, g.N, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout, g.kh, g.stride, g.pad)
    e = rows.setdefault(key, [0, 0.0]); e[0] += 1; e[1] += dt
    return r
L.call = call
import cat_amd.ops as ops, cat_amd.fused_block as fb
model.set_input(batch); model.optimize_parameters(3)
torch.cuda.synchronize()
tot = 0
for k, (cnt, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    name, n, h, w, cin, ho, wo, cout, kk, s, p = k
    gf = 2.0 * n * ho * wo * cout * kk * kk * cin / 1e9
    print(f'{name:18s} N{n} {h}x{w} {cin:4d}->{cout:4d} k{kk} s{s} p{p}  x{cnt}  {us/cnt:8.1f} us  {gf * cnt / us * 1e3:6.1f} TF' if us else k)
    tot += us
print('total us', tot)
