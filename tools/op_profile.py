"""Per-call-site timing of one distillation step: every C-ABI call is bracketed with HIP events (serial execution, branch streams
off) and aggregated by (entry point, geometry).  GPU only.

  python tools/op_profile.py --workload spade [--top 50]
"""
import argparse
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='spade')
    ap.add_argument('--top', type=int, default=50)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    import bench
    from cat_amd import _lib as L, ops
    L.load()
    args = argparse.Namespace(workload=a.workload, batch=4 if a.workload == 'spade' else 16, size=256,
                              target_flops=5.6e9 if a.workload == 'spade' else 4.6e9)
    ops.set_branch_streams(False)
    if a.workload == 'spade':
        model, opt = bench.build_spade_model(args, 0)
        batches = bench.spade_batches(args, 0, 2)
    else:
        from oracle import detfill
        model, opt = bench.build_model(args, 0)
        model.teacher_side_stream = False
        batches = [{'A': detfill.images((16, 3, 256, 256), 1 + i).cuda(), 'B': detfill.images((16, 3, 256, 256), 9 + i).cuda(),
                    'A_paths': [], 'B_paths': []} for i in range(2)]

    def step(i):
        model.set_input(batches[i % 2])
        model.optimize_parameters(i)
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    rec = collections.OrderedDict()
    orig = L.call

    def timed(name, *cargs):
        key = name
        flops = 0.0
        if name.startswith(('cat_conv2d', 'cat_dwconv2d')):
            g = cargs[0]._obj
            key = f'{name[4:]} N{g.N} {g.H}x{g.W} {g.Cin}->{g.Cout} k{g.kh} s{g.stride} p{g.pad}'
            flops = 2.0 * g.N * g.Ho * g.Wo * g.Cout * g.kh * g.kw * (g.Cin if 'dw' not in name else 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *cargs)
        e1.record()
        r = rec.setdefault(key, [0, [], flops])
        r[0] += 1
        r[1].append((e0, e1))

    L.call = timed
    ops.L.call = timed
    for i in range(a.steps):
        step(10 + i)
    torch.cuda.synchronize()
    L.call = orig
    ops.L.call = orig
    rows = []
    for k, (cnt, evs, fl) in rec.items():
        ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / a.steps
        rows.append((ms, k, cnt / a.steps, fl))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f'total {tot:.2f} ms/step over {sum(r[2] for r in rows):.0f} calls')
    by_name = collections.Counter()
    for ms, k, cnt, fl in rows:
        by_name[k.split(' ')[0]] += ms
    print('by entry point:', ', '.join(f'{k} {v:.2f}' for k, v in by_name.most_common(14)))
    for ms, k, cnt, fl in rows[:a.top]:
        if a.only and a.only not in k:
            continue
        tf = fl * cnt / ms / 1e9 if fl and ms > 0 else 0
        print(f'{ms:8.3f} ms  x{cnt:5.1f}  {tf:6.1f} TF/s  {k}')


if __name__ == '__main__':
    main()
