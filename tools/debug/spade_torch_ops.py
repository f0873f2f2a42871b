"""Which torch-native (aten) GPU work does one eager GauGAN step still launch?  (Our own kernels go through the C-ABI and do not show up as
aten ops; anything listed here is a candidate for removal from the step.)   python tools/debug/spade_torch_ops.py [c2|spade]"""
import argparse, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'spade'
args = argparse.Namespace(size=256, batch=4 if wl == 'spade' else 16, target_flops=5.6e9 if wl == 'spade' else 4.6e9, workload=wl)
torch.cuda.set_device(0)
if wl == 'spade':
    model, opt = bench.build_spade_model(args, 0)
    batches = bench.spade_batches(args, 0, 2)
else:
    args.workload = 'c2'
    model, opt = bench.build_model(args, 0)
    from cat_amd import synthetic
    batches = [synthetic.batch(opt, args.batch, args.size, 100 + i) for i in range(2)] if hasattr(synthetic, 'batch') else None
for i in range(3):
    model.set_input(batches[i % 2]); model.optimize_parameters(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for i in range(STEPS):
        model.set_input(batches[i % 2]); model.optimize_parameters(3 + i)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = []
for e in ka:
    dev_us = getattr(e, 'device_time_total', None)
    if dev_us is None:
        dev_us = getattr(e, 'cuda_time_total', 0)
    rows.append((e.key, e.count / STEPS, dev_us / STEPS, e.self_cpu_time_total / STEPS))
print('%-70s %10s %12s %12s' % ('op / kernel', 'calls/step', 'device us', 'self cpu us'))
for k, c, d, s in sorted(rows, key=lambda r: -r[1])[:60]:
    print('%-70s %10.1f %12.1f %12.1f' % (k[:70], c, d, s))
