"""One eager C2 step at a ragged plane size (SZ, default 232: 58 x 58 trunk, 116 x 116 coarse PatchGAN grid); dumps the flat gradient
buffers of D and of the student after the step to gpurun_out/ragged_<TAG>.pt.  Run twice with different kernel switches
(CAT_TWGRAD / CAT_SMALLCI_DGRAD / CAT_NORM_WALK = 0 / 1) and compare with --compare A B: the first backward involves no optimizer
amplification, so the two paths must agree to fp32 summation-order noise."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
if len(sys.argv) > 1 and sys.argv[1] == '--compare':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].double() - b[k].double()).abs().max().item()
        print(f'{k:12s} n={a[k].numel():9d}  max|a|={a[k].abs().max().item():.3e}  max|a-b|={d:.3e}  rel={d / max(a[k].abs().max().item(), 1e-30):.2e}')
    sys.exit(0)
import bench
from cat_amd import synthetic, ops
ops.set_tconv_min_tiles(1)
args = argparse.Namespace(workload='c2', batch=3, size=int(os.environ.get('SZ', '232')), target_flops=4.6e9)
model, opt = bench.build_model(args, 0)
b = {'A': synthetic.images((3, 3, args.size, args.size), 10).cuda(), 'B': synthetic.images((3, 3, args.size, args.size), 20).cuda(), 'A_paths': [], 'B_paths': []}
model.set_input(b); model.optimize_parameters(0)
torch.cuda.synchronize()
out = {'gD': model.optimizer_D.flat_grads()[0].cpu(), 'gG': model.optimizer_G.flat_grads()[0].cpu(), 'Sfake': model.Sfake_B.detach().float().cpu().contiguous()}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
torch.save(out, os.path.join(ROOT, 'gpurun_out', 'ragged_%s.pt' % os.environ.get('TAG', 'x')))
print('saved', {k: float(v.double().abs().sum()) for k, v in out.items()})
