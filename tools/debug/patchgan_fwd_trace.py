"""The PatchGAN forward of the headline workload (6 -> 128 -> 256 -> 512 -> 1024 -> 1, batch 16 @ 256 x 256) a few times, eagerly: a small
target for `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace` (the TCC counters over the whole bench command take ~10 minutes per pass).
Three of its launches are the wide 128 x 128 x 32 forward tile (conv_fwd32d), the step's dominant kernel family."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cat_amd import _lib as L, networks, ops, synthetic
L.load()
opt = synthetic.default_options(norm='batch', track=True, ndf=128)
D = networks.define_D(6, 128, 'n_layers', 3, 'batch', 'normal', 0.02, [0], opt=opt).cuda().train()
x = ops.to_nhwc(synthetic.images((16, 6, 256, 256), 3).cuda())
with torch.no_grad():
    for _ in range(int(os.environ.get('REPS', '4'))):
        y = D(x)
torch.cuda.synchronize()
print('done', tuple(y.shape))
