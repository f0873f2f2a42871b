"""tests/golden/inception_fid.npz: pool3 features of the REFERENCE's own FID feature extractor on seeded weights and inputs.

Runs only in the build container (imports /root/reference).  What executes is the reference's metric/inception.py (InceptionV3 wrapper, its
FIDInceptionA / C / E_1 / E_2 forwards, fid_inception_v3's patching) and metric/fid_score.py (get_activations_from_ims,
calculate_frechet_distance); torchvision 0.8.2 -- absent offline -- is replaced by tools/tv_inception_stub.py (constructors of the published
architecture + the un-patched InceptionB / D forwards), and `load_state_dict_from_url` by a seeded state_dict (oracle/ref_inception_cpu.
seeded_state_dict): the pt_inception-2015-12-05 checkpoint cannot be downloaded here.

    python tools/make_golden_inception.py        # rewrites tests/golden/inception_fid.npz"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_import  # noqa: E402
import tv_inception_stub as tvi  # noqa: E402
from oracle import detfill, ref_inception_cpu as RI  # noqa: E402

SEED_W, SEED_IMS = 901, 902


def main():
    ref_import.install()
    holder = {}
    # the reference does `from torchvision import models` / `models.inception.InceptionA` / `from torchvision.models.utils import
    # load_state_dict_from_url`: point those names at the restated architecture and at the seeded weights
    tvm = sys.modules['torchvision.models']
    tvm.inception_v3 = tvi.inception_v3
    tvm.inception = tvi
    sys.modules['torchvision.models.inception'] = tvi
    utils = types.ModuleType('torchvision.models.utils')
    utils.load_state_dict_from_url = lambda url, progress=True: holder['sd']
    sys.modules['torchvision.models.utils'] = utils
    tvm.utils = utils
    for name in [m for m in sys.modules if m == 'metric' or m.startswith('metric.')]:
        del sys.modules[name]
    import importlib
    inc = importlib.import_module('metric.inception')
    fid = importlib.import_module('metric.fid_score')
    assert inc.__file__.startswith('/root/reference/'), inc.__file__

    shapes = {k: torch.zeros(v.shape, dtype=v.dtype) for k, v in tvi.inception_v3(num_classes=1008, aux_logits=False).state_dict().items()}
    holder['sd'] = RI.seeded_state_dict(shapes, SEED_W)
    torch.manual_seed(0)
    model = inc.InceptionV3([0, 1, 2, 3])          # all four block outputs for localisation; [3] is what get_fid uses
    model.eval()
    pool3 = inc.InceptionV3([3])
    pool3.eval()

    # images as get_fid hands them over: util.tensor2im(...).astype(float), HWC in [0, 255]; two sizes (resized to 299 x 299 inside)
    ims = np.floor((detfill.images((3, 40, 56, 3), SEED_IMS).numpy().astype(np.float64) + 1) / 2 * 255)
    acts = fid.get_activations_from_ims(ims.copy(), pool3, batch_size=2, dims=2048, device=torch.device('cpu'), use_tqdm=False)
    with torch.no_grad():
        x = torch.from_numpy(ims.transpose(0, 3, 1, 2) / 255).float()
        blocks = model(x)
    assert np.allclose(blocks[3].numpy().reshape(3, -1), acts, rtol=0, atol=1e-6)

    # the oracle restatement against the reference's run, right here
    mine = RI.get_activations_from_ims(ims.copy(), holder['sd'], batch_size=2)
    err = np.abs(mine - acts).max() / np.abs(acts).max()
    print('oracle vs reference pool3: rel err %.2e, |features| max %.3e' % (err, np.abs(acts).max()))
    assert err < 1e-5
    ob = RI.inception_v3_blocks(holder['sd'], x)
    for i, (a, b) in enumerate(zip(ob, blocks)):
        print('block %d' % i, tuple(b.shape), 'rel err %.2e' % (float((a - b).abs().max()) / float(b.abs().max())))

    # Frechet distance of two feature sets through the reference's function (numpy / scipy): pins cat_amd.metric.fid_score's copy of the formula
    rng = np.random.default_rng(7)
    f1, f2 = rng.standard_normal((40, 16)), rng.standard_normal((40, 16)) * 1.3 + 0.2
    fd = fid.calculate_frechet_distance(f1.mean(0), np.cov(f1, rowvar=False), f2.mean(0), np.cov(f2, rowvar=False))

    def sub(t, cmax=6, step=3):
        return t[:, :cmax, ::step, ::step].contiguous().numpy()
    out = dict(shapes=json.dumps([[k, list(v.shape)] for k, v in shapes.items()]), seed_w=SEED_W, seed_ims=SEED_IMS, ims=ims.astype(np.float32),
               pool3=acts, block0=sub(blocks[0]), block1=sub(blocks[1]), block2=sub(blocks[2], 6, 2),
               block_checks=np.array([[float(b.double().sum()), float(b.double().abs().sum()), float((b.double() ** 2).sum())] for b in blocks]),
               wrapper_keys=json.dumps(list(model.state_dict().keys())[:12] + list(model.state_dict().keys())[-6:]),
               n_wrapper_keys=len(model.state_dict()), fd_f1=f1, fd_f2=f2, fd=float(fd))
    path = os.path.join(ROOT, 'tests', 'golden', 'inception_fid.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
