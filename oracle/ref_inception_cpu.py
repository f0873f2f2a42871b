"""ORACLE (test infrastructure, never on the product path): CPU restatement of the FID feature extractor the reference evaluates students with
-- `InceptionV3([3])` of metric/inception.py:16-150 (input resize :129-133, `2 * x - 1` :135-136, block grouping :72-108), its patched blocks
FIDInceptionA :177-200, FIDInceptionC :203-233, FIDInceptionE_1 :236-268, FIDInceptionE_2 :271-300, and `get_activations_from_ims` of
metric/fid_score.py:152-216 -- as plain functions over a state_dict with torchvision's key names (`Mixed_6b.branch7x7_2.conv.weight` ...).

Third-party part: the blocks' CONSTRUCTORS and the un-patched InceptionB / InceptionD / BasicConv2d forwards live in torchvision 0.8.2
(requirements.txt:97-104), which is neither under /root/reference nor installed here; their published architecture (Szegedy et al. 2015,
layer names as in the pt_inception-2015-12-05 checkpoint) is restated below.

PINNED: tests/golden/inception_fid.npz was written by tools/make_golden_inception.py, which imports the reference's own metric/inception.py +
metric/fid_score.py (over tools/tv_inception_stub.py for the absent torchvision classes) and records its pool3 features and block outputs for
seeded weights; tests/test_metric_inception.py holds this file to them.  What that pin cannot see is stated there: a disagreement between the
stub and real torchvision constructors (both restate the same published shapes)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import detfill

BN_EPS = 1e-3      # torchvision BasicConv2d: nn.BatchNorm2d(out_channels, eps=0.001)


def basic_conv(sd, p, x, stride=1, padding=0):
    """BasicConv2d.forward: conv (no bias) -> BatchNorm2d(eps 1e-3, eval) -> ReLU."""
    y = F.conv2d(x, sd[p + '.conv.weight'], None, stride=stride, padding=padding)
    y = F.batch_norm(y, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'], sd[p + '.bn.bias'], False, 0.0, BN_EPS)
    return F.relu(y)


def _avg3(x):
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=False)      # the FID patch (metric/inception.py:192-196)


def inception_a(sd, p, x):
    b1 = basic_conv(sd, p + '.branch1x1', x)
    b5 = basic_conv(sd, p + '.branch5x5_2', basic_conv(sd, p + '.branch5x5_1', x), padding=2)
    b3 = basic_conv(sd, p + '.branch3x3dbl_1', x)
    b3 = basic_conv(sd, p + '.branch3x3dbl_3', basic_conv(sd, p + '.branch3x3dbl_2', b3, padding=1), padding=1)
    bp = basic_conv(sd, p + '.branch_pool', _avg3(x))
    return torch.cat([b1, b5, b3, bp], 1)


def inception_b(sd, p, x):
    b3 = basic_conv(sd, p + '.branch3x3', x, stride=2)
    bd = basic_conv(sd, p + '.branch3x3dbl_2', basic_conv(sd, p + '.branch3x3dbl_1', x), padding=1)
    bd = basic_conv(sd, p + '.branch3x3dbl_3', bd, stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def inception_c(sd, p, x):
    b1 = basic_conv(sd, p + '.branch1x1', x)
    b7 = basic_conv(sd, p + '.branch7x7_1', x)
    b7 = basic_conv(sd, p + '.branch7x7_2', b7, padding=(0, 3))
    b7 = basic_conv(sd, p + '.branch7x7_3', b7, padding=(3, 0))
    bd = basic_conv(sd, p + '.branch7x7dbl_1', x)
    bd = basic_conv(sd, p + '.branch7x7dbl_2', bd, padding=(3, 0))
    bd = basic_conv(sd, p + '.branch7x7dbl_3', bd, padding=(0, 3))
    bd = basic_conv(sd, p + '.branch7x7dbl_4', bd, padding=(3, 0))
    bd = basic_conv(sd, p + '.branch7x7dbl_5', bd, padding=(0, 3))
    bp = basic_conv(sd, p + '.branch_pool', _avg3(x))
    return torch.cat([b1, b7, bd, bp], 1)


def inception_d(sd, p, x):
    b3 = basic_conv(sd, p + '.branch3x3_2', basic_conv(sd, p + '.branch3x3_1', x), stride=2)
    b7 = basic_conv(sd, p + '.branch7x7x3_1', x)
    b7 = basic_conv(sd, p + '.branch7x7x3_2', b7, padding=(0, 3))
    b7 = basic_conv(sd, p + '.branch7x7x3_3', b7, padding=(3, 0))
    b7 = basic_conv(sd, p + '.branch7x7x3_4', b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def inception_e(sd, p, x, pool):
    """pool: 'avg' = FIDInceptionE_1 (avg pool without the padding in the divisor), 'max' = FIDInceptionE_2 (:292: max_pool2d 3 / 1 / 1)."""
    b1 = basic_conv(sd, p + '.branch1x1', x)
    t = basic_conv(sd, p + '.branch3x3_1', x)
    b3 = torch.cat([basic_conv(sd, p + '.branch3x3_2a', t, padding=(0, 1)), basic_conv(sd, p + '.branch3x3_2b', t, padding=(1, 0))], 1)
    t = basic_conv(sd, p + '.branch3x3dbl_2', basic_conv(sd, p + '.branch3x3dbl_1', x), padding=1)
    bd = torch.cat([basic_conv(sd, p + '.branch3x3dbl_3a', t, padding=(0, 1)), basic_conv(sd, p + '.branch3x3dbl_3b', t, padding=(1, 0))], 1)
    px = _avg3(x) if pool == 'avg' else F.max_pool2d(x, kernel_size=3, stride=1, padding=1)
    bp = basic_conv(sd, p + '.branch_pool', px)
    return torch.cat([b1, b3, bd, bp], 1)


def inception_v3_blocks(sd, inp, resize_input=True, normalize_input=True):
    """InceptionV3.forward with output_blocks = [0, 1, 2, 3]: the four block outputs (metric/inception.py:110-150)."""
    x = inp
    if resize_input:
        x = F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=False)
    if normalize_input:
        x = 2 * x - 1
    outs = []
    x = basic_conv(sd, 'Conv2d_1a_3x3', x, stride=2)
    x = basic_conv(sd, 'Conv2d_2a_3x3', x)
    x = basic_conv(sd, 'Conv2d_2b_3x3', x, padding=1)
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    outs.append(x)
    x = basic_conv(sd, 'Conv2d_3b_1x1', x)
    x = basic_conv(sd, 'Conv2d_4a_3x3', x)
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    outs.append(x)
    for name in ('Mixed_5b', 'Mixed_5c', 'Mixed_5d'):
        x = inception_a(sd, name, x)
    x = inception_b(sd, 'Mixed_6a', x)
    for name in ('Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e'):
        x = inception_c(sd, name, x)
    outs.append(x)
    x = inception_d(sd, 'Mixed_7a', x)
    x = inception_e(sd, 'Mixed_7b', x, 'avg')
    x = inception_e(sd, 'Mixed_7c', x, 'max')
    outs.append(F.adaptive_avg_pool2d(x, (1, 1)))
    return outs


def get_activations_from_ims(ims, sd, batch_size=50, dims=2048):
    """metric/fid_score.py:152-216: `ims` float array [N, H, W, 3] (or [N, 3, H, W]) in [0, 255] -> [N, dims] float64 pool3 features."""
    ims = np.array(ims, dtype=np.float64)
    out = np.empty((len(ims), dims))
    for s in range(0, len(ims), batch_size):
        images = ims[s:s + batch_size]
        if images.shape[1] != 3:
            images = images.transpose((0, 3, 1, 2))
        batch = torch.from_numpy(images / 255).float()
        with torch.no_grad():
            pred = inception_v3_blocks(sd, batch)[3]
        out[s:s + len(images)] = pred.numpy().reshape(len(images), -1)
    return out


def seeded_state_dict(shapes, seed):
    """Deterministic weights for the fixture: detfill's fill with He-scaled conv weights (x sqrt 2: the ReLU stack keeps its variance over ~45
    sequential layers) -- the real pt_inception weights cannot be downloaded here."""
    sd = detfill.fill_state_dict(shapes, seed)
    for k, v in sd.items():
        if v.dim() == 4:
            sd[k] = v * np.float32(np.sqrt(2.0))
    return sd
