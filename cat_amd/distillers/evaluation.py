"""`evaluate_model` on the GPU (reference distillers/inception_distiller.py:204-281,
distillers/spade_distiller.py:96-180).

The generator passes run on the HIP kernels (`model.test()` with the student in eval mode: BatchNorm students take the frozen-block
fusion of cat_amd/frozen.py); images of the first 10 samples (or all) are written like the reference does.

FID (round 5): the InceptionV3 pool3 feature extractor runs on the HIP kernels too (cat_amd.metric.InceptionV3, cat_amd.metric.get_fid --
reference metric/inception.py, metric/fid_score.py:152-216, metric/__init__.py:11-21).  `attach_fid(model, checkpoint, real_stat_path)`
builds it exactly as the reference's __init__ does (base_inception_distiller.py:218-234: `InceptionV3([block_idx])`, `np.load(real_stat_path)`)
from the torchvision-keyed FID checkpoint the reference downloads; a model that carries `inception_model` + `npz` needs no `fid_fn`.
The mIoU network (DRN + cityscapes data) stays with the reference: the integrator attaches

    model.miou_fn = lambda fakes, names: get_mIoU(fakes, names, drn_model, model.device, table_path=..., data_dir=..., ...)

(`fakes` is the reference's list of NCHW CPU tensors, one per eval batch) and gets the reference's bookkeeping back: `is_best`,
`best_fid / best_mIoU`, the 3-evaluation running means and the `metric/*` dict that `Trainer` logs."""
import ntpath
import os

import numpy as np
import torch


def tensor2im(t):
    """[-1, 1] CHW tensor -> HWC uint8 (reference utils/util.py:58-88; truncating cast, like numpy's astype)."""
    a = t.detach().cpu().float().numpy()
    if a.ndim == 2:
        a = a[None]
    a = np.clip((np.transpose(a, (1, 2, 0)) + 1) / 2.0 * 255.0, 0, 255)
    if a.shape[2] == 1:
        a = a[:, :, 0]
    return a.astype(np.uint8)


def label_colormap(n):
    """The bit-interleaved colour table the reference uses for n != 35 labels (utils/util.py:176-188): label i -> id = i + 1, the
    low three bits of successive octal digits of id go to the high bits of r, g, b."""
    cmap = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        ident, r, g, b = i + 1, 0, 0, 0
        for j in range(7):
            r ^= (ident & 1) << (7 - j)
            g ^= ((ident >> 1) & 1) << (7 - j)
            b ^= ((ident >> 2) & 1) << (7 - j)
            ident >>= 3
        cmap[i] = (r, g, b)
    return cmap


def tensor2label(t, n_label):
    """One-hot (or index) CHW label tensor -> colour image (reference utils/util.py:90-115)."""
    t = t.detach().cpu().float()
    idx = t.max(0)[1] if t.shape[0] > 1 else t[0].long()
    cmap = label_colormap(n_label)
    out = np.zeros(tuple(idx.shape) + (3,), dtype=np.uint8)
    valid = (idx >= 0) & (idx < n_label)
    out[valid.numpy()] = cmap[idx[valid].numpy()]
    return out


def save_image(arr, path):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if arr.ndim == 2:
        arr = np.repeat(arr[:, :, None], 3, 2)
    Image.fromarray(arr).save(path.replace('.jpg', '.png'))


def _track(model, value, best_attr, hist_attr, better):
    """best value + is_best + running mean of the last 3 evaluations (inception_distiller.py:251-256, 271-276)."""
    if better(value, getattr(model, best_attr)):
        model.is_best = True
        setattr(model, best_attr, value)
    hist = getattr(model, hist_attr)
    hist.append(value)
    if len(hist) > 3:
        hist.pop(0)
    return sum(hist) / len(hist)


def attach_fid(model, state_dict, real_stat_path=None, npz=None, dims=2048):
    """What the reference's distiller __init__ does for FID (base_inception_distiller.py:218-234), with the feature extractor on the HIP kernels:
    `state_dict` = the torchvision-keyed FID checkpoint (pt_inception-2015-12-05-6726825d.pth, or a path to it); `real_stat_path` = the
    dataset's precomputed {'mu', 'sigma'} file."""
    from ..metric import InceptionV3
    if dims != 2048:      # metric.get_fid compares pool3 features with the dataset's 2048-wide {mu, sigma} (metric/__init__.py:11-21)
        raise ValueError('attach_fid: the FID path uses the 2048-wide pool3 features (dims=%r)' % (dims,))
    if isinstance(state_dict, (str, bytes, os.PathLike)):
        state_dict = torch.load(state_dict, map_location='cpu')
    net = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]])
    net.load_fid_state_dict(state_dict)
    model.inception_model = net.to(model.device).eval()
    model.npz = npz if npz is not None else np.load(real_stat_path)
    return net


def evaluate(model, step, student, feed, images, want_fid, want_miou, save_all=False):
    """feed(batch): set_input / set_single_input; images(j): {'input'|'real'|'Tfake'|'Sfake': HWC uint8} of sample j of the batch."""
    if getattr(model, 'eval_dataloader', None) is None:
        raise RuntimeError('evaluate_model: attach model.eval_dataloader (the reference builds it from --dataroot in __init__; '
                           'cat_amd does not own datasets)')
    if want_fid and getattr(model, 'fid_fn', None) is None:
        if getattr(model, 'inception_model', None) is not None and getattr(model, 'npz', None) is not None:
            from .. import metric      # the reference's own call: get_fid(fakes, self.inception_model, self.npz, ...) (inception_distiller.py:246-249)
            model.fid_fn = lambda fakes: metric.get_fid(fakes, model.inception_model, model.npz, device=model.device,
                                                        batch_size=getattr(model.opt, 'eval_batch_size', 1), use_tqdm=False)
        else:
            raise RuntimeError('evaluate_model: call evaluation.attach_fid(model, fid_checkpoint, real_stat_path) or attach model.fid_fn '
                               '(cat_amd/distillers/evaluation.py)')
    if want_miou and getattr(model, 'miou_fn', None) is None:
        raise RuntimeError('evaluate_model: attach model.miou_fn = lambda fakes, names: get_mIoU(...) (cat_amd/distillers/evaluation.py)')
    model.is_best = False
    save_dir = os.path.join(model.opt.log_dir, 'eval', str(step))
    os.makedirs(save_dir, exist_ok=True)
    student.eval()
    fakes, names, cnt = [], [], 0
    try:
        for batch in model.eval_dataloader:
            feed(batch)
            model.test()
            fakes.append(model.Sfake_B.detach().cpu().contiguous())
            for j in range(len(model.image_paths)):
                name = os.path.splitext(ntpath.basename(model.image_paths[j]))[0]
                names.append(name)
                if cnt < 10 or save_all:
                    for kind, arr in images(j).items():
                        save_image(arr, os.path.join(save_dir, kind, '%s.png' % name))
                cnt += 1
    finally:
        student.train()
    ret = {}
    if want_fid:
        fid = float(model.fid_fn(fakes))
        ret['metric/fid'] = fid
        ret['metric/fid-mean'] = _track(model, fid, 'best_fid', 'fids', lambda a, b: a < b)
        ret['metric/fid-best'] = model.best_fid
    if want_miou:
        miou = float(model.miou_fn(fakes, names))
        ret['metric/mIoU'] = miou
        ret['metric/mIoU-mean'] = _track(model, miou, 'best_mIoU', 'mIoUs', lambda a, b: a > b)
        ret['metric/mIoU-best'] = model.best_mIoU
    return ret
