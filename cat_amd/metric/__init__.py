"""Evaluation metrics of the distillers (reference metric/__init__.py): FID with its InceptionV3 feature extractor on the HIP kernels."""
import numpy as np
import torch

from .fid_score import _compute_statistics_of_ims, calculate_frechet_distance, get_activations_from_ims  # noqa: F401
from .inception import InceptionV3  # noqa: F401


def tensor2im_batch(t):
    """util.tensor2im on a [N, 3, H, W] batch in [-1, 1] (utils/util.py:58-88): [N, H, W, 3] uint8, truncating cast."""
    a = t.detach().cpu().float().numpy()
    return np.clip((np.transpose(a, (0, 2, 3, 1)) + 1) / 2.0 * 255.0, 0, 255).astype(np.uint8)


def get_fid(fakes, model, npz, device=None, batch_size=1, use_tqdm=True):
    """metric/__init__.py:11-21: `fakes` = list of [B, 3, H, W] tensors in [-1, 1]; npz = {'mu', 'sigma'} of the real set."""
    m1, s1 = npz['mu'], npz['sigma']
    ims = tensor2im_batch(torch.cat(fakes, dim=0)).astype(float)
    m2, s2 = _compute_statistics_of_ims(ims, model, batch_size, 2048, device, use_tqdm=use_tqdm)
    return float(calculate_frechet_distance(m1, s1, m2, s2))
