"""FID from pool3 features.  The feature half runs on the HIP kernels (cat_amd.metric.inception.InceptionV3); the
Frechet distance is host arithmetic on 2048 x 2048 matrices, as in the reference (metric/fid_score.py:217-275: numpy + scipy.linalg.sqrtm)."""
import numpy as np
import torch


def get_activations_from_ims(ims, model, batch_size=50, dims=2048, device=None, verbose=False, use_tqdm=True):
    """metric/fid_score.py:152-216.  ims: float array [N, H, W, 3] (or [N, 3, H, W]) in [0, 255] -- it is scaled IN PLACE like the reference
    does (`images /= 255` on a view); returns [N, dims] float64 features.  Same batching (the last batch may be short)."""
    model.eval()
    device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    n_batches = (len(ims) + batch_size - 1) // batch_size
    pred_arr = np.empty((len(ims), dims))
    it = range(n_batches)
    if use_tqdm:
        try:
            from tqdm import tqdm
            it = tqdm(it)
        except ImportError:
            pass
    for i in it:
        start, end = i * batch_size, min((i + 1) * batch_size, len(ims))
        images = ims[start:end]
        if images.shape[1] != 3:
            images = images.transpose((0, 3, 1, 2))
        images /= 255
        batch = torch.from_numpy(images).type(torch.FloatTensor).to(device)
        with torch.no_grad():
            pred = model(batch)[0]
        if pred.shape[2] != 1 or pred.shape[3] != 1:      # a block below pool3 was selected: adaptive_avg_pool2d(pred, (1, 1))
            from .inception import GlobalAvgPool
            pred = GlobalAvgPool()(pred)
        pred_arr[start:end] = pred.cpu().data.numpy().reshape(end - start, -1)
    if verbose:
        print(' done')
    return pred_arr


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1 - mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)) (metric/fid_score.py:217-275): matrix square root by scipy, the near-singular retry
    with eps on the diagonals, imaginary parts below 1e-3 dropped."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape, 'Training and test mean vectors have different lengths'
    assert sigma1.shape == sigma2.shape, 'Training and test covariances have different dimensions'
    diff = mu1 - mu2
    prod = sigma1.dot(sigma2)
    ok = True
    for _ in range(30):
        ok = True
        covmean, _ = linalg.sqrtm(prod, disp=False)
        if not np.isfinite(covmean).all():
            print('fid calculation produces singular product; adding %s to diagonal of cov estimates' % eps)
            offset = np.eye(sigma1.shape[0]) * eps
            covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
        if np.iscomplexobj(covmean):
            if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
                ok = False
            covmean = covmean.real
        if ok:
            break
    if not ok:
        print('Warning: the fid may be incorrect!')
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def _compute_statistics_of_ims(ims, model, batch_size, dims, device, use_tqdm=True):
    act = get_activations_from_ims(ims, model, batch_size, dims, device, verbose=False, use_tqdm=use_tqdm)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)
