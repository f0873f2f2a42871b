"""ORACLE-side helper (test infrastructure): deterministic, platform-independent tensor fills.

numpy's PCG64 + ziggurat normals are bit-reproducible across machines, unlike torch's vectorised normal_(), so the
golden-vector generator (tools/make_golden.py, run where /root/reference exists) and the tests (run anywhere) can
rebuild identical weights / inputs from a seed instead of shipping megabytes of parameters."""
import numpy as np
import torch


def _rng(seed):
    return np.random.default_rng(seed)


def normal(shape, seed, scale=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy((_rng(seed).standard_normal(n, dtype=np.float32) * np.float32(scale)).reshape(shape))


def images(shape, seed):
    """N(0,1).tanh() images in (-1,1), SURVEY §8(d) 'Inputs'."""
    return torch.tanh(normal(shape, seed))


def fill_state_dict(sd, seed, gamma_abs_normal=False):
    """Overwrite every tensor of a state_dict in key order.
    conv / conv-transpose weights ~ N(0, 1/fan_in); biases ~ 0.1 N(0,1); norm weight ~ 1 + 0.2 N(0,1)
    (or |N(0,1)| for the canonical teacher); running_mean ~ 0.1 N; running_var ~ U(0.5, 1.5)."""
    rng = _rng(seed)
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        n = int(np.prod(shape)) if len(shape) else 1
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros_like(v)
            continue
        z = rng.standard_normal(n, dtype=np.float32)
        if k.endswith('running_var'):
            t = 0.5 + rng.random(n, dtype=np.float32)
        elif k.endswith('running_mean'):
            t = 0.1 * z
        elif v.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = z / np.float32(np.sqrt(fan_in))
        elif k.endswith('.weight'):
            t = np.abs(z) if gamma_abs_normal else 1.0 + 0.2 * z
        else:
            t = 0.1 * z
        out[k] = torch.from_numpy(np.asarray(t, dtype=np.float32).reshape(shape)).clone()
    return out
