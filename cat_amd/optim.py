"""FusedAdam: torch.optim.Adam semantics (distillers/base_inception_distiller.py:205-214) over flat HBM buffers.

All parameters of a param group live in ONE flat fp32 buffer (each nn.Parameter becomes a strided view into it, conv
weights keep their channels_last physical layout), with matching flat gradient / exp_avg / exp_avg_sq buffers:
  * step()      = one cat_adam_step launch per group instead of ~470 per-tensor launches;
  * zero_grad() = one cat_fill launch;
  * wgrad kernels write (or accumulate) straight into the flat gradient buffer (`_cat_grad_view`), autograd never
    materialises or adds weight gradients;
  * the flat gradient buffer is the RCCL all-reduce bucket of cat_amd.parallel (no bucket copies).
It subclasses torch.optim.Optimizer only so that the reference's LambdaLR schedulers attach unchanged."""
import ctypes as C

import torch

from . import _lib as L

# The kernels update parameters (and BatchNorm running statistics) through raw pointers, so torch's `_version` counters never move.
# Caches derived from trainable weights (cat_amd/frozen.py folds an eval-mode block's weights once) key on this epoch instead:
# it advances with every optimizer step, eager or replayed.
_WEIGHTS_EPOCH = 0


def weights_epoch():
    return _WEIGHTS_EPOCH


def _bump_weights_epoch():
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = None          # per group: dict(p, g, m, v, n, step)
        self.grad_scale = 1.0      # applied to gradients inside the Adam kernel (DP averaging)

    # -- flattening ------------------------------------------------------------------------------------
    @staticmethod
    def _dense_1d(t):
        """The tensor's storage, in physical order, as a 1-D view (requires a dense, non-overlapping layout)."""
        n = t.numel()
        return torch.as_strided(t, (n,), (1,), t.storage_offset())

    def _flatten(self):
        flats = []
        for group in self.param_groups:
            params = [p for p in group['params']]
            if not params:
                flats.append(None)
                continue
            dev = params[0].device
            if dev.type != 'cuda':
                raise RuntimeError('FusedAdam runs on the GPU only (parameters are on %s)' % dev)
            from . import ops
            spans = []
            for p in params:
                if p.dtype != torch.float32 or p.device != dev:
                    raise RuntimeError('FusedAdam: parameters of a group must be fp32 on one device')
                if p.dim() == 4 and p.shape[1] > 1:
                    # conv weights live as [O][kh][kw][round_up(I,4)] (zero padded): every filter quad is an aligned float4
                    if ops.weight_wcs(p) != ops.cs_for(p.shape[1]):
                        new = ops.padded_weight_like(p.shape, dev)
                        new.copy_(p.data)
                        p.data = new
                    spans.append(p.shape[0] * p.shape[2] * p.shape[3] * ops.cs_for(p.shape[1]))
                else:
                    if not p.is_contiguous():
                        p.data = p.data.contiguous()
                    spans.append(p.numel())
            # 16-byte align every segment so float4 kernels can address parameters directly
            offs, total = [], 0
            for span in spans:
                offs.append(total)
                total += (span + 3) // 4 * 4
            fp = torch.zeros(total, device=dev, dtype=torch.float32)
            fg = torch.zeros(total, device=dev, dtype=torch.float32)
            for p, off, span in zip(params, offs, spans):
                fp[off:off + span].copy_(torch.as_strided(p.data, (span,), (1,), p.storage_offset()))
                shape, stride = tuple(p.shape), tuple(p.stride())
                p.data = torch.as_strided(fp, shape, stride, off)
                gv = torch.as_strided(fg, shape, stride, off)
                p._cat_grad_view = gv
                p._cat_grad_state = {'fresh': True}
                p.grad = gv
            flats.append(dict(p=fp, g=fg, m=torch.zeros_like(fp), v=torch.zeros_like(fp), n=total, step=0, params=params,
                              offs=offs, hyper=torch.zeros(8, device=dev, dtype=torch.float32), hyper_host=None))
        self._flat = flats

    def _ensure_flat(self):
        if self._flat is None:
            self._flatten()
        return self._flat

    def flat_grads(self):
        """The gradient buckets (one per non-empty param group) -- what data parallelism all-reduces."""
        return [f['g'] for f in self._ensure_flat() if f is not None]

    # -- torch.optim.Optimizer API ---------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for f in self._ensure_flat():
            if f is None:
                continue
            L.call('cat_fill', C.c_void_p(f['g'].data_ptr()), f['n'], 0.0, stream)
            for p in f['params']:
                p._cat_grad_state['fresh'] = True
                if p.grad is None or p.grad.data_ptr() != p._cat_grad_view.data_ptr():
                    p.grad = p._cat_grad_view

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not used by the distillers')
        _bump_weights_epoch()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for group, f in zip(self.param_groups, self._ensure_flat()):
            if f is None:
                continue
            b1, b2 = group['betas']
            # optimiser scalars live in HBM (hipGraph-replayable): the host only rewrites them when a value changed (LambdaLR)
            host = (float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']), float(f['step']))
            if f['hyper_host'] != host:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('FusedAdam: hyper-parameters changed inside a graph capture; run one eager step first')
                f['hyper'].copy_(torch.tensor(list(host) + [0.0, 0.0], dtype=torch.float32), non_blocking=False)
            if not torch.cuda.is_current_stream_capturing():     # a capture records the launch, it does not run it
                f['step'] += 1
                f['hyper_host'] = host[:5] + (float(f['step']),)
            L.call('cat_adam_step_dev', C.c_void_p(f['p'].data_ptr()), C.c_void_p(f['g'].data_ptr()), C.c_void_p(f['m'].data_ptr()),
                   C.c_void_p(f['v'].data_ptr()), f['n'], C.c_void_p(f['hyper'].data_ptr()), float(self.grad_scale), stream)

    def sync_hyper_for_replay(self):
        """Before replaying a captured step: if a scheduler (LambdaLR linear decay) or the user changed lr / betas / eps / weight_decay
        since the capture, rewrite those device-resident scalars (the step counter and the bias corrections stay with the device)."""
        for group, f in zip(self.param_groups, self._ensure_flat()):
            if f is None or f['hyper_host'] is None:
                continue
            b1, b2 = group['betas']
            host5 = (float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']))
            if f['hyper_host'][:5] != host5:
                f['hyper'][:5].copy_(torch.tensor(host5, dtype=torch.float32))
                f['hyper_host'] = host5 + f['hyper_host'][5:]

    def note_graph_replay(self):
        """A captured step was replayed: the device-side step counter advanced, keep the host mirror in sync."""
        _bump_weights_epoch()
        for f in self._ensure_flat():
            if f is not None:
                f['step'] += 1
                if f['hyper_host'] is not None:
                    f['hyper_host'] = f['hyper_host'][:5] + (float(f['step']),)

    # -- checkpoint interchange (torch.optim.Adam state_dict layout) ----------------------------------------
    def state_dict(self):
        state, idx = {}, 0
        groups = []
        for group, f in zip(self.param_groups, self._flat or [None] * len(self.param_groups)):
            ids = []
            for j, p in enumerate(group['params']):
                if f is not None and f['step'] > 0:
                    off, n = f['offs'][j], p.numel()
                    view = lambda buf: torch.as_strided(buf, tuple(p.shape), tuple(p.stride()), off).clone()
                    state[idx] = {'step': torch.tensor(float(f['step'])), 'exp_avg': view(f['m']), 'exp_avg_sq': view(f['v'])}
                ids.append(idx)
                idx += 1
            g = {k: v for k, v in group.items() if k != 'params'}
            g['params'] = ids
            groups.append(g)
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        flats = self._ensure_flat()
        idx = 0
        for group, f, saved in zip(self.param_groups, flats, sd['param_groups']):
            for k, v in saved.items():
                if k != 'params':
                    group[k] = v
            for j, p in enumerate(group['params']):
                st = sd['state'].get(idx)
                if st is not None and f is not None:
                    off = f['offs'][j]
                    torch.as_strided(f['m'], tuple(p.shape), tuple(p.stride()), off).copy_(st['exp_avg'])
                    torch.as_strided(f['v'], tuple(p.shape), tuple(p.stride()), off).copy_(st['exp_avg_sq'])
                    f['step'] = int(st['step'])
                idx += 1
