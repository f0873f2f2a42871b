"""The FID feature extractor on the HIP kernels.

Reference: `InceptionV3` of metric/inception.py:16-150 -- torchvision's Inception3 with the FID patches (FIDInceptionA :177-200, FIDInceptionC
:203-233, FIDInceptionE_1 :236-268, FIDInceptionE_2 :271-300), grouped into four output blocks, bilinear resize to 299 x 299 and `2 * x - 1`
in front -- built by the distillers at base_inception_distiller.py:218-224 and called from metric/fid_score.py:152-216.

Inference only (the reference never trains it): every BasicConv2d (conv without bias -> BatchNorm2d(eps 1e-3) -> ReLU) is ONE implicit-GEMM
launch with the running statistics folded into the filters and the ReLU in the epilogue; the 1x7 / 7x1 / 1x3 / 3x1 factorised filters go through
cat_conv2d_fwd_rect (separate padding along H and W); every branch writes its channel slice of the block's concatenated NHWC output directly
(`torch.cat(outputs, 1)` is never materialised); pools, the global average and the input resize + normalisation are csrc/eval_ops.hip.
Module / parameter names are the reference wrapper's (`blocks.2.4.branch7x7_2.conv.weight` ...); `load_fid_state_dict` takes the
torchvision-keyed checkpoint the reference downloads (pt_inception-2015-12-05-6726825d.pth; there is no network here: tests use seeded weights)."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops

POOL_MAX, POOL_AVG_EXCL = 0, 1


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class BasicConv2d(nn.Module):
    """conv (no bias) -> BatchNorm2d(eps=0.001) -> ReLU, sub-modules `conv` / `bn` as in torchvision (parameter holders: never called)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)
        self.out_channels = out_channels

    def _folded(self):
        """(filters in the kernels' padded channels-last storage with the BatchNorm scale folded in, bias): refreshed when a tensor changes."""
        ts = (self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in ts)
        cache = self.__dict__.get('_cat_fold')
        if cache is None or cache[0] != key:
            with torch.no_grad():
                scale = self.bn.weight * torch.rsqrt(self.bn.running_var + self.bn.eps)
                w = ops.padded_weight_like(self.conv.weight.shape, self.conv.weight.device)
                w.copy_(self.conv.weight * scale.view(-1, 1, 1, 1))
                bias = (self.bn.bias - self.bn.running_mean * scale).contiguous()
            cache = self.__dict__['_cat_fold'] = (key, w, ops.weight_wcs(w), bias)
        return cache[1:]

    def forward(self, x, out=None, c0=0):
        """x: NHWC activation.  out / c0: write channels [c0, c0 + Cout) of this wider activation instead of a fresh tensor."""
        if self.training or torch.is_grad_enabled():
            raise NotImplementedError('the FID InceptionV3 runs in eval mode under no_grad (metric/fid_score.py:183,203)')
        w, wcs, bias = self._folded()
        n, c, h, wd = x.shape
        (kh, kw), (sh, sw), (ph, pw) = _pair(self.conv.kernel_size), _pair(self.conv.stride), _pair(self.conv.padding)
        if sh != sw or c != self.conv.in_channels:
            raise ValueError('BasicConv2d: geometry')
        ho, wo = (h + 2 * ph - kh) // sh + 1, (wd + 2 * pw - kw) // sw + 1
        cout = self.out_channels
        if out is None:
            out, c0 = ops.empty_act(n, cout, ho, wo, x.device), 0
        elif tuple(out.shape[2:]) != (ho, wo) or c0 % 4 or cout % 4 or c0 + cout > out.shape[1]:
            raise ValueError('BasicConv2d: output slice')
        g = ops._conv_geom(n, h, wd, c, ops.act_cs(x), ho, wo, cout, ops.act_cs(out), kh, kw, sh, ph, L.PAD_ZERO, L.ACT_RELU, 0.0, ycw=cout, wcs=wcs)
        L.call('cat_conv2d_fwd_rect', C.byref(g), pw, ops._p(x), ops._p(w), ops._p(bias), C.c_void_p(out.data_ptr() + 4 * c0), ops._stream())
        return out


def pool2d(x, k, stride, pad, mode, out=None, c0=0):
    n, c, h, w = x.shape
    if c % 4:
        raise ValueError('pool2d: channel count must be a multiple of 4')
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    if out is None:
        out, c0 = ops.empty_act(n, c, ho, wo, x.device), 0
    elif tuple(out.shape[2:]) != (ho, wo) or out.shape[0] != n or c0 % 4 or c0 + c > out.shape[1]:
        raise ValueError('pool2d: output slice')      # a wrong slice would overwrite neighbouring channels (the kernel only checks the stride)
    L.call('cat_pool2d_fwd', ops._p(x), ops.act_cs(x), n, h, w, c, k, stride, pad, mode, C.c_void_p(out.data_ptr() + 4 * c0), ops.act_cs(out), ho, wo,
           ops._stream())
    return out


class MaxPool3x3s2(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2) (metric/inception.py:78,86)."""

    def forward(self, x):
        return pool2d(x, 3, 2, 0, POOL_MAX)


class GlobalAvgPool(nn.Module):
    """nn.AdaptiveAvgPool2d((1, 1)) (metric/inception.py:106)."""

    def forward(self, x):
        n, c, h, w = x.shape
        y = torch.empty((n, c), device=x.device, dtype=torch.float32)
        L.call('cat_global_avgpool_fwd', ops._p(x), ops.act_cs(x), n, h * w, c, ops._p(y), c, ops._stream())
        return y.view(n, c, 1, 1)


class _Block(nn.Module):
    """Four (three) branches whose last layers write channel slices of one output tensor."""

    def _out(self, x, widths, stride=1):
        n, _, h, w = x.shape
        if stride == 2:
            h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        offs = [sum(widths[:i]) for i in range(len(widths))]
        return ops.empty_act(n, sum(widths), h, w, x.device), offs


class FIDInceptionA(_Block):
    def __init__(self, in_channels, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(in_channels, 64, 1)
        self.branch5x5_1 = BasicConv2d(in_channels, 48, 1)
        self.branch5x5_2 = BasicConv2d(48, 64, 5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, padding=1)
        self.branch_pool = BasicConv2d(in_channels, pool_features, 1)
        self.widths = [64, 64, 96, pool_features]

    def forward(self, x):
        y, o = self._out(x, self.widths)
        self.branch1x1(x, y, o[0])
        self.branch5x5_2(self.branch5x5_1(x), y, o[1])
        self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)), y, o[2])
        self.branch_pool(pool2d(x, 3, 1, 1, POOL_AVG_EXCL), y, o[3])      # the FID patch: padding excluded from the divisor (:192-196)
        return y


class InceptionB(_Block):
    def __init__(self, in_channels):
        super().__init__()
        self.branch3x3 = BasicConv2d(in_channels, 384, 3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, stride=2)
        self.widths = [384, 96, in_channels]

    def forward(self, x):
        y, o = self._out(x, self.widths, stride=2)
        self.branch3x3(x, y, o[0])
        self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)), y, o[1])
        pool2d(x, 3, 2, 0, POOL_MAX, y, o[2])
        return y


class FIDInceptionC(_Block):
    def __init__(self, in_channels, channels_7x7):
        super().__init__()
        c7 = channels_7x7
        self.branch1x1 = BasicConv2d(in_channels, 192, 1)
        self.branch7x7_1 = BasicConv2d(in_channels, c7, 1)
        self.branch7x7_2 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, (7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(in_channels, c7, 1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, (1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(in_channels, 192, 1)
        self.widths = [192, 192, 192, 192]

    def forward(self, x):
        y, o = self._out(x, self.widths)
        self.branch1x1(x, y, o[0])
        self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)), y, o[1])
        t = self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x))))
        self.branch7x7dbl_5(t, y, o[2])
        self.branch_pool(pool2d(x, 3, 1, 1, POOL_AVG_EXCL), y, o[3])
        return y


class InceptionD(_Block):
    def __init__(self, in_channels):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(in_channels, 192, 1)
        self.branch3x3_2 = BasicConv2d(192, 320, 3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(in_channels, 192, 1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, (1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, (7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, 3, stride=2)
        self.widths = [320, 192, in_channels]

    def forward(self, x):
        y, o = self._out(x, self.widths, stride=2)
        self.branch3x3_2(self.branch3x3_1(x), y, o[0])
        self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))), y, o[1])
        pool2d(x, 3, 2, 0, POOL_MAX, y, o[2])
        return y


class FIDInceptionE(_Block):
    """pool = 'avg': FIDInceptionE_1 (:236-268); pool = 'max': FIDInceptionE_2 (:271-300, the max pool of the original TF graph)."""

    def __init__(self, in_channels, pool):
        super().__init__()
        self.branch1x1 = BasicConv2d(in_channels, 320, 1)
        self.branch3x3_1 = BasicConv2d(in_channels, 384, 1)
        self.branch3x3_2a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 448, 1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, 3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(in_channels, 192, 1)
        self.pool = pool
        self.widths = [320, 384, 384, 384, 384, 192]

    def forward(self, x):
        y, o = self._out(x, self.widths)
        self.branch1x1(x, y, o[0])
        t = self.branch3x3_1(x)
        self.branch3x3_2a(t, y, o[1])
        self.branch3x3_2b(t, y, o[2])
        t = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        self.branch3x3dbl_3a(t, y, o[3])
        self.branch3x3dbl_3b(t, y, o[4])
        self.branch_pool(pool2d(x, 3, 1, 1, POOL_AVG_EXCL if self.pool == 'avg' else POOL_MAX), y, o[5])
        return y


# torchvision's attribute name -> position in the reference wrapper's `blocks` (metric/inception.py:72-108)
TORCHVISION_TO_BLOCKS = {
    'Conv2d_1a_3x3': 'blocks.0.0', 'Conv2d_2a_3x3': 'blocks.0.1', 'Conv2d_2b_3x3': 'blocks.0.2',
    'Conv2d_3b_1x1': 'blocks.1.0', 'Conv2d_4a_3x3': 'blocks.1.1',
    'Mixed_5b': 'blocks.2.0', 'Mixed_5c': 'blocks.2.1', 'Mixed_5d': 'blocks.2.2', 'Mixed_6a': 'blocks.2.3', 'Mixed_6b': 'blocks.2.4',
    'Mixed_6c': 'blocks.2.5', 'Mixed_6d': 'blocks.2.6', 'Mixed_6e': 'blocks.2.7',
    'Mixed_7a': 'blocks.3.0', 'Mixed_7b': 'blocks.3.1', 'Mixed_7c': 'blocks.3.2',
}


class InceptionV3(nn.Module):
    """Same constructor surface, block grouping and return value as the reference's InceptionV3 (use_fid_inception=True only)."""

    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks=[DEFAULT_BLOCK_INDEX], resize_input=True, normalize_input=True, requires_grad=False, use_fid_inception=True):
        super().__init__()
        if not use_fid_inception:
            raise NotImplementedError('only the FID variant of the network is built (the reference never uses the other)')
        if requires_grad:
            raise NotImplementedError('the FID InceptionV3 is a frozen feature extractor here (forward kernels only)')
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, 'Last possible output block index is 3'
        self.blocks = nn.ModuleList()
        self.blocks.append(nn.Sequential(BasicConv2d(3, 32, 3, stride=2), BasicConv2d(32, 32, 3), BasicConv2d(32, 64, 3, padding=1), MaxPool3x3s2()))
        if self.last_needed_block >= 1:
            self.blocks.append(nn.Sequential(BasicConv2d(64, 80, 1), BasicConv2d(80, 192, 3), MaxPool3x3s2()))
        if self.last_needed_block >= 2:
            self.blocks.append(nn.Sequential(FIDInceptionA(192, 32), FIDInceptionA(256, 64), FIDInceptionA(288, 64), InceptionB(288),
                                             FIDInceptionC(768, 128), FIDInceptionC(768, 160), FIDInceptionC(768, 160), FIDInceptionC(768, 192)))
        if self.last_needed_block >= 3:
            self.blocks.append(nn.Sequential(InceptionD(768), FIDInceptionE(1280, 'avg'), FIDInceptionE(2048, 'max'), GlobalAvgPool()))
        for p in self.parameters():
            p.requires_grad = False

    def load_fid_state_dict(self, state_dict, strict=True):
        """Load a torchvision-keyed FID checkpoint (`Mixed_6b.branch7x7_2.conv.weight` ..., what fid_inception_v3 feeds to
        inception.load_state_dict, metric/inception.py:172-173).  `fc.*` / `AuxLogits.*` entries belong to layers the wrapper drops."""
        mapped = {}
        for k, v in state_dict.items():
            head, _, rest = k.partition('.')
            if head in ('fc', 'AuxLogits'):
                continue
            if head not in TORCHVISION_TO_BLOCKS:
                if strict:
                    raise KeyError('unexpected key in the FID checkpoint: %s' % k)
                continue
            pos = TORCHVISION_TO_BLOCKS[head]
            if int(pos.split('.')[1]) > self.last_needed_block:
                continue
            mapped[pos + '.' + rest] = v
        return self.load_state_dict(mapped, strict=strict)

    def forward(self, inp):
        """inp: [B, 3, H, W] in (0, 1) on the GPU (NCHW as the reference passes it, or an NHWC activation).  Returns the selected block outputs."""
        if self.training or torch.is_grad_enabled():
            raise NotImplementedError('the FID InceptionV3 runs in eval mode under no_grad (metric/fid_score.py:183,203)')
        x = ops.to_nhwc(inp.float())
        n, c, h, w = x.shape
        a, b = (2.0, -1.0) if self.normalize_input else (1.0, 0.0)
        if self.resize_input or self.normalize_input:
            ho, wo = (299, 299) if self.resize_input else (h, w)
            y = ops.empty_act(n, c, ho, wo, x.device)
            L.call('cat_resize_bilinear_fwd', ops._p(x), ops.act_cs(x), n, h, w, c, ops._p(y), ops.act_cs(y), ho, wo, a, b, ops._stream())
            x = y
        outp = []
        for idx, block in enumerate(self.blocks):
            x = block(x)
            if idx in self.output_blocks:
                outp.append(x)
            if idx == self.last_needed_block:
                break
        return outp
