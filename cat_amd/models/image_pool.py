"""ImagePool (reference utils/image_pool.py:5-53): history buffer of generated images for the CycleGAN discriminators.  Same
algorithm and the same sequence of `random` draws as the reference (so a seeded run replays identically); images stay NHWC
activations on the device, copies go through the library's kernels."""
import ctypes as C
import random

from .. import _lib as L
from .. import ops


def _copy(src):
    src = ops.conform(src)
    n, c, h, w = src.shape
    dst = ops.empty_act(n, c, h, w, src.device, ops.act_cs(src))
    arr = (C.c_void_p * 1)(src.data_ptr())
    L.call('cat_add_n', arr, 1, ops._p(dst), n * h * w * ops.act_cs(src), ops._stream())
    return dst


class ImagePool:
    def __init__(self, pool_size):
        self.pool_size = pool_size
        if self.pool_size > 0:
            self.num_imgs = 0
            self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        images = ops.conform(images.detach())
        out = []
        for i in range(images.shape[0]):
            image = images[i:i + 1]
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(_copy(image))        # the reference keeps a view of the batch; a copy decouples lifetimes
                out.append(image)
            else:
                p = random.uniform(0, 1)
                if p > 0.5:
                    random_id = random.randint(0, self.pool_size - 1)
                    tmp = self.images[random_id]
                    self.images[random_id] = _copy(image)
                    out.append(tmp)
                else:
                    out.append(image)
        if len(out) == 1:
            return out[0]
        n, c, h, w = images.shape
        cs = ops.act_cs(images)
        dst = ops.empty_act(n, c, h, w, images.device, cs)
        per = h * w * cs
        for i, t in enumerate(out):
            arr = (C.c_void_p * 1)(t.data_ptr())
            L.call('cat_add_n', arr, 1, C.c_void_p(dst.data_ptr() + 4 * i * per), per, ops._stream())
        return dst
