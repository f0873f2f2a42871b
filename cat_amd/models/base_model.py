"""BaseModel: the host-side glue of models/base_model.py:12-232 that `train.py` / `Trainer` touch (device, save_dir, setup,
schedulers, loss dictionary, requires_grad toggling, checkpoints with the reference's file names and state_dict keys).  Dataset
loaders and FID / mIoU evaluation stay with the reference."""
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch

from .. import networks, ops


class BaseModel(ABC):
    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = list(getattr(opt, 'gpu_ids', [0]))
        self.isTrain = opt.isTrain
        if not torch.cuda.is_available():
            raise RuntimeError('cat_amd models need an MI355X (HIP kernels only; there is no CPU path)')
        dev_index = int(os.environ.get('LOCAL_RANK', self.gpu_ids[0] if self.gpu_ids else 0))
        self.device = torch.device('cuda', dev_index)
        torch.cuda.set_device(self.device)
        self._dev_ids = [dev_index]
        self.save_dir = os.path.join(getattr(opt, 'log_dir', '.'), 'checkpoints')
        self.model_names, self.visual_names, self.image_paths, self.loss_names, self.optimizers = [], [], [], [], []
        self.metric = 0
        self._seeds = {}
        self.dp = None

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def set_input(self, input):
        pass

    @abstractmethod
    def forward(self):
        pass

    @abstractmethod
    def optimize_parameters(self, steps):
        pass

    def _to_device_act(self, x):
        return ops.to_nhwc(x.to(self.device, dtype=torch.float32, non_blocking=True))

    def seed(self, value):
        """Constant 0-d device tensors used as backward seeds (d total / d term)."""
        value = float(value)
        t = self._seeds.get(value)
        if t is None:
            t = torch.full((), value, device=self.device, dtype=torch.float32)
            self._seeds[value] = t
        return t

    def backward_terms(self, terms):
        """sum_i w_i * t_i .backward() without building the sum: [(w_i, t_i)] seeds torch.autograd.backward."""
        terms = [(w, t) for w, t in terms if t.requires_grad]
        torch.autograd.backward([t for _, t in terms], [self.seed(w) for w, _ in terms])
        ops.sync_side_streams()

    def setup(self, opt, verbose=True):
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
        self.load_networks(verbose)
        if verbose:
            self.print_networks()

    def print_networks(self):
        for name in self.model_names:
            net = getattr(self, 'net' + name)
            print('[Network %s] Total number of parameters : %.3f M' % (name, sum(p.numel() for p in net.parameters()) / 1e6))

    def eval(self):
        for name in self.model_names:
            getattr(self, 'net' + name).eval()

    def train(self):
        for name in self.model_names:
            getattr(self, 'net' + name).train()

    def test(self):
        with torch.no_grad():
            self.forward()

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self, logger=None):
        for scheduler in self.schedulers:
            scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        (logger.print_info if logger is not None else print)('learning rate = %.7f\n' % lr)

    def get_current_visuals(self):
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if hasattr(self, n))

    def get_current_losses(self):
        errors_set = OrderedDict()
        for name in self.loss_names:
            if not hasattr(self, 'loss_' + name):
                continue
            if any(ch.isdigit() for ch in name):
                key = 'Specific_loss/' + name
            elif name.startswith('D_'):
                key = 'D_loss/' + name
            elif name.startswith('G_'):
                key = 'G_loss/' + name
            else:
                assert False
            errors_set[key] = float(getattr(self, 'loss_' + name))
        return errors_set

    def load_networks(self, verbose=True, teacher_only=False, restore_pretrain=True):
        for name in self.model_names:
            path = getattr(self.opt, 'restore_%s_path' % name, None)
            if path is not None:
                if verbose:
                    print('Load network at %s' % path)
                getattr(self, 'net' + name).load_state_dict(torch.load(path, map_location='cpu'))

    def save_networks(self, epoch):
        os.makedirs(self.save_dir, exist_ok=True)
        for name in self.model_names:
            net = getattr(self, 'net' + name)
            sd = OrderedDict((k, v.detach().cpu().contiguous()) for k, v in net.state_dict().items())
            torch.save(sd, os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name)))

    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

    def enable_data_parallel(self, reducer):
        """One process per GPU (cat_amd.parallel): replicas are synchronised once, then the flat gradient buckets are all-reduced
        after each backward pass.  Every loss of these models is a mean over the batch = plain gradient averaging."""
        self.dp = reducer
        reducer.broadcast_parameters([getattr(self, 'net' + n) for n in self.model_names])

    def evaluate_model(self, step):
        raise NotImplementedError('teacher-training models: evaluate with the distillers\' path -- cat_amd.distillers.evaluation.evaluate + '
                                  'attach_fid (InceptionV3 pool3 features on the HIP kernels, cat_amd.metric); mIoU needs the reference\'s DRN '
                                  'weights and the cityscapes data (SURVEY §2 rows 18-19)')
