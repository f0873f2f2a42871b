"""BaseSPADEDistiller: the GauGAN distillation step with the reference's class surface (distillers/base_spade_distiller.py:26-234
on top of models/spade_model.py:132-203): set_input / preprocess_input / get_edges, forward, backward_G / backward_D,
optimize_parameters (G step first, then D step), loss_* attributes, get_current_losses.

Data parallelism is process-per-GPU: SynchronizedBatchNorm statistics and the two gradient buckets travel over RCCL
(cat_amd.parallel); losses are per-replica means averaged over replicas (spade_model.py:191,200) = plain gradient averaging."""
import argparse
import os
from collections import OrderedDict

import torch

from .. import ops
from ..prune import model_profiling
from ..spade_modules import SPADEDistillerModules


class BaseSPADEDistiller:
    _FLAGS = [  # base_spade_distiller.py:27-138
        ('--num_upsampling_layers', dict(choices=('normal', 'more', 'most'), default='more')),
        ('--teacher_netG', dict(type=str, default='inception_spade', choices=['inception_spade'])),
        ('--student_netG', dict(type=str, default='inception_spade', choices=['inception_spade'])),
        ('--teacher_ngf', dict(type=int, default=64)),
        ('--student_ngf', dict(type=int, default=48)),
        ('--teacher_norm_G', dict(type=str, default='spadesyncbatch3x3')),
        ('--student_norm_G', dict(type=str, default='spadesyncbatch3x3')),
        ('--restore_teacher_G_path', dict(type=str, required=False, default=None)),
        ('--restore_student_G_path', dict(type=str, default=None)),
        ('--restore_A_path', dict(type=str, default=None)),
        ('--restore_D_path', dict(type=str, default=None)),
        ('--restore_O_path', dict(type=str, default=None)),
        ('--lambda_gan', dict(type=float, default=1)),
        ('--lambda_feat', dict(type=float, default=10)),
        ('--lambda_vgg', dict(type=float, default=10)),
        ('--lambda_distill', dict(type=float, default=10)),
        ('--distill_G_loss_type', dict(type=str, default='mse', choices=['mse', 'ka'])),
        ('--beta2', dict(type=float, default=0.999)),
        ('--no_TTUR', dict(action='store_true')),
        ('--no_fid', dict(action='store_true')),
        ('--no_mIoU', dict(action='store_true')),
    ]

    @staticmethod
    def modify_commandline_options(parser, is_train):
        assert isinstance(parser, argparse.ArgumentParser)
        for flag, kw in BaseSPADEDistiller._FLAGS:
            parser.add_argument(flag, **kw)
        parser.set_defaults(netD='multi_scale', ndf=64, dataset_mode='cityscapes', batch_size=16, print_freq=50,
                            save_latest_freq=10000000000, save_epoch_freq=10, nepochs=100, nepochs_decay=100, init_type='xavier')
        return parser

    def __init__(self, opt):
        assert opt.isTrain
        self.opt = opt
        self.gpu_ids = list(getattr(opt, 'gpu_ids', [0]))
        self.isTrain = opt.isTrain
        if not torch.cuda.is_available():
            raise RuntimeError('cat_amd distillers need an MI355X (HIP kernels only; there is no CPU path)')
        ops.default_branch_streams(True)       # ~1 500 small launches per step: branch / weight-gradient side streams gain 6 - 9 %
        dev_index = int(os.environ.get('LOCAL_RANK', self.gpu_ids[0] if self.gpu_ids else 0))
        self.device = torch.device('cuda', dev_index)
        torch.cuda.set_device(self.device)
        self.save_dir = os.path.join(getattr(opt, 'log_dir', '.'), 'checkpoints')
        self.model_names = ['G_student', 'G_teacher', 'D']
        self.visual_names = ['labels', 'Tfake_B', 'Sfake_B', 'real_B']
        self.loss_names = ['G_gan', 'G_feat', 'G_vgg', 'G_distill', 'D_real', 'D_fake']
        mopt = argparse.Namespace(**vars(opt))
        mopt.gpu_ids = [dev_index]
        self.modules = SPADEDistillerModules(mopt).to(self.device)
        self.modules_on_one_gpu = self.modules
        for i in range(len(self.modules_on_one_gpu.mapping_layers)):
            self.loss_names.append('G_distill%d' % i)
        self.optimizer_G, self.optimizer_D = self.modules_on_one_gpu.create_optimizers()
        self.optimizers = [self.optimizer_G, self.optimizer_D]
        self.best_fid = 1e9
        self.best_mIoU = -1e9
        self.fids, self.mIoUs = [], []
        self.is_best = False
        self.dp = None
        h, w, c = getattr(opt, 'data_height', None), getattr(opt, 'data_width', None), getattr(opt, 'data_channel', opt.semantic_nc)
        if h is not None and w is not None:
            model_profiling(self.modules_on_one_gpu.netG_teacher, h, w, channel=c)
            model_profiling(self.modules_on_one_gpu.netG_student, h, w, channel=c)

    # -- data (models/spade_model.py:132-179) ---------------------------------------------------------------------------------
    def set_input(self, input):
        self.data = input
        self.image_paths = input.get('path', [])
        self.labels = input['label'].to(self.device)
        self.input_semantics, self.real_B = self.preprocess_input(input)

    def preprocess_input(self, data):
        label = data['label'].to(self.device)
        nc = self.opt.input_nc + 1 if getattr(self.opt, 'contain_dontcare_label', False) else self.opt.input_nc
        inst = None if getattr(self.opt, 'no_instance', False) else data['instance'].to(self.device)
        input_semantics = ops.onehot_edges(label, inst, nc)
        return input_semantics, ops.to_nhwc(data['image'].to(self.device, dtype=torch.float32))

    def get_edges(self, t):
        n, c, h, w = t.shape
        zero = torch.zeros((n, 1, h, w), device=t.device, dtype=torch.int32)
        return ops.onehot_edges(zero, t.to(self.device), 0)

    # -- the step (models/spade_model.py:189-203, base_spade_distiller.py:226-234) --------------------------------------------
    def forward(self, on_one_gpu=False):
        self.Tfake_B, self.Sfake_B = self.modules_on_one_gpu(self.input_semantics)

    def test(self):
        with torch.no_grad():
            self.forward(on_one_gpu=True)

    def backward_G(self):
        losses = self.modules(self.input_semantics, self.real_B, mode='G_loss')
        for loss_name in self.loss_names:
            if loss_name.startswith('G'):
                setattr(self, 'loss_%s' % loss_name, losses[loss_name])
        self.Tfake_B, self.Sfake_B = self.modules_on_one_gpu._last
        losses['loss_G'].backward()

    def backward_D(self):
        losses = self.modules(self.input_semantics, self.real_B, mode='D_loss')
        for loss_name in self.loss_names:
            if loss_name.startswith('D'):
                setattr(self, 'loss_%s' % loss_name, losses[loss_name])
        losses['loss_D'].backward()

    def optimize_parameters(self, steps):
        self.set_requires_grad(self.modules_on_one_gpu.netD, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        if self.dp is not None:
            self.dp.reduce(self.optimizer_G)
        self.optimizer_G.step()
        self.set_requires_grad(self.modules_on_one_gpu.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        if self.dp is not None:
            self.dp.reduce(self.optimizer_D)
        self.optimizer_D.step()

    def enable_data_parallel(self, reducer):
        """Attach a cat_amd.parallel.DataParallelReducer: replicas are synchronised once; afterwards SynchronizedBatchNorm
        statistics ([sum x | sum x^2] per layer) and the two flat gradient buckets are all-reduced every step."""
        self.dp = reducer
        m = self.modules_on_one_gpu
        reducer.broadcast_parameters([m.netG_teacher, m.netG_student, m.netD] + list(m.netAs))
        ops.set_bn_sync(reducer)

    # -- bookkeeping (models/base_model.py:146-232) ---------------------------------------------------------------------------
    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

    def setup(self, opt, verbose=True):
        from .. import networks
        self.schedulers = [networks.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
        self.load_networks(verbose)
        if verbose:
            self.print_networks()

    def get_current_losses(self):
        errors_set = OrderedDict()
        for name in self.loss_names:
            if not hasattr(self, 'loss_' + name):
                continue
            if any(ch.isdigit() for ch in name):
                key = 'Specific_loss/' + name
            elif name.startswith('D_'):
                key = 'D_loss/' + name
            else:
                key = 'G_loss/' + name
            errors_set[key] = float(getattr(self, 'loss_' + name))
        return errors_set

    def finish_pending(self):
        """No deferred work in the SPADE step (gradient buckets are reduced synchronously); kept for the Trainer-facing surface."""

    def get_current_visuals(self):
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if hasattr(self, n))

    def update_learning_rate(self, logger=None):
        for scheduler in self.schedulers:
            scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        (logger.print_info if logger is not None else print)('learning rate = %.7f\n' % lr)

    def print_networks(self):
        m = self.modules_on_one_gpu
        for name, net in (('G_student', m.netG_student), ('G_teacher', m.netG_teacher), ('D', m.netD)):
            print('[Network %s] Total number of parameters : %.3f M' % (name, sum(p.numel() for p in net.parameters()) / 1e6))

    def load_networks(self, verbose=True, teacher_only=False, restore_pretrain=True):
        self.modules_on_one_gpu.load_networks(verbose, teacher_only=teacher_only, restore_pretrain=restore_pretrain)
        if getattr(self.opt, 'restore_O_path', None) is not None:
            for i, optimizer in enumerate(self.optimizers):
                optimizer.load_state_dict(torch.load('%s-%d.pth' % (self.opt.restore_O_path, i), map_location='cpu'))
                for param_group in optimizer.param_groups:
                    param_group['lr'] = self.opt.lr

    def save_networks(self, epoch):
        os.makedirs(self.save_dir, exist_ok=True)
        self.modules_on_one_gpu.save_networks(epoch, self.save_dir)
        for i, optimizer in enumerate(self.optimizers):
            torch.save(optimizer.state_dict(), os.path.join(self.save_dir, '%s_optim-%d.pth' % (epoch, i)))

    def evaluate_model(self, step, save_image=False):
        """reference spade_distiller.py:96-180: see cat_amd/distillers/evaluation.py (generator passes here, metric networks attached by
        the integrator as `self.fid_fn`, `self.miou_fn`)."""
        from . import evaluation as E

        def images(j):
            return {'input': E.tensor2label(self.input_semantics[j], self.opt.input_nc + 2), 'real': E.tensor2im(self.real_B[j]),
                    'Tfake': E.tensor2im(self.Tfake_B[j]), 'Sfake': E.tensor2im(self.Sfake_B[j])}
        want_fid = not getattr(self.opt, 'no_fid', False)
        want_miou = 'cityscapes' in str(getattr(self.opt, 'dataroot', '')) and not getattr(self.opt, 'no_mIoU', False)
        return E.evaluate(self, step, self.modules_on_one_gpu.netG_student, self.set_input, images, want_fid, want_miou, save_all=save_image)
