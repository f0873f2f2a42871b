"""BaseInceptionDistiller: the attribute / method surface Trainer and shrink() rely on, following
distillers/base_inception_distiller.py:103-312 for everything on the hot path.  Dataset loaders, FID / mIoU models
and checkpoint directories of the reference constructor are host-side I/O that stays with CAT:
they are only touched when `opt` carries the corresponding paths."""
import itertools
import os
from collections import OrderedDict

import torch

from .. import loss as closs
from .. import networks, ops
from .. import nn as cnn
from ..optim import FusedAdam


class LossValue:
    """A scalar loss kept on the device: sum_i w_i * t_i.  float() synchronises (trainer.py:135-139 does that only every
    print_freq iterations); the hot loop never builds these sums with torch arithmetic kernels."""

    def __init__(self, terms):
        self.terms = [(float(w), t) for w, t in terms]

    def __float__(self):
        return float(sum(w * float(t) for w, t in self.terms))

    def item(self):
        return float(self)

    def __mul__(self, k):
        return LossValue([(w * k, t) for w, t in self.terms])

    __rmul__ = __mul__

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        if isinstance(other, torch.Tensor):
            other = LossValue([(1.0, other)])
        return LossValue(self.terms + other.terms)

    __radd__ = __add__


class BaseInceptionDistiller:
    _FLAGS = [  # base_inception_distiller.py:29-101
        ('--teacher_netG', dict(type=str, default='inception_9blocks')),
        ('--student_netG', dict(type=str, default='inception_9blocks')),
        ('--teacher_ngf', dict(type=int, default=64)),
        ('--student_ngf', dict(type=int, default=48)),
        ('--restore_teacher_G_path', dict(type=str, required=False, default=None)),
        ('--restore_student_G_path', dict(type=str, default=None)),
        ('--restore_A_path', dict(type=str, default=None)),
        ('--restore_D_path', dict(type=str, default=None)),
        ('--restore_O_path', dict(type=str, default=None)),
        ('--recon_loss_type', dict(type=str, default='l1', choices=['l1', 'l2', 'smooth_l1', 'vgg'])),
        ('--distill_G_loss_type', dict(type=str, default='mse', choices=['mse', 'ka'])),
        ('--lambda_distill', dict(type=float, default=1)),
        ('--lambda_recon', dict(type=float, default=100)),
        ('--lambda_gan', dict(type=float, default=1)),
        ('--teacher_dropout_rate', dict(type=float, default=0)),
        ('--student_dropout_rate', dict(type=float, default=0)),
    ]

    @staticmethod
    def modify_commandline_options(parser, is_train):
        assert is_train
        for flag, kw in BaseInceptionDistiller._FLAGS:
            parser.add_argument(flag, **kw)
        return parser

    def __init__(self, opt):
        assert opt.isTrain
        self.opt = opt
        self.gpu_ids = list(getattr(opt, 'gpu_ids', [0]))
        self.isTrain = opt.isTrain
        if not torch.cuda.is_available():
            raise RuntimeError('cat_amd distillers need an MI355X (HIP kernels only; there is no CPU path)')
        ops.default_branch_streams(False)      # one stream: the blocks are a few chip-filling launches (ops.py, branch-level concurrency)
        # one process drives one GPU (torch.distributed / RCCL handles data parallelism): gpu_ids[0] or LOCAL_RANK
        dev_index = int(os.environ.get('LOCAL_RANK', self.gpu_ids[0] if self.gpu_ids else 0))
        self.device = torch.device('cuda', dev_index)
        torch.cuda.set_device(self.device)
        self.save_dir = os.path.join(getattr(opt, 'log_dir', '.'), 'checkpoints')
        self.loss_names = ['G_gan', 'G_distill', 'G_recon', 'D_fake', 'D_real']
        self.optimizers = []
        self.image_paths = []
        self.visual_names = ['real_A', 'Sfake_B', 'Tfake_B', 'real_B']
        self.model_names = ['netG_student', 'netG_teacher', 'netD']
        dev = [dev_index]
        self.netG_teacher = networks.define_G(opt.input_nc, opt.output_nc, opt.teacher_ngf, opt.teacher_netG, opt.norm,
                                              opt.teacher_dropout_rate, opt.init_type, opt.init_gain, dev, opt=opt)
        self.netG_student = networks.define_G(opt.input_nc, opt.output_nc, opt.student_ngf, opt.student_netG, opt.norm,
                                              opt.student_dropout_rate, opt.init_type, opt.init_gain, dev, opt=opt)
        if opt.dataset_mode in ['aligned', 'cityscapes']:
            d_in = opt.input_nc + opt.output_nc
        elif opt.dataset_mode == 'unaligned':
            d_in = opt.output_nc
        else:
            raise NotImplementedError('Unknown dataset mode [%s]!!!' % opt.dataset_mode)
        self.netD = networks.define_D(d_in, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type, opt.init_gain, dev, opt=opt)

        self.netG_teacher.eval()
        self.criterionGAN = closs.GANLoss(opt.gan_mode)
        if opt.recon_loss_type == 'l1':
            self.criterionRecon = closs.L1Loss()
        elif opt.recon_loss_type == 'l2':
            self.criterionRecon = closs.MSELoss()
        else:
            raise NotImplementedError('Unknown reconstruction loss type [%s]!' % opt.recon_loss_type)

        self.mapping_layers = ['down_sampling.9'] + ['features.%d' % i for i in range(2, 11, 3)]
        self.netAs = []
        self.Tacts, self.Sacts = {}, {}
        G_params = []
        for i, n in enumerate(self.mapping_layers):
            ft, fs = opt.teacher_ngf, opt.student_ngf
            netA = cnn.Conv2d(in_channels=fs * 4, out_channels=ft * 4, kernel_size=1).to(self.device)
            G_params.append(netA.parameters())
            self.netAs.append(netA)
            self.loss_names.append('G_distill%d' % i)
        self.optimizer_G = FusedAdam([{'params': self.netG_student.parameters()}, {'params': itertools.chain(*G_params)}],
                                     lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizer_D = FusedAdam(self.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizers += [self.optimizer_G, self.optimizer_D]
        self.is_best = False
        self.mapping_hooks = []
        self._seeds = {}
        self.dp = None     # cat_amd.parallel.DataParallelReducer when world_size > 1

    # -- setup / hooks (base_inception_distiller.py:237-269) -------------------------------------------
    def setup(self, opt, verbose=True):
        self.schedulers = [networks.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
        self.load_networks(verbose)
        if verbose:
            self.print_networks()
        self.add_mapping_hook()

    def add_mapping_hook(self):
        self.mapping_hooks = []
        if self.opt.lambda_distill <= 0:
            return

        def teacher_hook(mem, name):
            def hook(module, input, output):
                mem[name + str(output.device)] = output
            return hook

        def student_hook(mem, name):
            # the tapped activation has two consumers (the next layer and the KA loss): hand each its own alias so the
            # two gradients are summed by FanoutFn's add_n kernel, and return the alias the network continues with
            def hook(module, input, output):
                if torch.is_grad_enabled() and output.requires_grad:
                    keep, cont = ops.fanout(output, 2)
                    mem[name + str(output.device)] = keep
                    return cont
                mem[name + str(output.device)] = output
            return hook

        for net, mem, mk in ((self.netG_teacher, self.Tacts, teacher_hook), (self.netG_student, self.Sacts, student_hook)):
            for n, m in net.named_modules():
                if n in self.mapping_layers:
                    self.mapping_hooks.append(m.register_forward_hook(mk(mem, n)))

    def remove_mapping_hook(self):
        for hook in self.mapping_hooks:
            hook.remove()
        self.mapping_hooks = []

    # -- data ----------------------------------------------------------------------------------------------
    def _to_device_act(self, x):
        return ops.to_nhwc(x.to(self.device, dtype=torch.float32, non_blocking=True))

    def set_input(self, input):
        if self.opt.dataset_mode == 'cityscapes':
            self.image_paths = input['path']
            self.real_A, self.real_B = self._to_device_act(input['label']), self._to_device_act(input['image'])
        else:
            AtoB = self.opt.direction == 'AtoB'
            self.real_A = self._to_device_act(input['A' if AtoB else 'B'])
            self.real_B = self._to_device_act(input['B' if AtoB else 'A'])
            self.image_paths = input.get('A_paths' if AtoB else 'B_paths', [])

    def set_single_input(self, input):
        key = 'label' if self.opt.dataset_mode == 'cityscapes' else 'A'
        self.real_A = self._to_device_act(input[key])
        self.image_paths = input.get('path' if self.opt.dataset_mode == 'cityscapes' else 'A_paths', [])

    # -- the D half of the step (base_inception_distiller.py:293-312) --------------------------------------
    def seed(self, value):
        """Constant 0-d device tensors used as backward seeds (d total / d term)."""
        value = float(value)
        t = self._seeds.get(value)
        if t is None:
            t = torch.full((), value, device=self.device, dtype=torch.float32)
            self._seeds[value] = t
        return t

    def backward_D(self):
        with torch.no_grad():
            if self.opt.dataset_mode == 'aligned':
                fake = ops.Concat2Fn.apply(self.real_A, self.Sfake_B.detach())
                real = ops.Concat2Fn.apply(self.real_A, self.real_B)
            else:
                fake = self.Sfake_B.detach()
                real = self.real_B
        pred_fake = self.netD(fake)
        self.loss_D_fake = self.criterionGAN(pred_fake, False, for_discriminator=True)
        pred_real = self.netD(real)
        self.loss_D_real = self.criterionGAN(pred_real, True, for_discriminator=True)
        self.loss_D = LossValue([(0.5, self.loss_D_fake), (0.5, self.loss_D_real)])
        torch.autograd.backward([self.loss_D_fake, self.loss_D_real], [self.seed(0.5), self.seed(0.5)])
        ops.sync_side_streams()

    # -- backward_D in gradient-ready STAGES (data-parallel schedule, SURVEY 8e legal overlap 2) ---------------------------------------
    def d_stage_plan(self):
        """How backward_D can be cut so that the discriminator's gradient bucket leaves in slices while the rest of the pass still runs:
        -> (segments, slices) or None.  `segments`: the PatchGAN's layer list split in front of its two widest convolutions (pix2pix D:
        [conv1 .. bn2 act | conv3 bn3 act | conv4 bn4 act conv5]); `slices`: per segment the (lo, hi) float range its parameters occupy in
        optimizer_D's flat gradient buffer.  The last segment (the 512 -> 1024 conv: 8.4 M of 11 M parameters) is differentiated first."""
        cached = self.__dict__.get('_d_stage_plan')
        if cached is not None and cached[0] is self.netD:
            return cached[1]
        plan = None
        from ..discriminators import NLayerDiscriminator
        from .. import nn as cnn
        mods = getattr(self.netD, 'model', None)
        flat = self.optimizer_D._ensure_flat() if hasattr(self.optimizer_D, '_ensure_flat') else None
        if isinstance(self.netD, NLayerDiscriminator) and isinstance(mods, cnn.FusedSequential) and flat is not None and len(flat) == 1 and \
                flat[0] is not None and not any(m._forward_hooks or m._forward_pre_hooks for m in mods):
            convs = [i for i, m in enumerate(mods) if isinstance(m, cnn.Conv2d)]
            if len(convs) >= 4:
                order = sorted(convs[1:], key=lambda i: -mods[i].weight.numel())[:2]      # the two widest convs that are not the first layer
                cuts = sorted(order)
                segments = [mods[:cuts[0]], mods[cuts[0]:cuts[1]], mods[cuts[1]:]]
                f = flat[0]
                where = {id(q): (o, f['offs'][j + 1] if j + 1 < len(f['offs']) else f['n']) for j, (q, o) in enumerate(zip(f['params'], f['offs']))}
                slices, ok, pos = [], True, 0
                for seg in segments:
                    ps = list(seg.parameters())
                    lo, hi = where[id(ps[0])][0], where[id(ps[-1])][1]
                    ok = ok and lo == pos and all(where[id(a)][1] == where[id(b)][0] for a, b in zip(ps, ps[1:]))
                    slices.append((lo, hi))
                    pos = hi
                dparams = list(self.netD.parameters())
                if ok and pos == f['n'] and len(dparams) == len(f['params']) and all(a is b for a, b in zip(dparams, f['params'])):
                    plan = (segments, slices)
        self.__dict__['_d_stage_plan'] = (self.netD, plan)
        return plan

    def backward_D_stages(self):
        """backward_D as three callables, same kernels in the same per-parameter order (real graph, then fake graph) -- bit-identical
        gradients: stage 0 = both discriminator forwards, the losses and the backward pass of the LAST segment; stage 1 / 2 = the
        middle / first segment.  After stage k the gradient slice `slices[2 - k]` of optimizer_D's bucket is final."""
        segments, slices = self.d_stage_plan()
        st = {}

        def stage0():
            with torch.no_grad():
                if self.opt.dataset_mode == 'aligned':
                    fake = ops.Concat2Fn.apply(self.real_A, self.Sfake_B.detach())
                    real = ops.Concat2Fn.apply(self.real_A, self.real_B)
                else:
                    fake = self.Sfake_B.detach()
                    real = self.real_B
            cf = [segments[0](fake)]
            cf.append(segments[1](cf[0]))
            pred_fake = segments[2](cf[1])
            self.loss_D_fake = self.criterionGAN(pred_fake, False, for_discriminator=True)
            cr = [segments[0](real)]
            cr.append(segments[1](cr[0]))
            pred_real = segments[2](cr[1])
            self.loss_D_real = self.criterionGAN(pred_real, True, for_discriminator=True)
            self.loss_D = LossValue([(0.5, self.loss_D_fake), (0.5, self.loss_D_real)])
            st['cuts'] = (cf, cr)
            st['g'] = torch.autograd.grad([self.loss_D_fake, self.loss_D_real], [cf[1], cr[1]], [self.seed(0.5), self.seed(0.5)])
            ops.sync_side_streams()

        def stage1():
            cf, cr = st['cuts']
            st['g'] = torch.autograd.grad([cf[1], cr[1]], [cf[0], cr[0]], list(st['g']))
            ops.sync_side_streams()

        def stage2():
            cf, cr = st['cuts']
            torch.autograd.backward([cf[0], cr[0]], list(st['g']))
            st.clear()
            ops.sync_side_streams()
        return [stage0, stage1, stage2], [slices[2], slices[1], slices[0]]

    # -- bookkeeping shared with models/base_model.py:146-232 -------------------------------------------------
    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

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

    def finish_pending(self):
        """Complete work a schedule deferred past optimize_parameters (the data-parallel step keeps the student's gradient all-reduce
        and Adam update in flight until the weights are needed): called by everything that reads weights or optimizer state."""

    def get_current_visuals(self):
        self.finish_pending()
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if hasattr(self, n))

    def update_learning_rate(self, logger=None):
        self.finish_pending()       # a pending Adam step must use the learning rate of the step that produced its gradient
        for scheduler in self.schedulers:
            scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        (logger.print_info if logger is not None else print)('learning rate = %.7f\n' % lr)

    def print_networks(self):
        for name in self.model_names:
            net = getattr(self, name, None)
            if net is not None:
                n = sum(p.numel() for p in net.parameters())
                print('[Network %s] Total number of parameters : %.3f M' % (name, n / 1e6))

    # -- checkpoints: same file names / state_dict keys as base_inception_distiller.py:342-396 ------------------
    def _load(self, net, path, verbose=True):
        if verbose:
            print('Load network at %s' % path)
        net.load_state_dict(torch.load(path, map_location='cpu'))

    def load_networks(self, verbose=True, teacher_only=False, restore_pretrain=True):
        opt = self.opt
        if getattr(opt, 'restore_teacher_G_path', None):
            self._load(self.netG_teacher, opt.restore_teacher_G_path, verbose)
        else:       # the reference loads it unconditionally (base_inception_distiller.py:343): distilling from a random teacher is never intended
            import warnings
            warnings.warn('restore_teacher_G_path is not set: the teacher keeps its initialisation (synthetic-weight runs only)')
        if getattr(opt, 'restore_student_G_path', None):
            self._load(self.netG_student, opt.restore_student_G_path, verbose)
        if getattr(opt, 'restore_D_path', None):
            self._load(self.netD, opt.restore_D_path, verbose)
        if getattr(opt, 'restore_A_path', None):
            for i, netA in enumerate(self.netAs):
                self._load(netA, '%s-%d.pth' % (opt.restore_A_path, i), verbose)
        if getattr(opt, 'restore_O_path', None):
            for i, optimizer in enumerate(self.optimizers):
                optimizer.load_state_dict(torch.load('%s-%d.pth' % (opt.restore_O_path, i), map_location='cpu'))
                for param_group in optimizer.param_groups:
                    param_group['lr'] = opt.lr

    def save_networks(self, epoch):
        self.finish_pending()
        os.makedirs(self.save_dir, exist_ok=True)

        def cpu_sd(net):   # NCHW/OIHW-contiguous values, the checkpoint wire format
            return OrderedDict((k, v.detach().cpu().contiguous()) for k, v in net.state_dict().items())

        torch.save(cpu_sd(self.netG_student), os.path.join(self.save_dir, '%s_net_G.pth' % epoch))
        torch.save(cpu_sd(self.netD), os.path.join(self.save_dir, '%s_net_D.pth' % epoch))
        for i, net in enumerate(self.netAs):
            torch.save(cpu_sd(net), os.path.join(self.save_dir, '%s_net_A-%d.pth' % (epoch, i)))
        for i, optimizer in enumerate(self.optimizers):
            torch.save(optimizer.state_dict(), os.path.join(self.save_dir, '%s_optim-%d.pth' % (epoch, i)))

    def evaluate_model(self, step, save_image=False):
        """reference inception_distiller.py:204-281: student (and teacher) inference over `self.eval_dataloader` on the HIP kernels, image
        dumps, `is_best` / running-mean bookkeeping; the FID / mIoU networks themselves are the integrator's callables
        `self.fid_fn(fakes)`, `self.miou_fn(fakes, names)` (cat_amd/distillers/evaluation.py)."""
        from . import evaluation as E
        self.finish_pending()
        aligned = self.opt.dataset_mode == 'aligned'

        def images(j):
            out = {'input': E.tensor2im(self.real_A[j]), 'Sfake': E.tensor2im(self.Sfake_B[j]), 'Tfake': E.tensor2im(self.Tfake_B[j])}
            if aligned:
                out['real'] = E.tensor2im(self.real_B[j])
            return out
        want_miou = 'cityscapes' in str(getattr(self.opt, 'dataroot', '')) and getattr(self.opt, 'direction', 'AtoB') == 'BtoA'
        return E.evaluate(self, step, self.netG_student, self.set_input if aligned else self.set_single_input, images, True, want_miou,
                          save_all=save_image)

    def test(self, teacher_forward=True):
        self.finish_pending()
        with torch.no_grad():
            self.forward(teacher_forward=teacher_forward)
