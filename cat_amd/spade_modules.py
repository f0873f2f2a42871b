"""SPADEDistillerModules: the nn.Module that owns teacher / student / discriminator and evaluates the GauGAN distillation losses
inside `forward(mode=...)`, with the reference's attribute names, modes and loss keys
(models/modules/spade_modules/base_spade_distiller_modules.py:14-175, spade_distiller_modules.py:11-31,
spade_model_modules.py:136-155).

The reference wraps this module in DataParallelWithCallback; here one process drives one GPU, SynchronizedBatchNorm layers
all-reduce their statistics over RCCL (ops.set_bn_sync) and the gradient buckets are averaged by cat_amd.parallel, which is
what `.mean()` over replicas (models/spade_model.py:191,200) amounts to.

Losses are LossValue objects (weighted sums of 0-d device tensors): nothing is added with torch arithmetic on the hot path,
`LossValue.backward()` seeds torch.autograd.backward with the weights."""
import copy

import torch
from torch import nn

from . import loss as closs
from . import networks, ops
from . import nn as cnn
from .optim import FusedAdam


class LossValue:
    """sum_i w_i * t_i over 0-d device tensors; float() synchronises."""

    def __init__(self, terms):
        self.terms = [(float(w), t) for w, t in terms]

    def __float__(self):
        return float(sum(w * float(t) for w, t in self.terms))

    def item(self):
        return float(self)

    def detach(self):
        return self

    def mean(self):          # losses['loss_G'].mean() in models/spade_model.py:191 (one replica per process)
        return self

    def __mul__(self, k):
        return LossValue([(w * k, t) for w, t in self.terms])

    __rmul__ = __mul__

    def __truediv__(self, k):
        return self * (1.0 / k)

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        if isinstance(other, torch.Tensor):
            other = LossValue([(1.0, other)])
        return LossValue(self.terms + other.terms)

    __radd__ = __add__

    _SEEDS = {}

    def backward(self):
        terms = [(w, t) for w, t in self.terms if t.requires_grad]
        seeds = []
        for w, t in terms:
            key = (t.device, w)
            s = self._SEEDS.get(key)
            if s is None:
                s = torch.full((), w, device=t.device, dtype=torch.float32)
                self._SEEDS[key] = s
            seeds.append(s)
        torch.autograd.backward([t for _, t in terms], seeds)
        ops.sync_side_streams()


class SPADEDistillerModules(nn.Module):
    def __init__(self, opt):
        assert opt.isTrain
        opt = copy.deepcopy(opt)
        if len(opt.gpu_ids) > 0:
            opt.gpu_ids = opt.gpu_ids[:1]
        self.gpu_ids = opt.gpu_ids
        super(SPADEDistillerModules, self).__init__()
        self.opt = opt
        self.model_names = ['G_student', 'G_teacher', 'D']
        teacher_opt = copy.deepcopy(opt)
        teacher_opt.norm_G = opt.teacher_norm_G
        teacher_opt.ngf = opt.teacher_ngf
        self.netG_teacher = networks.define_G(opt.input_nc, opt.output_nc, opt.teacher_ngf, opt.teacher_netG, opt.norm, 0, opt.init_type,
                                              opt.init_gain, self.gpu_ids, opt=teacher_opt)
        student_opt = copy.deepcopy(opt)
        student_opt.norm_G = opt.student_norm_G
        student_opt.ngf = opt.student_ngf
        self.netG_student = networks.define_G(opt.input_nc, opt.output_nc, opt.student_ngf, opt.student_netG, opt.norm, 0, opt.init_type,
                                              opt.init_gain, self.gpu_ids, opt=student_opt)
        # netG_pretrained (base_spade_distiller_modules.py:49-61) only feeds load_pretrained_weight: it is built (on the host) inside
        # load_networks when `restore_pretrained_G_path` is set, not kept resident here.
        self.netD = networks.define_D(opt.input_nc + opt.output_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type,
                                      opt.init_gain, self.gpu_ids, opt=opt)
        self.mapping_layers = ['head_0', 'G_middle_1', 'up_1']
        self.netAs = nn.ModuleList()
        for mapping_layer in self.mapping_layers:
            if mapping_layer != 'up_1':
                fs, ft = opt.student_ngf * 16, opt.teacher_ngf * 16
            else:
                fs, ft = opt.student_ngf * 4, opt.teacher_ngf * 4
            netA = cnn.Conv2d(in_channels=fs, out_channels=ft, kernel_size=1)
            networks.init_net(netA, opt.init_type, opt.init_gain, self.gpu_ids)
            self.netAs.append(netA)
        self.criterionGAN = closs.GANLoss(opt.gan_mode)
        self.criterionFeat = closs.L1Loss()
        self.criterionMSE = closs.MSELoss()
        self.criterionVGG = closs.VGGLoss(width_div=getattr(opt, 'vgg_width_div', 1))
        if len(self.gpu_ids) > 0:
            self.criterionVGG.to(torch.device('cuda', self.gpu_ids[0]))
        self.optimizers = []
        self.netG_teacher.eval()

    def train(self, mode=True):
        """nn.Module.train, except that the frozen teacher and VGG stay in eval mode (base_spade_distiller_modules.py:89)."""
        super().train(mode)
        self.netG_teacher.eval()
        self.criterionVGG.eval()
        return self

    def create_optimizers(self):
        """base_spade_distiller_modules.py:91-107 (TTUR: betas (0, 0.9), lr/2 for G, lr*2 for D)."""
        if self.opt.no_TTUR:
            beta1, beta2 = self.opt.beta1, self.opt.beta2
            G_lr, D_lr = self.opt.lr, self.opt.lr
        else:
            beta1, beta2 = 0.0, 0.9
            G_lr, D_lr = self.opt.lr / 2, self.opt.lr * 2
        G_params = list(self.netG_student.parameters())
        for netA in self.netAs:
            G_params += list(netA.parameters())
        optimizer_G = FusedAdam(G_params, lr=G_lr, betas=(beta1, beta2))
        optimizer_D = FusedAdam(list(self.netD.parameters()), lr=D_lr, betas=(beta1, beta2))
        return optimizer_G, optimizer_D

    def forward(self, input_semantics, real_B=None, mode='generate_fake'):
        if mode == 'generate_fake':
            with torch.no_grad():
                Tfake_B = self.netG_teacher(input_semantics)
                Sfake_B = self.netG_student(input_semantics)
            return Tfake_B, Sfake_B
        elif mode == 'G_loss':
            assert real_B is not None
            return self.compute_G_loss(input_semantics, real_B)
        elif mode == 'D_loss':
            assert real_B is not None
            return self.compute_D_loss(input_semantics, real_B)
        raise NotImplementedError('Unknown forward mode [%s]!!!' % mode)

    def profile(self, input_semantics):
        raise NotImplementedError('The distiller is only for training!!!')

    # -- losses -------------------------------------------------------------------------------------------------------------
    def calc_distill_loss(self, Tacts, Sacts):
        """spade_distiller_modules.py:17-31."""
        kind = self.opt.distill_G_loss_type
        if kind not in ('ka', 'mse'):
            raise NotImplementedError(kind)
        losses = {}
        for i, netA in enumerate(self.netAs):
            layer = self.mapping_layers[i]
            if kind == 'mse':
                losses['G_distill%d' % i] = LossValue([(1.0, self.criterionMSE(netA(Sacts[layer]), Tacts[layer]))])
            else:
                losses['G_distill%d' % i] = LossValue([(-1.0, closs.KA(Sacts[layer], Tacts[layer]))])
        total = LossValue([(w * self.opt.lambda_distill, t) for lv in losses.values() for w, t in lv.terms])
        return total, losses

    def compute_G_loss(self, input_semantics, real_B):
        """base_spade_distiller_modules.py:128-156."""
        opt = self.opt
        with torch.no_grad():
            Tfake_B, Tacts = self.netG_teacher(input_semantics, mapping_layers=self.mapping_layers)
        Sfake_B, Sacts = self.netG_student(input_semantics, mapping_layers=self.mapping_layers)
        # every tapped student activation feeds the next block AND the KA loss; the generator returns the tensor the network
        # continued with, so alias it for the loss (gradients are summed by FanoutFn's add_n)
        loss_G_distill, losses = self.calc_distill_loss(Tacts, Sacts)
        sf_d, sf_v = ops.fanout(Sfake_B, 2)
        pred_fake, pred_real = self.discriminate(input_semantics, sf_d, real_B)
        num_D = len(pred_fake)
        loss_G_gan = LossValue([(opt.lambda_gan / num_D, self.criterionGAN(p[-1], True, for_discriminator=False)) for p in pred_fake])
        feat = []
        for i in range(num_D):
            for j in range(len(pred_fake[i]) - 1):
                feat.append((opt.lambda_feat / num_D, self.criterionFeat(pred_fake[i][j], pred_real[i][j])))
        loss_G_feat = LossValue(feat)
        loss_G_vgg = LossValue([(w * opt.lambda_vgg, t) for w, t in self.criterionVGG.terms(sf_v, real_B)])
        loss_G = loss_G_gan + loss_G_distill + loss_G_feat + loss_G_vgg
        losses.update({'loss_G': loss_G, 'G_gan': loss_G_gan, 'G_distill': loss_G_distill, 'G_feat': loss_G_feat, 'G_vgg': loss_G_vgg})
        self._last = (Tfake_B, Sfake_B)
        return losses

    def compute_D_loss(self, input_semantics, real_B):
        """base_spade_distiller_modules.py:158-175."""
        with torch.no_grad():
            fake_B = self.netG_student(input_semantics)
        pred_fake, pred_real = self.discriminate(input_semantics, fake_B, real_B)
        num_D = len(pred_fake)
        loss_D_fake = LossValue([(1.0 / num_D, self.criterionGAN(p[-1], False, for_discriminator=True)) for p in pred_fake])
        loss_D_real = LossValue([(1.0 / num_D, self.criterionGAN(p[-1], True, for_discriminator=True)) for p in pred_real])
        return {'loss_D': loss_D_fake + loss_D_real, 'D_fake': loss_D_fake, 'D_real': loss_D_real}

    def discriminate(self, input_semantics, fake_B, real_B):
        """spade_model_modules.py:136-141: ONE discriminator pass over the 2N batch [sem|fake ; sem|real]."""
        fake_and_real = ops.DiscInputFn.apply(input_semantics, fake_B, real_B)
        return self.divide_pred(self.netD(fake_and_real))

    def divide_pred(self, pred):
        """spade_model_modules.py:143-155.  Intermediate features feed the next layer AND the feature-matching loss."""
        fake, real = [], []
        for p in pred:
            halves = [ops.BatchHalvesFn.apply(t) for t in p]
            fake.append([h[0] for h in halves])
            real.append([h[1] for h in halves])
        return fake, real

    # -- checkpoints (same file layout as base_spade_distiller_modules.py:177-214) ---------------------------------------------
    def load_networks(self, verbose=True, teacher_only=False, restore_pretrain=True):
        opt = self.opt

        def load(net, path):
            if verbose:
                print('Load network at %s' % path)
            net.load_state_dict(torch.load(path, map_location='cpu'))

        if getattr(opt, 'restore_pretrained_G_path', None) is not None and restore_pretrain:
            # spade_distiller_modules.py:33-44: a wider pretrained generator seeds the student (host-side, once; cat_amd/weight_transfer.py)
            import copy
            from . import networks
            from .weight_transfer import load_pretrained_weight
            popt = copy.deepcopy(opt)
            popt.norm_G, popt.ngf = opt.pretrained_norm_G, opt.pretrained_ngf
            pre = networks.define_G(opt.input_nc, opt.output_nc, opt.pretrained_ngf, opt.pretrained_netG, opt.norm, 0, opt.init_type, opt.init_gain,
                                    [], opt=popt)
            load(pre, opt.restore_pretrained_G_path)
            load_pretrained_weight(opt.pretrained_netG, opt.student_netG, pre, self.netG_student, opt.pretrained_ngf, opt.student_ngf)
            del pre
        if getattr(opt, 'restore_teacher_G_path', None):
            load(self.netG_teacher, opt.restore_teacher_G_path)
        else:       # the reference loads it unconditionally (spade_model_modules.py): a random teacher is never intended
            import warnings
            warnings.warn('restore_teacher_G_path is not set: the teacher keeps its initialisation (synthetic-weight runs only)')
        if teacher_only:
            return
        if getattr(opt, 'restore_student_G_path', None) is not None:
            load(self.netG_student, opt.restore_student_G_path)
        if getattr(opt, 'restore_D_path', None) is not None:
            load(self.netD, opt.restore_D_path)
        if getattr(opt, 'restore_A_path', None) is not None:
            for i, netA in enumerate(self.netAs):
                load(netA, '%s-%d.pth' % (opt.restore_A_path, i))

    def save_networks(self, epoch, save_dir):
        import os
        from collections import OrderedDict

        def cpu_sd(net):
            return OrderedDict((k, v.detach().cpu().contiguous()) for k, v in net.state_dict().items())

        torch.save(cpu_sd(self.netG_student), os.path.join(save_dir, '%s_net_G.pth' % epoch))
        torch.save(cpu_sd(self.netD), os.path.join(save_dir, '%s_net_D.pth' % epoch))
        for i, net in enumerate(self.netAs):
            torch.save(cpu_sd(net), os.path.join(save_dir, '%s_net_A-%d.pth' % (epoch, i)))
