"""CycleGANModel: the teacher-training step of `train.py --model cycle_gan` (reference models/cycle_gan_model.py:24-303) on the
gfx950 kernels: two generators, two PatchGANs, GAN + cycle + identity losses, image history pools; G step first, then both D steps.
Each generator runs three times per step (translation, reconstruction, identity): its parameter gradients accumulate in the flat
gradient buffer across the three graphs."""
import itertools

import torch

from .. import loss as closs
from .. import networks, ops
from ..distillers.base_inception_distiller import LossValue
from ..optim import FusedAdam
from .base_model import BaseModel
from .image_pool import ImagePool


class CycleGANModel(BaseModel):
    _FLAGS = [  # cycle_gan_model.py:27-108
        ('--restore_G_A_path', dict(type=str, default=None)),
        ('--restore_D_A_path', dict(type=str, default=None)),
        ('--restore_G_B_path', dict(type=str, default=None)),
        ('--restore_D_B_path', dict(type=str, default=None)),
        ('--lambda_A', dict(type=float, default=10.0)),
        ('--lambda_B', dict(type=float, default=10.0)),
        ('--lambda_identity', dict(type=float, default=0.5)),
        ('--real_stat_A_path', dict(type=str, required=False, default=None)),
        ('--real_stat_B_path', dict(type=str, required=False, default=None)),
    ]

    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        assert is_train
        for flag, kw in CycleGANModel._FLAGS:
            parser.add_argument(flag, **kw)
        parser.set_defaults(norm='instance', dataset_mode='unaligned', batch_size=1, ndf=64, gan_mode='lsgan', nepochs=100,
                            nepochs_decay=100, save_epoch_freq=20)
        return parser

    def __init__(self, opt):
        assert opt.isTrain
        assert opt.direction == 'AtoB'
        assert opt.dataset_mode == 'unaligned'
        BaseModel.__init__(self, opt)
        self.loss_names = ['D_A', 'G_A', 'G_cycle_A', 'G_idt_A', 'D_B', 'G_B', 'G_cycle_B', 'G_idt_B']
        visual_names_A = ['real_A', 'fake_B', 'rec_A']
        visual_names_B = ['real_B', 'fake_A', 'rec_B']
        if opt.lambda_identity > 0.0:
            visual_names_A.append('idt_B')
            visual_names_B.append('idt_A')
        self.visual_names = visual_names_A + visual_names_B
        self.model_names = ['G_A', 'G_B', 'D_A', 'D_B']
        dev = self._dev_ids
        self.netG_A = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, opt.dropout_rate, opt.init_type,
                                        opt.init_gain, dev, opt=opt)
        self.netG_B = networks.define_G(opt.output_nc, opt.input_nc, opt.ngf, opt.netG, opt.norm, opt.dropout_rate, opt.init_type,
                                        opt.init_gain, dev, opt=opt)
        self.netD_A = networks.define_D(opt.output_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type, opt.init_gain, dev,
                                        opt=opt)
        self.netD_B = networks.define_D(opt.input_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type, opt.init_gain, dev,
                                        opt=opt)
        if opt.lambda_identity > 0.0:
            assert opt.input_nc == opt.output_nc
        self.fake_A_pool = ImagePool(opt.pool_size)
        self.fake_B_pool = ImagePool(opt.pool_size)
        self.criterionGAN = closs.GANLoss(opt.gan_mode)
        self.criterionCycle = closs.L1Loss()
        self.criterionIdt = closs.L1Loss()
        self.optimizer_G = FusedAdam(itertools.chain(self.netG_A.parameters(), self.netG_B.parameters()), lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizer_D = FusedAdam(itertools.chain(self.netD_A.parameters(), self.netD_B.parameters()), lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizers = [self.optimizer_G, self.optimizer_D]
        self.best_fid_A, self.best_fid_B, self.best_mIoU = 1e9, 1e9, -1e9
        self.fids_A, self.fids_B, self.mIoUs = [], [], []
        self.is_best_A = self.is_best_B = False

    def set_input(self, input):
        self.real_A = self._to_device_act(input['A'])
        self.real_B = self._to_device_act(input['B'])

    def set_single_input(self, input):
        self.real_A = self._to_device_act(input['A'])
        self.image_paths = input.get('A_paths', [])

    def forward(self):
        grad = torch.is_grad_enabled()
        self.fake_B = self.netG_A(self.real_A)
        self._fb_rec, self._fb_gan = ops.fanout(self.fake_B, 2) if grad else (self.fake_B, self.fake_B)
        self.rec_A = self.netG_B(self._fb_rec)
        self.fake_A = self.netG_B(self.real_B)
        self._fa_rec, self._fa_gan = ops.fanout(self.fake_A, 2) if grad else (self.fake_A, self.fake_A)
        self.rec_B = self.netG_A(self._fa_rec)

    def backward_D_basic(self, netD, real, fake):
        loss_D_real = self.criterionGAN(netD(real), True)
        loss_D_fake = self.criterionGAN(netD(fake.detach()), False)
        self.backward_terms([(0.5, loss_D_real), (0.5, loss_D_fake)])
        return LossValue([(0.5, loss_D_real), (0.5, loss_D_fake)])

    def backward_D_A(self):
        fake_B = self.fake_B_pool.query(self.fake_B)
        self.loss_D_A = self.backward_D_basic(self.netD_A, self.real_B, fake_B)

    def backward_D_B(self):
        fake_A = self.fake_A_pool.query(self.fake_A)
        self.loss_D_B = self.backward_D_basic(self.netD_B, self.real_A, fake_A)

    def backward_G(self):
        lambda_idt, lambda_A, lambda_B = self.opt.lambda_identity, self.opt.lambda_A, self.opt.lambda_B
        terms = []
        if lambda_idt > 0:
            self.idt_A = self.netG_A(self.real_B)
            idt_a = self.criterionIdt(self.idt_A, self.real_B)
            self.idt_B = self.netG_B(self.real_A)
            idt_b = self.criterionIdt(self.idt_B, self.real_A)
            self.loss_G_idt_A = LossValue([(lambda_B * lambda_idt, idt_a)])
            self.loss_G_idt_B = LossValue([(lambda_A * lambda_idt, idt_b)])
            terms += [(lambda_B * lambda_idt, idt_a), (lambda_A * lambda_idt, idt_b)]
        else:
            self.loss_G_idt_A = 0
            self.loss_G_idt_B = 0
        g_a = self.criterionGAN(self.netD_A(self._fb_gan), True)
        g_b = self.criterionGAN(self.netD_B(self._fa_gan), True)
        cyc_a = self.criterionCycle(self.rec_A, self.real_A)
        cyc_b = self.criterionCycle(self.rec_B, self.real_B)
        self.loss_G_A, self.loss_G_B = LossValue([(1.0, g_a)]), LossValue([(1.0, g_b)])
        self.loss_G_cycle_A, self.loss_G_cycle_B = LossValue([(lambda_A, cyc_a)]), LossValue([(lambda_B, cyc_b)])
        terms += [(1.0, g_a), (1.0, g_b), (lambda_A, cyc_a), (lambda_B, cyc_b)]
        self.loss_G = LossValue(terms)
        self.backward_terms(terms)

    def optimize_parameters(self, steps):
        self.forward()
        self.set_requires_grad([self.netD_A, self.netD_B], False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        if self.dp is not None:
            self.dp.reduce(self.optimizer_G)
        self.optimizer_G.step()
        self.set_requires_grad([self.netD_A, self.netD_B], True)
        self.optimizer_D.zero_grad()
        self.backward_D_A()
        self.backward_D_B()
        if self.dp is not None:
            self.dp.reduce(self.optimizer_D)
        self.optimizer_D.step()

    def test_single_side(self, direction):
        generator = getattr(self, 'netG_%s' % direction[0])
        with torch.no_grad():
            self.fake_B = generator(self.real_A)
