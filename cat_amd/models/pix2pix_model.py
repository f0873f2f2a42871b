"""Pix2PixModel: the teacher-training step of `train.py --model pix2pix` (reference models/pix2pix_model.py:19-207) on the gfx950
kernels -- generator forward, PatchGAN D step (0.5 * (fake + real)), G step (GAN + lambda_recon * L1 / L2), two Adam updates.
Same op set as the distillation step; LossValue / seeded backward keep torch arithmetic off the path."""
from .. import loss as closs
from .. import networks, ops
from ..distillers.base_inception_distiller import LossValue
from ..optim import FusedAdam
from .base_model import BaseModel


class Pix2PixModel(BaseModel):
    _FLAGS = [  # pix2pix_model.py:21-65
        ('--restore_G_path', dict(type=str, default=None)),
        ('--restore_D_path', dict(type=str, default=None)),
        ('--recon_loss_type', dict(type=str, default='l1', choices=['l1', 'l2', 'smooth_l1'])),
        ('--lambda_recon', dict(type=float, default=100)),
        ('--lambda_gan', dict(type=float, default=1)),
        ('--real_stat_path', dict(type=str, required=False, default=None)),
        ('--lambda_comp_cost', dict(type=float, default=0)),
        ('--comp_cost', dict(type=str, default='l1', choices=['l1'])),
        ('--l1_renorm', dict(action='store_true')),
    ]

    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        assert is_train
        for flag, kw in Pix2PixModel._FLAGS:
            parser.add_argument(flag, **kw)
        return parser

    def __init__(self, opt):
        assert opt.isTrain
        BaseModel.__init__(self, opt)
        self.loss_names = ['G_gan', 'G_recon', 'D_real', 'D_fake', 'G_comp_cost']
        self.visual_names = ['real_A', 'fake_B', 'real_B']
        self.model_names = ['G', 'D']
        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, opt.dropout_rate, opt.init_type,
                                      opt.init_gain, self._dev_ids, opt=opt)
        self.netD = networks.define_D(opt.input_nc + opt.output_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type,
                                      opt.init_gain, self._dev_ids, opt=opt)
        self.criterionGAN = closs.GANLoss(opt.gan_mode)
        if opt.recon_loss_type == 'l1':
            self.criterionRecon = closs.L1Loss()
        elif opt.recon_loss_type == 'l2':
            self.criterionRecon = closs.MSELoss()
        else:
            raise NotImplementedError('Unknown reconstruction loss type [%s]!' % opt.recon_loss_type)
        self.optimizer_G = FusedAdam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizer_D = FusedAdam(self.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizers = [self.optimizer_G, self.optimizer_D]
        self.best_fid, self.best_mIoU = 1e9, -1e9
        self.fids, self.mIoUs = [], []
        self.is_best = False

    def set_input(self, input):
        AtoB = self.opt.direction == 'AtoB'
        self.real_A = self._to_device_act(input['A' if AtoB else 'B'])
        self.real_B = self._to_device_act(input['B' if AtoB else 'A'])
        self.image_paths = input.get('A_paths' if AtoB else 'B_paths', [])

    def forward(self):
        self.fake_B = self.netG(self.real_A)

    def backward_D(self):
        import torch
        with torch.no_grad():
            fake_AB = ops.Concat2Fn.apply(self.real_A, self.fake_B.detach())
            real_AB = ops.Concat2Fn.apply(self.real_A, self.real_B)
        self.loss_D_fake = self.criterionGAN(self.netD(fake_AB), False, for_discriminator=True)
        self.loss_D_real = self.criterionGAN(self.netD(real_AB), True, for_discriminator=True)
        self.loss_D = LossValue([(0.5, self.loss_D_fake), (0.5, self.loss_D_real)])
        self.backward_terms([(0.5, self.loss_D_fake), (0.5, self.loss_D_real)])

    def backward_G(self):
        opt = self.opt
        f_gan, f_recon = ops.fanout(self.fake_B, 2)
        gan = self.criterionGAN(self.netD(ops.Concat2Fn.apply(self.real_A, f_gan)), True, for_discriminator=False)
        recon = self.criterionRecon(f_recon, self.real_B)
        self.loss_G_gan = LossValue([(opt.lambda_gan, gan)])
        self.loss_G_recon = LossValue([(opt.lambda_recon, recon)])
        self.loss_G = self.loss_G_gan + self.loss_G_recon
        if getattr(opt, 'lambda_comp_cost', 0) > 0:
            # pix2pix_model.py:186-195: the term is netG.get_comp_cost(...) * lambda, and NO generator of the reference defines
            # get_comp_cost -- its getattr default returns 0, so the flag only adds a zero `G_comp_cost` entry to the loss report
            cost = getattr(self.netG, 'get_comp_cost', lambda p, renorm: 0)(p=int(opt.comp_cost[-1]), renorm=getattr(opt, 'l1_renorm', False))
            if cost != 0:
                raise NotImplementedError('a generator with get_comp_cost: the cost term has no kernel path')
            self.loss_G_comp_cost = 0.0 * opt.lambda_comp_cost
        self.backward_terms([(opt.lambda_gan, gan), (opt.lambda_recon, recon)])

    def optimize_parameters(self, steps):
        self.forward()
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        if self.dp is not None:
            self.dp.reduce(self.optimizer_D)
        self.optimizer_D.step()
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        if self.dp is not None:
            self.dp.reduce(self.optimizer_G)
        self.optimizer_G.step()
