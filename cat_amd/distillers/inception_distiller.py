"""InceptionDistiller: the distillation step of distillers/inception_distiller.py:100-188 on the MI355X kernels.

Data parallelism is process-per-GPU (cat_amd.parallel): each rank runs this class on its shard; the gradient
buckets are all-reduced over RCCL and the loss seeds reproduce nn.DataParallel's semantics: recon / GAN
terms are means over the global batch, the KA term is the SUM of per-shard KAs."""
import torch
from torch import nn

from .. import ops
from ..loss import KA
from ..prune import model_profiling
from .base_inception_distiller import BaseInceptionDistiller, LossValue


class InceptionDistiller(BaseInceptionDistiller):
    _FLAGS = [  # inception_distiller.py:39-69
        ('--restore_pretrained_G_path', dict(type=str, default=None)),
        ('--pretrained_netG', dict(type=str, default='inception_9blocks', choices=['inception_9blocks'])),
        ('--pretrained_ngf', dict(type=int, default=64)),
        ('--target_flops', dict(type=float, default=0)),
        ('--prune_cin_lb', dict(type=int, default=0)),
        ('--pretrained_student_G_path', dict(type=str, default=None)),
        ('--prune_only', dict(action='store_true')),
        ('--prune_continue', dict(action='store_true')),
        ('--prune_logging_verbose', dict(action='store_true')),
    ]

    @staticmethod
    def modify_commandline_options(parser, is_train):
        assert is_train
        parser = BaseInceptionDistiller.modify_commandline_options(parser, is_train)
        for flag, kw in InceptionDistiller._FLAGS:
            parser.add_argument(flag, **kw)
        parser.set_defaults(norm='instance', dataset_mode='aligned', log_dir='logs/inception',
                            teacher_netG='inception_9blocks', student_netG='inception_9blocks')
        return parser

    def __init__(self, opt):
        assert opt.isTrain
        super(InceptionDistiller, self).__init__(opt)
        from ..loss import MSELoss
        self.criterionMSE = MSELoss()
        self.best_fid = 1e9
        self.best_mIoU = -1e9
        self.fids, self.mIoUs = [], []
        h, w = getattr(opt, 'data_height', 256), getattr(opt, 'data_width', 256)
        model_profiling(self.netG_teacher, h, w, channel=getattr(opt, 'data_channel', 3))
        model_profiling(self.netG_student, h, w, channel=getattr(opt, 'data_channel', 3))

    def load_networks(self, verbose=True, teacher_only=False, restore_pretrain=True):
        """inception_distiller.py:190-203: a wider pretrained generator seeds the student by top-k L1 channel selection
        (cat_amd.weight_transfer, host-side, once) before the regular checkpoints are restored."""
        opt = self.opt
        if getattr(opt, 'restore_pretrained_G_path', None) is not None and restore_pretrain:
            from .. import networks
            from ..weight_transfer import load_pretrained_weight
            pre = networks.define_G(opt.input_nc, opt.output_nc, opt.pretrained_ngf, opt.pretrained_netG, opt.norm, 0, opt.init_type,
                                    opt.init_gain, [], opt=opt)
            self._load(pre, opt.restore_pretrained_G_path, verbose)
            load_pretrained_weight(opt.pretrained_netG, opt.student_netG, pre, self.netG_student, opt.pretrained_ngf, opt.student_ngf)
            del pre
        super(InceptionDistiller, self).load_networks(verbose, teacher_only=teacher_only, restore_pretrain=restore_pretrain)

    def forward(self, teacher_forward=True):
        if teacher_forward:
            with torch.no_grad():
                self.Tfake_B = self.netG_teacher(self.real_A)
        self.Sfake_B = self.netG_student(self.real_A)

    def calc_distill_loss(self):
        """sum_i -KA(Sact_i, Tact_i) (inception_distiller.py:106-157, 'ka' branch).  Returns the per-layer terms; the
        weighted total is a LossValue (no torch arithmetic on the hot path)."""
        kind = self.opt.distill_G_loss_type
        if kind not in ('ka', 'mse'):
            raise NotImplementedError(kind)
        terms = []
        for i, netA in enumerate(self.netAs):
            assert isinstance(netA, nn.Conv2d)
            n = self.mapping_layers[i]
            key = n + str(self.device)
            if kind == 'ka':
                term = KA(self.Sacts[key], self.Tacts[key])
                setattr(self, 'loss_G_distill%d' % i, LossValue([(-1.0, term)]))
            else:       # 'mse' (:113-132): the student activation goes through the 1x1 adaptor netA first
                term = self.criterionMSE(netA(self.Sacts[key]), self.Tacts[key])
                setattr(self, 'loss_G_distill%d' % i, LossValue([(1.0, term)]))
            terms.append(term)
        return terms

    def backward_G(self, steps):
        opt = self.opt
        ws = self.dp.world_size if self.dp is not None else 1
        sf_recon, sf_gan = ops.fanout(self.Sfake_B, 2)
        if opt.dataset_mode == 'aligned':
            recon = self.criterionRecon(sf_recon, self.real_B)
            fake = ops.Concat2Fn.apply(self.real_A, sf_gan)
        else:
            recon = self.criterionRecon(sf_recon, self.Tfake_B)
            fake = sf_gan
        pred_fake = self.netD(fake)
        gan = self.criterionGAN(pred_fake, True, for_discriminator=False)
        self.loss_G_recon = LossValue([(opt.lambda_recon, recon)])
        self.loss_G_gan = LossValue([(opt.lambda_gan, gan)])
        terms, seeds = [gan, recon], [self.seed(opt.lambda_gan), self.seed(opt.lambda_recon)]
        if opt.lambda_distill > 0:
            kas = self.calc_distill_loss()
            sign = -1.0 if opt.distill_G_loss_type == 'ka' else 1.0
            self.loss_G_distill = LossValue([(sign * opt.lambda_distill, k) for k in kas])
            terms += kas
            # DataParallel sums the per-shard distillation terms (KA, or the per-device MSE) while every other term is a mean over
            # the gathered batch: with gradient AVERAGING across ranks their seed therefore carries a factor world_size
            seeds += [self.seed(sign * opt.lambda_distill * ws)] * len(kas)
        else:
            self.loss_G_distill = 0
        self.loss_G = self.loss_G_gan + self.loss_G_recon + self.loss_G_distill
        torch.autograd.backward(terms, seeds)
        ops.sync_side_streams()

    def optimize_parameters(self, steps):
        """forward -> D step -> G step (inception_distiller.py:179-188).  Optionally (`teacher_side_stream`) the frozen teacher's
        forward -- which depends on no trainable state -- runs on a side stream while the main stream does the student forward and
        the discriminator step; the streams join before backward_G reads the teacher's activations.  Off by default on one GPU;
        the data-parallel schedule (_optimize_parameters_dp) uses it to hide the gradient all-reduces."""
        if self.dp is not None and getattr(self, 'dp_overlap', True):
            return self._optimize_parameters_dp(steps)
        overlap = getattr(self, 'teacher_side_stream', False)   # measured on 1 GPU: -2 % (two MFMA-bound streams only contend)
        if overlap:
            main = torch.cuda.current_stream(self.device)
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            side = self._side_stream
            side.wait_stream(main)
            self.real_A.record_stream(side)
            with torch.cuda.stream(side), torch.no_grad():
                self.Tfake_B = self.netG_teacher(self.real_A)
            t_done = side.record_event()
            self.Sfake_B = self.netG_student(self.real_A)
        else:
            self.forward()
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        if self.dp is not None:
            self.dp.reduce(self.optimizer_D)
        self.optimizer_D.step()
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        if overlap:
            main.wait_event(t_done)
            for t in [self.Tfake_B] + list(self.Tacts.values()):
                t.record_stream(main)
        self.backward_G(steps)
        if self.dp is not None:
            self.dp.reduce(self.optimizer_G)
        self.optimizer_G.step()

    # -- data-parallel schedule ---------------------------------------------------------------------------------
    def enable_data_parallel(self, reducer, overlap=True):
        """Attach a cat_amd.parallel.DataParallelReducer: replicas are synchronised once, then only gradients move."""
        self.dp = reducer
        self.dp_overlap = overlap
        self._pending_G = None
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        reducer.broadcast_parameters([self.netG_teacher, self.netG_student, self.netD] + list(self.netAs))

    def finish_pending(self):
        """Complete a deferred student update (all-reduce wait + Adam G).  Called before anything reads the weights."""
        if getattr(self, '_pending_G', None) is not None:
            self._pending_G.wait()
            self.optimizer_G.step()
            self._pending_G = None

    def _optimize_parameters_dp(self, steps):
        """Same arithmetic as optimize_parameters, scheduled for xGMI overlap:
          side stream : frozen-teacher forward of THIS batch
          main stream : [wait G all-reduce of the previous step -> Adam G] -> student forward -> D step (D bucket
                        all-reduce, Adam D) -> backward_G -> launch G bucket all-reduce (awaited next call).
        The three compute pieces (_dp_teacher / _dp_first / _dp_second) are what cat_amd.graph.GraphedDPStep captures as hipGraph
        segments; the collectives and the one-launch Adam G between them stay eager."""
        main = torch.cuda.current_stream(self.device)
        side = self._side_stream
        side.wait_stream(main)
        self.real_A.record_stream(side)
        with torch.cuda.stream(side):
            self._dp_teacher()
        t_done = side.record_event()
        self.finish_pending()
        pend = []
        for run, (lo, hi) in self._dp_first_stages():      # each slice of the D bucket leaves as soon as its stage has been issued
            run()
            pend.append(self.dp.reduce_slice_async(self.optimizer_D, lo, hi) if lo is not None else self.dp.reduce_async(self.optimizer_D))
        for w in pend:
            w.wait()
        main.wait_event(t_done)
        for t in [self.Tfake_B] + list(self.Tacts.values()):
            t.record_stream(main)
        self._dp_second(steps)
        self._pending_G = self.dp.reduce_async(self.optimizer_G)

    def _dp_teacher(self):
        with torch.no_grad():
            self.Tfake_B = self.netG_teacher(self.real_A)

    def _dp_first(self):
        """Student forward + the discriminator's backward pass: everything up to the D-bucket all-reduce."""
        self.Sfake_B = self.netG_student(self.real_A)
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()

    def _dp_first_stages(self):
        """_dp_first cut where slices of the discriminator's gradient bucket become final: [(callable, (lo, hi))], the float range of
        optimizer_D's flat gradient buffer to all-reduce after each callable -- the 512 -> 1024 layer's 8.4 M of 11 M parameters are
        ready first and travel while the remaining two thirds of backward_D run (SURVEY 8e overlap 2).  A discriminator that cannot be
        cut (d_stage_plan() is None) is one stage with the whole bucket, (None, None)."""
        if getattr(self, 'dp_sliced_D', True) and self.d_stage_plan() is not None:
            def head():
                self.Sfake_B = self.netG_student(self.real_A)
                self.set_requires_grad(self.netD, True)
                self.optimizer_D.zero_grad()
            stages, slices = self.backward_D_stages()

            def first():
                head()
                stages[0]()
            return [(first, slices[0]), (stages[1], slices[1]), (stages[2], slices[2])]
        return [(self._dp_first, (None, None))]

    def _dp_second(self, steps):
        """Adam D on the reduced bucket + the generator's backward pass: everything up to the G-bucket all-reduce."""
        self.optimizer_D.step()
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.backward_G(steps)
