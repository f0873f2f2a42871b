"""SPADEDistiller (reference distillers/spade_distiller.py:21-96): the GauGAN distiller `create_distiller(opt)` resolves for
`--distiller spade`; flags and defaults as in the reference, the step itself lives in BaseSPADEDistiller."""
from ..discriminators import MultiscaleDiscriminator
from .base_spade_distiller import BaseSPADEDistiller


class SPADEDistiller(BaseSPADEDistiller):
    _FLAGS = [  # spade_distiller.py:24-70
        ('--restore_pretrained_G_path', dict(type=str, default=None)),
        ('--pretrained_student_G_path', dict(type=str, default=None)),
        ('--pretrained_netG', dict(type=str, default='mobile_spade', choices=['inception_spade'])),
        ('--pretrained_ngf', dict(type=int, default=64)),
        ('--pretrained_norm_G', dict(type=str, default='spadesyncbatch3x3')),
        ('--target_flops', dict(type=float, default=0)),
        ('--prune_cin_lb', dict(type=int, default=1)),
        ('--prune_only', dict(action='store_true')),
        ('--prune_continue', dict(action='store_true')),
        ('--prune_logging_verbose', dict(action='store_true')),
    ]

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser = BaseSPADEDistiller.modify_commandline_options(parser, is_train)
        for flag, kw in SPADEDistiller._FLAGS:
            parser.add_argument(flag, **kw)
        parser.set_defaults(netD='multi_scale', dataset_mode='cityscapes', batch_size=16, print_freq=50,
                            save_latest_freq=10000000000, save_epoch_freq=10, nepochs=100, nepochs_decay=100, init_type='xavier',
                            teacher_ngf=64, student_ngf=48)
        # networks.modify_commandline_options (models/networks.py) -> the multiscale discriminator's flags
        parser = MultiscaleDiscriminator.modify_commandline_options(parser, is_train)
        return parser

    def __init__(self, opt):
        super(SPADEDistiller, self).__init__(opt)
