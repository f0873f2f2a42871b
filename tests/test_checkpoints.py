"""CPU: the checkpoint / scheduler surface Trainer uses: file names and state_dict keys of save_networks, the restore_* paths of
load_networks, the per-epoch linear-decay schedule.  The distiller is assembled with __new__ (its constructor needs a GPU); the networks are
the product's own modules on the CPU -- state_dict I/O and schedulers involve no kernels."""
import os

import torch

import helpers as H
from cat_amd import networks, nn as cnn
from cat_amd.distillers.inception_distiller import InceptionDistiller
from cat_amd.optim import FusedAdam


def _stub(tmp_path, seed, **kw):
    opt = H.make_opt(norm='batch', track=True, ndf=16, log_dir=str(tmp_path), nepochs=2, nepochs_decay=3, **kw)
    torch.manual_seed(seed)
    m = InceptionDistiller.__new__(InceptionDistiller)
    m.opt = opt
    m.save_dir = os.path.join(str(tmp_path), 'checkpoints')
    m.model_names = ['netG_student', 'netG_teacher', 'netD']
    m.netG_teacher = networks.define_G(3, 3, 16, 'inception_9blocks', 'batch', 0, 'normal', 0.02, [], opt=opt)
    m.netG_student = networks.define_G(3, 3, 8, 'inception_9blocks', 'batch', 0, 'normal', 0.02, [], opt=opt)
    m.netD = networks.define_D(6, 16, 'n_layers', 3, 'batch', 'normal', 0.02, [], opt=opt)
    m.netAs = [cnn.Conv2d(32, 64, 1) for _ in range(4)]
    import itertools
    m.optimizer_G = FusedAdam([{'params': m.netG_student.parameters()}, {'params': itertools.chain(*[a.parameters() for a in m.netAs])}],
                              lr=opt.lr, betas=(opt.beta1, 0.999))
    m.optimizer_D = FusedAdam(m.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
    m.optimizers = [m.optimizer_G, m.optimizer_D]
    m.Tacts, m.Sacts, m.mapping_layers = {}, {}, []
    return m, opt


def test_save_and_restore_round_trip(tmp_path, capsys):
    a, opt = _stub(tmp_path, 1)
    a.save_networks('latest')
    files = sorted(os.listdir(a.save_dir))
    assert files == ['latest_net_A-0.pth', 'latest_net_A-1.pth', 'latest_net_A-2.pth', 'latest_net_A-3.pth', 'latest_net_D.pth', 'latest_net_G.pth',
                     'latest_optim-0.pth', 'latest_optim-1.pth']                              # base_inception_distiller.py:367-396
    sd = torch.load(os.path.join(a.save_dir, 'latest_net_G.pth'))
    assert list(sd) == list(a.netG_student.state_dict()) and all(v.is_contiguous() and v.device.type == 'cpu' for v in sd.values())
    assert sd['down_sampling.1.weight'].shape == (8, 3, 7, 7)                                  # logical OIHW on the wire
    b, optb = _stub(tmp_path / 'b', 2, restore_student_G_path=os.path.join(a.save_dir, 'latest_net_G.pth'),
                    restore_D_path=os.path.join(a.save_dir, 'latest_net_D.pth'), restore_A_path=os.path.join(a.save_dir, 'latest_net_A'))
    # (restore_O_path re-houses the Adam moments in the flat HBM buffers: FusedAdam.load_state_dict is GPU code)
    assert not torch.equal(b.netD.state_dict()['model.0.weight'], a.netD.state_dict()['model.0.weight'])
    b.load_networks(verbose=True)
    assert 'Load network at' in capsys.readouterr().out
    for na, nb in ((a.netG_student, b.netG_student), (a.netD, b.netD), (a.netAs[3], b.netAs[3])):
        for (k, va), (_, vb) in zip(na.state_dict().items(), nb.state_dict().items()):
            assert torch.equal(va, vb), k
    osd = torch.load(os.path.join(a.save_dir, 'latest_optim-0.pth'))
    assert [len(g['params']) for g in osd['param_groups']] == [len(list(a.netG_student.parameters())), 8]   # torch.optim.Adam layout


def test_linear_schedule_and_print(tmp_path, capsys):
    m, opt = _stub(tmp_path, 3)
    m.schedulers = [networks.get_scheduler(o, opt) for o in m.optimizers]
    lrs = []
    for _ in range(6):
        m.update_learning_rate()
        lrs.append(m.optimizers[0].param_groups[0]['lr'])
    # reference networks.py:80-87 with nepochs 2, nepochs_decay 3: factor(epoch) = 1 - max(0, epoch + 1 - 2) / 4
    want = [opt.lr * (1.0 - max(0, e + 1 - 2) / 4.0) for e in range(1, 7)]
    assert all(abs(a - b) < 1e-12 for a, b in zip(lrs, want)), (lrs, want)
    assert m.optimizers[1].param_groups[0]['lr'] == lrs[-1] and m.optimizers[0].param_groups[1]['lr'] == lrs[-1]
    assert 'learning rate = ' in capsys.readouterr().out
    m.print_networks()
    out = capsys.readouterr().out
    assert '[Network netG_student] Total number of parameters' in out and '[Network netD]' in out
