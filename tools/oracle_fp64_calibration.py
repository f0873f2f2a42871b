"""How far is the CPU oracle's OWN fp32 gradient from the exact (fp64) gradient of the same step?  Calibration for the network-level
gradient parity bars (tests/test_spade_gpu.py::check_grads): the C2 headline step (256 x 256, batch 2, pruned 4.6e9-MAC student,
ndf-128 PatchGAN, hinge + 100 L1 + 1.3 KA) is run twice on the host -- fp32 and fp64, same weights and images -- and the per-tensor
deviation of the student's parameter gradients is printed in the units check_grads uses (max |d| / max(own max, 3 % of the global max)).
Runs on CPU only (no GPU, no reference import):   python tools/oracle_fp64_calibration.py [size] [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


CONFIGS = {      # bench.py's C2 / C3 (BASELINE configs[1] / configs[2] per GPU)
    'c2': dict(opt=dict(norm='batch', track=True, ndf=128, dataset_mode='aligned', gan_mode='hinge', lambda_recon=100.0, lambda_distill=1.3),
               target_flops=4.6e9, d_in=6),
    'c3': dict(opt=dict(norm='instance', track=False, ndf=64, dataset_mode='unaligned', gan_mode='lsgan', lambda_recon=5.0, lambda_distill=1.0),
               target_flops=2.6e9, d_in=3),
}


def state_dicts(which='c2'):
    import copy
    from cat_amd import networks, prune, synthetic
    c = CONFIGS[which]
    opt = synthetic.default_options(**c['opt'], target_flops=c['target_flops'], prune_cin_lb=16, student_ngf=32, gpu_ids=[])
    torch.manual_seed(233)
    norm = c['opt']['norm']
    T = networks.define_G(3, 3, 64, 'inception_9blocks', norm, 0, 'normal', 0.02, [], opt=opt)
    T.load_state_dict(synthetic.fill_state_dict(T.state_dict(), synthetic.SEED_TEACHER, gamma_abs_normal=True))
    T.eval()
    thr, _ = prune.search_threshold(T, c['target_flops'], opt)
    S = copy.deepcopy(T)
    prune._apply_structure(S, T, thr, opt, copy_weights=True)
    S = networks.init_net(S, 'normal', 0.02, [])
    D = networks.define_D(c['d_in'], c['opt']['ndf'], 'n_layers', 3, norm, 'normal', 0.02, [], opt=opt)
    cpu = lambda net: {k: v.detach().clone() for k, v in net.state_dict().items()}
    return opt, cpu(T), cpu(S), cpu(D)


def spade_state_dicts(size=256, target_flops=5.6e9):
    """bench.py's GauGAN model (BASELINE configs[3]) built on the host: teacher ngf 64 with the canonical fill, the student architecture
    shrink_spade_model searches for `target_flops` (fresh initialisation, as the reference leaves it), multiscale SN-PatchGAN ndf 64, VGG19 with
    random weights of the real topology.  Returns (opt, cfg for oracle/ref_spade_cpu.SpadeState, teacher, student, D, VGG state dicts)."""
    import argparse
    from cat_amd import prune, synthetic
    from cat_amd.spade_modules import SPADEDistillerModules
    opt = synthetic.default_options(norm='instance', gpu_ids=[])
    opt.__dict__.update(dict(
        distiller='spade', input_nc=35, output_nc=3, semantic_nc=36, contain_dontcare_label=False, no_instance=False,
        teacher_ngf=64, student_ngf=48, pretrained_ngf=64, teacher_netG='inception_spade', student_netG='inception_spade',
        pretrained_netG='inception_spade', teacher_norm_G='spadesyncbatch3x3', student_norm_G='spadesyncbatch3x3',
        pretrained_norm_G='spadesyncbatch3x3', num_upsampling_layers='more', crop_size=2 * size, aspect_ratio=2.0,
        netD='multi_scale', ndf=64, n_layers_D=4, num_D=2, norm_D='spectralinstance', init_type='xavier', init_gain=0.02,
        gan_mode='hinge', lambda_gan=1.0, lambda_feat=10.0, lambda_vgg=10.0, lambda_distill=0.5, distill_G_loss_type='ka',
        no_TTUR=False, lr=2e-4, beta1=0.5, beta2=0.999, target_flops=target_flops, prune_cin_lb=16,
        data_height=256, data_width=512, data_channel=36, restore_pretrained_G_path=None))
    torch.manual_seed(233)
    m = SPADEDistillerModules(argparse.Namespace(**vars(opt)))
    m.netG_teacher.load_state_dict(synthetic.fill_state_dict(m.netG_teacher.state_dict(), synthetic.SEED_TEACHER_SPADE, gamma_abs_normal=True))
    _, _, student, _ = prune.spade_search(m.netG_teacher, target_flops, opt)
    cpu = lambda net: {k: v.detach().clone() for k, v in net.state_dict().items()}
    vsd = {k.split('.', 1)[1]: v for k, v in cpu(m.criterionVGG.vgg).items()}
    cfg = dict(G=dict(crop_size=opt.crop_size, aspect_ratio=opt.aspect_ratio, num_upsampling_layers=opt.num_upsampling_layers), num_D=2, n_layers_D=4,
               lambda_gan=1.0, lambda_feat=10.0, lambda_vgg=10.0, lambda_distill=0.5, lr=opt.lr, beta1=0.5, beta2=0.999, no_TTUR=False)
    return opt, cfg, cpu(m.netG_teacher), cpu(student), cpu(m.netD), vsd


def c2_state_dicts():
    return state_dicts('c2')


def oracle_cfg(opt):
    ncfg = {'norm': opt.norm, 'eps': opt.norm_epsilon, 'momentum': opt.norm_momentum}
    return dict(T=ncfg, S=ncfg, D=ncfg, dataset_mode=opt.dataset_mode, gan_mode=opt.gan_mode, lambda_recon=opt.lambda_recon,
                lambda_distill=opt.lambda_distill, lambda_gan=1.0, lr=opt.lr, beta1=opt.beta1)


def to64(sd):
    return {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}


def main():
    from oracle import detfill, ref_cpu
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    opt, T, S, D = c2_state_dicts()
    ncfg = {'norm': 'batch', 'eps': opt.norm_epsilon, 'momentum': opt.norm_momentum}
    cfg = dict(T=ncfg, S=ncfg, D=ncfg, dataset_mode='aligned', gan_mode='hinge', lambda_recon=100.0, lambda_distill=1.3, lambda_gan=1.0, lr=opt.lr,
               beta1=opt.beta1)
    A, B = detfill.images((nb, 3, size, size), 71), detfill.images((nb, 3, size, size), 72)
    st32 = ref_cpu.DistillState(T, S, D, cfg)
    ref_cpu.distill_step(st32, A, B)
    st64 = ref_cpu.DistillState(to64(T), to64(S), to64(D), cfg)
    ref_cpu.distill_step(st64, A.double(), B.double())
    g32, g64 = st32.grads_S, st64.grads_S
    gmax = max(float(v.abs().max()) for v in g64.values())
    rows = []
    for k, v in g64.items():
        own = float(v.abs().max())
        err = float((g32[k].double() - v).abs().max())
        rows.append((err / max(own, 3e-2 * gmax), err / gmax, k))
    rows.sort(reverse=True)
    rel = np.array([r[0] for r in rows])
    print('oracle fp32 vs fp64, student gradients, %d tensors @%dx%d batch %d' % (len(rows), size, size, nb))
    print('  worst err/gmax %.2e   median rel %.2e   90%% quantile %.2e   tensors within 1e-3: %.1f %%' %
          (max(r[1] for r in rows), float(np.median(rel)), float(np.quantile(rel, 0.9)), 100 * float((rel < 1e-3).mean())))
    for r in rows[:8]:
        print('  %.2e  (%.2e of gmax)  %s' % r)
    for k in ('G_gan', 'G_recon', 'G_distill'):
        print('  loss %-10s fp32 %.8f  fp64 %.8f' % (k, st32.losses[k], st64.losses[k]))


if __name__ == '__main__':
    main()
