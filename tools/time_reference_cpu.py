"""Time the REFERENCE's own InceptionDistiller.optimize_parameters on this container's host cores, next to the oracle port that bench.py
times on the GPU box (`cpu_baseline.kind = "port"`; the reference itself cannot travel).  BASELINE config C2: pix2pix,
BatchNorm, hinge, ndf 128, student pruned to 4.6e9 MACs, 256x256, batch 2.  Build container only:

  python tools/time_reference_cpu.py [--batch 2] [--reps 3]"""
import argparse
import contextlib
import io
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the reference import recipe)
import torch  # noqa: E402

from oracle import detfill, ref_cpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    threads = os.cpu_count()
    torch.set_num_threads(threads)
    opt = G.ref_import.make_opt(norm='batch', track=True, target_flops=4.6e9, dataset_mode='aligned', gan_mode='hinge', ndf=128,
                                lambda_recon=100.0, lambda_distill=1.3, distill_G_loss_type='ka')
    T = G.teacher(opt)
    D = G.networks.define_D(6, 128, 'n_layers', 3, 'batch', 'normal', 0.02, [], opt=opt)
    D.load_state_dict(detfill.fill_state_dict(D.state_dict(), G.SEED_D))
    m = G.ref_distiller(opt, T, D)
    G.model_profiling(T, 256, 256, use_cuda=False, num_forwards=0, verbose=False)
    with contextlib.redirect_stdout(io.StringIO()):
        G.uc.shrink_model(m, 4.6e9, opt)
    S = m.netG_student
    S.load_state_dict(detfill.fill_state_dict(S.state_dict(), G.SEED_S))
    S.train()
    D.train()
    A = detfill.images((a.batch, 3, 256, 256), 1)
    B = detfill.images((a.batch, 3, 256, 256), 2)
    cpu = lambda net: {k: v.detach().clone() for k, v in net.state_dict().items()}
    ncfg = {'norm': 'batch', 'eps': 1e-5, 'momentum': 0.1}
    st = ref_cpu.DistillState(cpu(T), cpu(S), cpu(D), dict(T=ncfg, S=ncfg, D=ncfg, dataset_mode='aligned', gan_mode='hinge', lambda_recon=100.0,
                                                          lambda_distill=1.3, lambda_gan=1.0, lr=opt.lr, beta1=opt.beta1))

    def ref_step(i):
        m.set_input({'A': A, 'B': B, 'A_paths': [], 'B_paths': []})
        m.optimize_parameters(i)

    def port_step(i):
        ref_cpu.distill_step(st, A, B)

    for name, fn in (('reference InceptionDistiller.optimize_parameters', ref_step), ('oracle/ref_cpu.distill_step (port)', port_step)):
        fn(0)
        t0 = time.perf_counter()
        for i in range(a.reps):
            fn(1 + i)
        dt = (time.perf_counter() - t0) / a.reps
        print(f'{name:52s} {dt:7.2f} s/step  {a.batch / dt:6.3f} images/s  (batch {a.batch} @ 256x256, {threads} torch threads)')


if __name__ == '__main__':
    main()
