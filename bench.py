#!/usr/bin/env python
"""bench.py — distill-step images/sec of the MI355X build on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (N>1)

Workload (BASELINE configs[1]): pix2pix InceptionDistiller.optimize_parameters at 256x256, per-GPU batch
16 -- frozen teacher ngf 64 (BatchNorm, running stats), student pruned by shrink_model to 4.6e9 MACs and re-initialised,
PatchGAN 6ch ndf 128, hinge GAN + L1*100 + KA*1.3, two Adam steps.  Synthetic N(0,1).tanh() images, random-init
weights of the named architecture, fp32 everywhere.  Data parallel = one process per GPU, batch sharded (weak scaling:
per-GPU batch fixed), RCCL all-reduce of the gradient buckets.

The JSON line carries `roofline` for the dominant kernel family (the fp32-MFMA implicit-GEMM convolutions; achieved =
algorithmic FLOPs / HIP-event time of those launches, measured in a short SERIAL pass after the timed region -- stream-level
concurrency off -- so the events neither perturb `value` nor see co-running kernels) and `cpu_baseline` (the CPU oracle timed on a bounded sample of the same workload on rank 0)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CUs @ 2.4 GHz


C2 = dict(norm='batch', track=True, ndf=128, dataset_mode='aligned', gan_mode='hinge', lambda_recon=100.0, lambda_distill=1.3)
C3 = dict(norm='instance', track=False, ndf=64, dataset_mode='unaligned', gan_mode='lsgan', lambda_recon=5.0, lambda_distill=1.0)


def build_model(args, device_index):
    """BASELINE configs[1]: canonical teacher, student pruned from it, PatchGAN, both optimizers.
    `--workload c3` = configs[2] per GPU: CycleGAN-style distillation (InstanceNorm, lsgan, unaligned, ndf 64, student S_2.6)."""
    from cat_amd import networks, prune, synthetic
    from cat_amd.distillers import create_distiller
    opt = synthetic.default_options(**(C3 if getattr(args, 'workload', 'c2') == 'c3' else C2), target_flops=args.target_flops, prune_cin_lb=16,
                                    student_ngf=32, gpu_ids=[device_index], data_height=args.size, data_width=args.size)
    torch.manual_seed(233)
    model = create_distiller(opt, verbose=False)
    # canonical teacher: deterministic weights, |N(0,1)| norm scales (a trained teacher's scales are non-uniform)
    model.netG_teacher.load_state_dict(synthetic.fill_state_dict(model.netG_teacher.state_dict(), synthetic.SEED_TEACHER, gamma_abs_normal=True))
    model.setup(opt, verbose=False)
    # student shapes are defined at 256x256 like the reference's launch scripts (data_height is the dataset's size)
    opt.data_height = opt.data_width = 256
    prune.shrink(model, opt)                                                              # trainer.py:106
    model.netG_student = networks.init_net(model.netG_student, opt.init_type, opt.init_gain, []).to(model.device)   # trainer.py:107-109
    model.remove_mapping_hook()
    import itertools
    from cat_amd.optim import FusedAdam
    gp = [a.parameters() for a in model.netAs]
    model.optimizer_G = FusedAdam([{'params': model.netG_student.parameters()}, {'params': itertools.chain(*gp)}], lr=opt.lr,
                                  betas=(opt.beta1, 0.999))
    model.optimizers = [model.optimizer_G, model.optimizer_D]
    model.add_mapping_hook()
    model.netG_student.train()
    model.netD.train()
    return model, opt


def build_spade_model(args, device_index):
    """BASELINE configs[3]: GauGAN SPADEDistiller.optimize_parameters at 512x256 (crop 512, aspect 2), per-GPU
    batch 4 -- frozen teacher ngf 64, student ngf 48 pruned by shrink_spade_model to 5.6e9 MACs (the launch script's budget),
    multiscale spectral-norm PatchGAN ndf 64, hinge + feature matching + VGG + KA.  VGG19 carries random weights of the real
    topology (no network to fetch torchvision's): same kernels, same FLOPs."""
    from cat_amd import prune, synthetic
    from cat_amd.distillers import create_distiller
    opt = synthetic.default_options(norm='instance', gpu_ids=[device_index])
    opt.__dict__.update(dict(
        distiller='spade', input_nc=35, output_nc=3, semantic_nc=36, contain_dontcare_label=False, no_instance=False,
        teacher_ngf=64, student_ngf=48, pretrained_ngf=64, teacher_netG='inception_spade', student_netG='inception_spade',
        pretrained_netG='inception_spade', teacher_norm_G='spadesyncbatch3x3', student_norm_G='spadesyncbatch3x3',
        pretrained_norm_G='spadesyncbatch3x3', num_upsampling_layers='more', crop_size=2 * args.size, aspect_ratio=2.0,
        netD='multi_scale', ndf=64, n_layers_D=4, num_D=2, norm_D='spectralinstance', init_type='xavier', init_gain=0.02,
        gan_mode='hinge', lambda_gan=1.0, lambda_feat=10.0, lambda_vgg=10.0, lambda_distill=0.5, distill_G_loss_type='ka',
        no_TTUR=False, lr=2e-4, beta1=0.5, beta2=0.999, target_flops=args.target_flops, prune_cin_lb=16,
        data_height=args.size, data_width=2 * args.size, data_channel=36, restore_pretrained_G_path=None))
    torch.manual_seed(233)
    model = create_distiller(opt, verbose=False)
    m = model.modules_on_one_gpu
    m.netG_teacher.load_state_dict(synthetic.fill_state_dict(m.netG_teacher.state_dict(), synthetic.SEED_TEACHER_SPADE, gamma_abs_normal=True))
    model.setup(opt, verbose=False)
    opt.data_height, opt.data_width = 256, 512          # student shapes are defined at the dataset's size (launch script)
    prune.shrink(model, opt)
    m.train()
    return model, opt


def spade_batches(args, rank, nbuf, device='cuda'):
    from cat_amd import synthetic
    h, w = args.size, 2 * args.size
    out = []
    for i in range(nbuf):
        lab, ins = synthetic.label_maps(args.batch, h, w, 3000 + 10 * i + rank)
        out.append({'label': lab.to(device), 'instance': ins.to(device),
                    'image': synthetic.images((args.batch, 3, h, w), 4000 + 10 * i + rank).to(device), 'path': []})
    return out


def cpu_baseline_spade(opt, model, args):
    """oracle/ref_spade_cpu.spade_step (a port: plain PyTorch ATen ops) on a bounded sample: batch 1, one timed step."""
    import numpy as np
    from oracle import detfill, ref_spade_cpu as R
    m = model.modules_on_one_gpu
    cpu = lambda net: {k: v.detach().cpu().contiguous().clone() for k, v in net.state_dict().items()}
    vsd = {k.split('.', 1)[1]: v for k, v in cpu(m.criterionVGG.vgg).items()}
    cfg = dict(G=dict(crop_size=opt.crop_size, aspect_ratio=opt.aspect_ratio, num_upsampling_layers=opt.num_upsampling_layers), num_D=2,
               n_layers_D=4, lambda_gan=1.0, lambda_feat=10.0, lambda_vgg=10.0, lambda_distill=0.5, lr=opt.lr, beta1=0.5, beta2=0.999,
               no_TTUR=False)
    st = R.SpadeState(cpu(m.netG_teacher), cpu(m.netG_student), cpu(m.netD), vsd, cfg)
    h, w = args.size, 2 * args.size
    rng = np.random.default_rng(5)
    lab = torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 35, (1, 1, h // 16, w // 16)), 16, 2), 16, 3))
    ins = torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 1000, (1, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32))
    sem = R.preprocess_input(lab, ins, 35)
    img = detfill.images((1, 3, h, w), 6)
    was = torch.get_num_threads()
    cores = physical_cores()
    torch.set_num_threads(cores)
    try:
        R.spade_step(st, sem, img)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            R.spade_step(st, sem, img)
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(was)
    dt = sorted(times)[1]
    return {'value': round(1 / dt, 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/ref_spade_cpu.spade_step, batch 1 @ {w}x{h}, 1 warm-up + median of 3 timed steps ({min(times):.1f}-{max(times):.1f} s), '
                      f'torch.set_num_threads({cores}) = physical cores of {os.cpu_count()} logical'}


# The reference itself cannot travel to the GPU box (it is imported only in the build container).  Timed side by side there
# (tools/time_reference_cpu.py: this C2 step at batch 2, 256x256, 8 threads, torch 2.10 CPU): the reference's own optimize_parameters
# 4.02 s/step = 0.50 images/s, this port 2.39 s/step = 0.84 images/s -- the port is the FASTER of the two (no module / hook overhead),
# so `value` flatters the CPU side by that ratio.
REFERENCE_VS_PORT = ('kind=port: the reference cannot travel to this box; side by side in the build container (tools/time_reference_cpu.py, batch 2, '
                     '8 threads) the reference runs 0.50 images/s and this port 0.84 images/s: reference / port = 0.59, i.e. the reference CPU '
                     'path would read ~0.59 x this value')


def physical_cores():
    """Physical cores of the host (time the CPU path on all PHYSICAL cores; 2 x SMT threads oversubscribe ATen's pools:
    the same port ran 0.30 images/s on 128 threads and 0.84 on 8)."""
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:      # noqa: BLE001
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def oracle_cfg(opt):
    ncfg = {'norm': opt.norm, 'eps': opt.norm_epsilon, 'momentum': opt.norm_momentum}
    return dict(T=ncfg, S=ncfg, D=ncfg, dataset_mode=opt.dataset_mode, gan_mode=opt.gan_mode, lambda_recon=opt.lambda_recon,
                lambda_distill=opt.lambda_distill, lambda_gan=1.0, lr=opt.lr, beta1=opt.beta1)


def cpu_baseline(opt, model, args):
    """The CPU oracle (a port: plain PyTorch ATen ops, same algorithm) on a bounded sample with SURVEY §8d's protocol: all physical
    cores, batch 4 at the bench resolution, 2 warm-up steps + the median of 5 timed steps (`--cpu-baseline-quick`: batch 2, 1 + 3)."""
    from oracle import detfill, ref_cpu
    nb, warm, reps = (2, 1, 3) if args.cpu_baseline_quick else (4, 2, 5)
    cpu = lambda net: {k: v.detach().cpu().contiguous().clone() for k, v in net.state_dict().items()}
    st = ref_cpu.DistillState(cpu(model.netG_teacher), cpu(model.netG_student), cpu(model.netD), oracle_cfg(opt))
    was = torch.get_num_threads()
    cores = physical_cores()
    torch.set_num_threads(cores)
    try:
        A = detfill.images((nb, 3, args.size, args.size), 1)
        B = detfill.images((nb, 3, args.size, args.size), 2)
        for _ in range(warm):
            ref_cpu.distill_step(st, A, B)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ref_cpu.distill_step(st, A, B)
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(was)
    dt = sorted(times)[reps // 2]
    return {'value': round(nb / dt, 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/ref_cpu.distill_step, batch {nb} @ {args.size}x{args.size}, {warm} warm-up + median of {reps} timed steps '
                      f'({min(times):.2f}-{max(times):.2f} s), torch.set_num_threads({cores}) = physical cores of {os.cpu_count()} logical. '
                      + REFERENCE_VS_PORT}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--sustained-steps', type=int, default=100, help='extra steps after the timed region (N=1): the clock-settled rate')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c3', 'spade'],
                    help='c2 = pix2pix InceptionDistiller (BASELINE configs[1], the headline metric); c3 = CycleGAN-style InceptionDistiller '
                         '(configs[2] per GPU: InstanceNorm, lsgan, unaligned, ndf 64, batch 8); spade = GauGAN SPADEDistiller '
                         '(configs[3], per-GPU batch 4 @ 512x256)')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: 16 for c2, 8 for c3, 4 for spade)')
    ap.add_argument('--cpu-baseline-quick', action='store_true', dest='cpu_baseline_quick',
                    help='cpu_baseline on batch 2, 1 warm-up + median of 3 (default: SURVEY §8d protocol, batch 4, 2 + 5)')
    ap.add_argument('--size', type=int, default=256, help='image height (spade: width = 2 * size)')
    ap.add_argument('--target-flops', type=float, default=None, dest='target_flops')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-overlap', action='store_true')
    ap.add_argument('--teacher-side-stream', type=int, default=0, help='c2: run the frozen teacher forward on a side stream (1 GPU)')
    ap.add_argument('--graph', type=int, default=1, help='1 (default): replay the captured step (one hipGraph on 1 GPU; hipGraph segments around '
                                                         'the two collectives with N > 1 ranks, c2 only); 0: eager launches')
    ap.add_argument('--segments', type=int, default=0,
                    help='1 GPU, c2 / c3: 1 = replay the step as hipGraph segments with the frozen teacher on a side stream (cat_amd.graph.GraphedDPStep '
                         'without a reducer: the launch mode of the N > 1 points, minus the collectives) instead of one graph.  Measured on one box '
                         '(profiles/r06_tconv_ab.txt sections 6 - 7): one graph on ONE stream 321.3, segments 320.1 images/s')
    ap.add_argument('--dp-schedule', type=int, default=0, dest='dp_schedule',
                    help='1 GPU only: run the DATA-PARALLEL schedule through a world_size-1 RCCL group (same launch mode as the N > 1 points '
                         'of a scaling curve: teacher on a side stream, bucket all-reduces, deferred Adam G)')
    ap.add_argument('--mfma', default='f32', choices=['f32', 'bf16x3'],
                    help='bf16x3: the opt-in split-bf16 backward tiles of the wide PatchGAN layers (csrc/conv_split.hip; DESIGN section 6). Reported with its '
                         'own dtype; the f32 line is the graded one')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the short GauGAN (configs[3], batch 4) and CycleGAN-style (configs[2], batch 8) measurements that the default '
                         '1-GPU headline run attaches as `secondary`')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run (one rank per GPU, RCCL over xGMI);
        # rank 0 of the child job prints the JSON line on our stdout
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))))
    # stdout carries exactly ONE line, the JSON record: everything the build prints (pruning search log, ...) goes to stderr
    # (also at the file-descriptor level: RCCL / gloo / HIP runtime messages are written by native code straight to fd 1)
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    spade = args.workload == 'spade'
    if args.batch is None:
        args.batch = 4 if spade else (8 if args.workload == 'c3' else 16)
    if args.target_flops is None:
        args.target_flops = 5.6e9 if spade else (2.6e9 if args.workload == 'c3' else 4.6e9)

    from cat_amd import _lib, ops, parallel
    _lib.load()
    if args.mfma == 'bf16x3':
        ops.set_mfma_split(True)
    rank, world, local = parallel.init_distributed()
    if world == 1 and args.dp_schedule:
        parallel.init_single_rank_group()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path is HIP-only')
    os.environ['LOCAL_RANK'] = str(local)
    torch.cuda.set_device(local)
    from cat_amd import synthetic
    model, opt = (build_spade_model if spade else build_model)(args, local)
    if args.no_overlap:
        model.teacher_side_stream = False
    elif args.teacher_side_stream and not spade:
        model.teacher_side_stream = True
    dp = world > 1 or bool(args.dp_schedule)
    if dp:
        if spade:
            model.enable_data_parallel(parallel.DataParallelReducer())
        else:
            model.enable_data_parallel(parallel.DataParallelReducer(), overlap=not args.no_overlap)

    # synthetic batches, resident in HBM before the timed region (global batch = batch * world; this rank's shard)
    nbuf = 4
    batches = []
    if spade:
        batches = spade_batches(args, rank, nbuf)
    for i in range(0 if spade else nbuf):
        A = synthetic.images((args.batch, 3, args.size, args.size), 1000 + 10 * i + rank).cuda()
        B = synthetic.images((args.batch, 3, args.size, args.size), 2000 + 10 * i + rank).cuda()
        batches.append({'A': A, 'B': B, 'A_paths': [], 'B_paths': []})

    def eager_step(i):
        model.set_input(batches[i % nbuf])
        model.optimize_parameters(i)

    step = eager_step
    graphed = None
    launch = 'eager'
    if args.graph and (dp and not spade and not args.no_overlap or not dp):
        from cat_amd.graph import GraphedDPStep, GraphedStep
        # (the frozen teacher on a side stream INSIDE one captured graph crashes the HIP runtime at capture: --teacher-side-stream with
        # --graph 1 therefore means the segment schedule, whose teacher graph is replayed on the side stream)
        seg = bool((dp or args.segments or args.teacher_side_stream) and not spade)
        for use_seg in ([True, False] if seg and not dp else [seg]):
            try:
                graphed = (GraphedDPStep if use_seg else GraphedStep)(model, batches[0])
                launch = ('hipGraph segments%s (teacher on a side stream | student fwd + D bwd | Adam D + G bwd)' % (' around the collectives' if dp else '')) \
                    if use_seg else 'hipGraph replay'
                break
            except Exception as e:      # a failed capture must not cost the measurement: fall back (one graph, then eager launches) and say so
                # (deterministic across ranks: every rank captures the same launch sequence, so all of them fall back together)
                print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back', file=sys.stderr, flush=True)
                graphed = None
                torch.cuda.synchronize()
        if graphed is not None:
            def step(i):
                graphed(batches[i % nbuf])

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if dp and hasattr(model, 'finish_pending'):
        model.finish_pending()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = model.dp.max_over_ranks(dt)
    losses = model.get_current_losses()
    assert all(v == v for v in losses.values()), 'NaN loss'
    assert ops.STATS['conform_copies'] == 0
    # sustained rate: `value` above is the contract's K steps (the driver asks for 20: ~1 s of GPU work, inside the boost window and below an
    # external utilisation sampler's cadence).  A second, longer run of the SAME step shows the clock-settled rate and keeps the GPU busy
    # for several seconds; reported beside `value`, never instead of it.
    sustained = None
    if world == 1 and args.sustained_steps > 0:
        barrier()
        t1 = time.perf_counter()
        for i in range(args.sustained_steps):
            step(args.warmup + args.steps + i)
        barrier()
        ds = time.perf_counter() - t1
        sustained = {'steps': args.sustained_steps, 'ms_per_step': round(1e3 * ds / args.sustained_steps, 3),
                     'value': round(args.batch * args.sustained_steps / ds, 3)}

    # the measurement passes run the step / the student forward again: with N > 1 ranks those contain collectives (gradient buckets,
    # SynchronizedBatchNorm statistics), so EVERY rank executes them; rank 0 reports
    roofline = student_fwd = None
    if not args.no_kernel_profile and (rank == 0 or world > 1):
        roofline = kernel_roofline(model, eager_step, args)
        if dp and hasattr(model, 'finish_pending'):
            model.finish_pending()
        student_fwd = student_forward_rate(model, batches[0], spade, graph=world == 1)
    if world > 1:
        torch.distributed.barrier()
    ips = args.batch * world * args.steps / dt
    if spade:
        metric = f'distill-step images/sec @{2 * args.size}x{args.size} bs={args.batch} (GauGAN SPADEDistiller)'
        workload = ('GauGAN SPADEDistiller.optimize_parameters (BASELINE configs[3]): teacher ngf64 frozen + student ngf48 pruned to '
                    f'{args.target_flops:.2g} MACs + multiscale SN-PatchGAN ndf64, hinge + feat + VGG + KA, TTUR Adam x2')
        image, n_macs = f'{2 * args.size}x{args.size}', int(model.modules_on_one_gpu.netG_student.n_macs)
    else:
        metric = f'distill-step images/sec @{args.size}x{args.size} bs={args.batch}'      # default: BASELINE.json's @256x256 bs=16
        if args.workload == 'c3':
            workload = ('CycleGAN-style InceptionDistiller.optimize_parameters (BASELINE configs[2], per GPU): teacher ngf64 frozen + student '
                        f'pruned to {args.target_flops:.2g} MACs + PatchGAN ndf64, InstanceNorm, lsgan + L1(teacher) + KA, Adam x2')
        else:
            workload = ('pix2pix InceptionDistiller.optimize_parameters (BASELINE configs[1]): teacher ngf64 frozen + student '
                        f'pruned to {args.target_flops:.2g} MACs + PatchGAN ndf128, hinge + L1 + KA, Adam x2')
        image, n_macs = f'{args.size}x{args.size}', int(model.netG_student.n_macs)
    out = {
        'metric': metric, 'value': round(ips, 3), 'unit': 'images/sec', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.mfma == 'f32' else 'bf16x3', 'data': 'synthetic',
        'config': {'workload': workload, 'image': image, 'per_gpu_batch': args.batch, 'global_batch': args.batch * world,
                   'parallelism': f'dp{world}', 'ranks': world,
                   'collectives': (f'{parallel.backend_version()} all-reduce of the flat gradient buckets' if dp else 'none'),
                   'schedule': 'data-parallel' if dp else 'single-GPU',
                   'student_n_macs': n_macs, 'launch': launch if graphed is not None else 'eager'},
        'roofline': roofline,
        'student_forward': student_fwd,
    }
    if sustained is not None:
        out['sustained'] = sustained
    if rank == 0:
        headline = world == 1 and args.workload == 'c2' and args.size == 256 and args.batch == 16 and not args.dp_schedule and args.mfma == 'f32'
        if headline and not args.no_secondary:
            out['secondary'] = secondary_measurements()
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = (cpu_baseline_spade if spade else cpu_baseline)(opt, model, args)
        print(json.dumps(out), file=json_out, flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


# child runs of the default 1-GPU headline invocation: (key, extra argv, what it is)
SECONDARY = (
    ('spade', ['--workload', 'spade'], 'GauGAN SPADEDistiller, BASELINE configs[3] per GPU: 512x256, batch 4'),
    ('c3', ['--workload', 'c3'], 'CycleGAN-style InceptionDistiller, BASELINE configs[2] per GPU: batch 8'),
    ('512x512', ['--size', '512'], 'the headline step at 512x512, batch 16 (north_star asks for both sizes)'),
    ('c5', ['--batch', '8', '--no-kernel-profile'], 'BASELINE configs[4] per GPU: the headline nets at per-GPU batch 8 (global 64 on 8 GPUs)'),
    ('dp1', ['--dp-schedule', '1', '--no-kernel-profile'],
     'the N = 1 point of the scaling curve in the DATA-PARALLEL launch mode: world_size-1 RCCL group, hipGraph segments around the two '
     'bucket all-reduces, teacher on a side stream, deferred Adam G'),
    ('bf16x3', ['--mfma', 'bf16x3', '--no-kernel-profile'],
     'OPT-IN, narrower arithmetic than the reference: split-bf16 (3-product) data / weight gradients of the wide PatchGAN layers; '
     'never the graded line'),
)


def secondary_measurements():
    """The other single-GPU measurements BASELINE.json / north_star name, taken by the SAME invocation after the headline's timed region so
    that they appear in the driver-timed record (SECONDARY above).  Each is a short child run of this script (own process: own
    caching-allocator pools and hipGraph; 8 timed steps after 3 warm-up steps, the same barrier / synchronize bracket, no CPU baseline)
    whose JSON line is condensed here.  The headline fields of the parent line are untouched."""
    import subprocess
    out = {}
    for key, extra, what in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__)] + extra + ['--steps', '8', '--warmup', '3', '--sustained-steps', '0',
                                                                     '--no-cpu-baseline', '--no-secondary']
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not line:
                out[key] = {'error': f'rc {r.returncode}: ' + r.stderr[-300:]}
                continue
            j = json.loads(line[-1])
            fams = (j.get('roofline') or {}).get('families') or {}
            sf = j.get('student_forward') or {}
            out[key] = {'what': what, 'metric': j['metric'], 'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'],
                        'steps': j['steps'], 'dtype': j['dtype'], 'per_gpu_batch': j['config']['per_gpu_batch'], 'launch': j['config']['launch'],
                        'schedule': j['config']['schedule'], 'collectives': j['config']['collectives'],
                        'launches_per_step': round(sum(v['launches_per_step'] for v in fams.values()), 1) if fams else None,
                        'student_forward_ms': sf.get('ms'), 'student_forward_frac_of_fp32_mfma_peak': sf.get('frac_of_fp32_mfma_peak'),
                        'dominant_kernel': (j.get('roofline') or {}).get('kernel'), 'dominant_frac': (j.get('roofline') or {}).get('frac'),
                        'wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:      # noqa: BLE001  (a secondary measurement must never cost the headline line)
            out[key] = {'error': f'{type(e).__name__}: {e}'}
    return out


def csrc_fingerprint(skip=()):
    """sha256 over the HIP sources + headers the library is built from (sorted by name): identifies the kernels a profile was taken from
    without git (the GPU box receives a snapshot without .git).  `skip`: translation units ADDED after the profiles were taken
    (profiles/<tag>_meta.json `added_after`): a new file changes no kernel the profiles describe; an edit to any file that existed does."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'cat_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h')) and name not in skip:
            h.update(name.encode())
            h.update(open(os.path.join(d, name), 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 'cat_hip.h'), 'rb').read())
    return h.hexdigest()[:16]


def student_forward_rate(model, batch, spade, graph=True):
    """BASELINE's second headline figure: the student generator's forward alone (train-mode norms, no grad), as TFLOP/s of true conv
    FLOPs and as a fraction of the fp32 MFMA peak.  FLOPs come from the library's own per-launch accounting (cat_prof_*)."""
    import ctypes as C
    from cat_amd import _lib, optim
    lib = _lib.load()
    model.set_input(batch)
    # every forward below stands for the forward of a NEW optimizer step: the per-step operand preparation of the fused blocks
    # (filter packing, parameter gathers -- keyed on the optimizer epoch) is part of what is timed / captured
    net_fwd = lambda net_, x_: (optim._bump_weights_epoch(), net_(x_))[1]
    if spade:
        net, x = model.modules_on_one_gpu.netG_student, model.input_semantics
    else:
        net, x = model.netG_student, model.real_A
    with torch.no_grad():
        for _ in range(3):
            net_fwd(net, x)
        torch.cuda.synchronize()
        lib.cat_prof_enable(1)
        net_fwd(net, x)
        torch.cuda.synchronize()
        n = lib.cat_prof_collect()
        lib.cat_prof_enable(0)
        name = C.create_string_buffer(64)
        cnt, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        gflop = 0.0
        for i in range(n):
            lib.cat_prof_family(i, name, 64, C.byref(cnt), C.byref(ms), C.byref(fl))
            gflop += fl.value / 1e9
        # ~330 launches of a few microseconds each: issued from Python the forward is host-bound, so it is timed as a captured
        # hipGraph (what GraphedStep replays inside the step); the eager figure is reported next to it
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net_fwd(net, x)
        e1.record()
        torch.cuda.synchronize()
        ms_eager = e0.elapsed_time(e1) / reps
        if graph:
            from cat_amd import parallel
            parallel.settle_collectives()      # (--dp-schedule 1: collectives were issued in this process; see its docstring)
            g = torch.cuda.CUDAGraph()
            optim._bump_weights_epoch()
            with torch.cuda.graph(g):
                net(x)
            g.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
    ms_fwd = e0.elapsed_time(e1) / reps
    tf = gflop / ms_fwd
    return {'ms': round(ms_fwd, 3), 'ms_eager_launches': round(ms_eager, 3), 'gflop': round(gflop, 2), 'tflops': round(tf, 2), 'frac_of_fp32_mfma_peak': round(tf / MFMA_F32_PEAK_TFLOPS, 4),
            'batch': int(x.shape[0])}


PROFILE_TAG = 'r06'
PMC_FILE = f'profiles/{PROFILE_TAG}_pmc_hbm.json'
STATS_FILE = f'profiles/{PROFILE_TAG}_kernel_stats_c2.txt'
META_FILE = f'profiles/{PROFILE_TAG}_meta.json'          # {"commit": ..., "date": ...}: the tree the committed profiles were taken from


def _profile_meta():
    path = os.path.join(ROOT, META_FILE)
    return json.load(open(path)) if os.path.exists(path) else {}


def _profile_commit():
    return _profile_meta().get('commit')


def _profiles_stale():
    """True if the committed profiles (kernel stats / PMC tables this line quotes) were taken from OTHER kernel sources than the ones running
    now (profiles/<tag>_meta.json records the csrc fingerprint of the tree tools/profile_r6.sh ran on); None if the meta file has none."""
    meta = _profile_meta()
    fp = meta.get('csrc_sha')
    return None if fp is None else fp != csrc_fingerprint(tuple(meta.get('added_after', ())))


def pmc_traffic(family):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (PMC_FILE: FETCH_SIZE and WRITE_SIZE in separate
    passes, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md §HBM; round 6: whole-step passes restricted to the library's conv /
    norm / depthwise kernels by --kernel-include-regex, tools/profile_r6.sh).  PMC collection cannot run
    inside the timed process, hence the lookup; None if absent."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None
    base = family.rsplit('_', 1)[0]               # conv_fwd32d_4x4x2x2 -> conv_fwd32d
    table = json.load(open(path))
    for k, v in table.items():
        if isinstance(v, dict) and k.startswith(base + '_kernel'):
            return v.get('hbm_bytes_per_launch')
    return None


def rocprof_avg_us(family):
    """Average launch duration (us) of the dominant kernel in the committed `rocprofv3 --kernel-trace --stats` summary of this command."""
    path = os.path.join(ROOT, STATS_FILE)
    if not os.path.exists(path):
        return None
    base = family.rsplit('_', 1)[0] + '_kernel'
    for line in open(path):
        if line.startswith('#') or base not in line:
            continue
        cols = line.split()
        try:
            return float(cols[-3])       # ... calls total_us avg_us pct us/step
        except (ValueError, IndexError):
            continue
    return None


def kernel_roofline(model, step, args):
    """HIP-event timing of every implicit-GEMM conv launch (events recorded on the launch stream by libcat_hip's
    profiling hook) over a few extra steps; achieved = algorithmic conv FLOPs / summed kernel time."""
    from cat_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, 'cat_prof_enable'):
        return None
    import ctypes as C
    from cat_amd import ops
    nsteps = 2
    # kernels are timed one at a time: the stream-level concurrency used in the timed region (block branches and the teacher on
    # side streams) is switched off here, otherwise co-running kernels share the chip and every per-kernel duration is inflated
    was_branch, was_teacher = ops.branch_streams_enabled(), getattr(model, 'teacher_side_stream', False)
    ops.set_branch_streams(False)
    model.teacher_side_stream = False
    step(9_999)
    torch.cuda.synchronize()
    lib.cat_prof_enable(1)
    for i in range(nsteps):
        step(10_000 + i)
    torch.cuda.synchronize()
    n = lib.cat_prof_collect()
    lib.cat_prof_enable(0)
    ops.set_branch_streams(was_branch)
    model.teacher_side_stream = was_teacher
    fams = {}
    name = C.create_string_buffer(64)
    cnt, ms, fl = C.c_int64(), C.c_double(), C.c_double()
    for i in range(n):
        lib.cat_prof_family(i, name, 64, C.byref(cnt), C.byref(ms), C.byref(fl))
        fams[name.value.decode()] = {'launches_per_step': cnt.value / nsteps, 'ms_per_step': ms.value / nsteps,
                                     'gflop_per_step': fl.value / 1e9 / nsteps}
    conv = {k: v for k, v in fams.items() if k.startswith('conv')}
    if not conv:
        return None
    dom = max(conv, key=lambda k: conv[k]['ms_per_step'])
    tot_ms = sum(v['ms_per_step'] for v in conv.values())
    tot_gf = sum(v['gflop_per_step'] for v in conv.values())
    d = conv[dom]
    achieved = d['gflop_per_step'] / d['ms_per_step'] if d['ms_per_step'] > 0 else 0.0   # GFLOP/ms == TFLOP/s
    headline = getattr(args, 'workload', 'c2') == 'c2' and args.size == 256 and args.batch == 16
    avg_us = 1e3 * d['ms_per_step'] / max(d['launches_per_step'], 1)
    rp_us = rocprof_avg_us(dom) if headline else None
    gflop_launch = d['gflop_per_step'] / max(d['launches_per_step'], 1)
    return {'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 3), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': pmc_traffic(dom) if headline else None,
            'traffic_source': PMC_FILE + ' (tools/profile_r6.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only, over 2 serial '
                              'steps of this command, counters collected for the conv / norm / depthwise kernels through --kernel-include-regex; per-launch '
                              'average over ALL launches of the family in the step, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; not a '
                              'same-run measurement)',
            'avg_launch_us': round(avg_us, 3),
            # the same fraction from the committed rocprofv3 --kernel-trace --stats summary (its average launch duration of this kernel):
            # HIP events see the kernel alone, rocprof's span includes dispatch overhead -- the two bracket the truth (~3 % apart)
            'rocprof': None if rp_us is None else {'avg_launch_us': rp_us, 'achieved': round(1e3 * gflop_launch / rp_us, 3),
                                                   'frac': round(1e3 * gflop_launch / rp_us / MFMA_F32_PEAK_TFLOPS, 4), 'source': STATS_FILE},
            'profiles_commit': _profile_commit(), 'profiles_stale': _profiles_stale(), 'csrc_sha': csrc_fingerprint(),
            'profiles_added_after': list(_profile_meta().get('added_after', ())),
            'all_conv': {'achieved': round(tot_gf / tot_ms, 3) if tot_ms else 0.0, 'ms_per_step': round(tot_ms, 3),
                         'gflop_per_step': round(tot_gf, 2)},
            'families': {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in fams.items()}}


if __name__ == '__main__':
    main()
