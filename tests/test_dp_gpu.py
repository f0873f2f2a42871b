"""Data-parallel steps on REAL kernels: two ranks (two processes sharing the one MI355X of the test box, gloo as the transport
-- RCCL refuses two ranks on one device) run the distillers' DP path on their shard; the result is compared with the oracle's
DataParallel restatement (n_shards=2) and the replicas must stay bit-identical.  This exercises exactly the code the 2/4/8-GPU
bench runs, minus RCCL itself: bucket all-reduce + deferred student update (inception), SynchronizedBatchNorm statistics
exchange with the clamp formula + gradient averaging (SPADE)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import helpers as H
from oracle import detfill

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from cat_amd import parallel
    parallel.init_distributed(backend='gloo')
    return parallel.DataParallelReducer()


def _probe(net, keys):
    sd = net.state_dict()      # numpy: plain pickling through the queue (torch tensors travel as shared-memory handles that die with the worker)
    return {k: sd[k].detach().float().cpu().reshape(-1)[:64].numpy().copy() for k in keys}


def _worker_inception(rank, world, port, q):
    red = _init(rank, world, port)
    from cat_amd import parallel
    g = H.load('step_in.npz')
    meta = json.loads(str(g['meta']))
    opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                     lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16)
    model = H.build_distiller(opt, g['student_shapes'])
    model.enable_data_parallel(red, overlap=True)
    n, s = 4, meta['size']
    for step in range(2):
        A, B = detfill.images((n, 3, s, s), 810 + step), detfill.images((n, 3, s, s), 820 + step)
        model.set_input(parallel.shard_batch({'A': A, 'B': B, 'A_paths': [], 'B_paths': []}, rank, world))
        model.optimize_parameters(step)
    # a Trainer saves / evaluates right after optimize_parameters: the deferred student all-reduce + Adam step must be completed by
    # save_networks itself (no explicit finish_pending here), and the checkpoint must hold the completed update
    assert model._pending_G is not None
    import tempfile
    model.save_dir = tempfile.mkdtemp(prefix=f'cat_dp_{rank}_')
    model.save_networks('dp')
    assert model._pending_G is None
    torch.cuda.synchronize()
    keys = ['down_sampling.1.weight', 'features.4.res_ops.1.1.0.weight', 'up_sampling.7.weight']
    saved = torch.load(os.path.join(model.save_dir, 'dp_net_G.pth'), map_location='cpu')
    live = model.netG_student.state_dict()
    for k in keys:
        assert torch.equal(saved[k], live[k].detach().cpu()), k
    q.put((rank, {k: float(v) for k, v in model.get_current_losses().items()}, _probe(model.netG_student, keys),
           _probe(model.netD, ['model.0.weight', 'model.8.weight'])))
    torch.distributed.destroy_process_group()


def _worker_spade(rank, world, port, q):
    red = _init(rank, world, port)
    from cat_amd import parallel
    import test_spade_gpu as TS
    g, opt, lab, ins, img, sds, cfg = TS.fixture()
    opt.isTrain, opt.distiller, opt.log_dir = True, 'spade', '/tmp/cat_amd_logs'
    model = TS.build_spade_distiller(opt, sds)
    model.enable_data_parallel(red)
    batch = spade_batch(opt, int(g['h']), int(g['w']))
    model.set_input(parallel.shard_batch(batch, rank, world))
    model.optimize_parameters(0)
    torch.cuda.synchronize()
    m = model.modules_on_one_gpu
    q.put((rank, {k: float(v) for k, v in model.get_current_losses().items()},
           _probe(m.netG_student, ['fc.weight', 'up_1.shortcut.1.conv.weight', 'head_0.spade.param_free_norm.running_var', 'conv_img.weight']),
           _probe(m.netD, ['discriminator_0.model0.0.weight', 'discriminator_1.model3.0.0.weight_orig'])))
    torch.distributed.destroy_process_group()


def spade_batch(opt, h, w, n=4):
    rng = np.random.default_rng(77)
    lab = np.repeat(np.repeat(rng.integers(0, opt.input_nc, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int64)
    ins = np.repeat(np.repeat(rng.integers(0, 99, (n, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32)
    return {'label': torch.from_numpy(lab), 'instance': torch.from_numpy(ins), 'image': detfill.images((n, 3, h, w), 78), 'path': []}


def _run(worker):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        r = q.get(timeout=600)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return out


@pytest.mark.timeout(900)
def test_inception_data_parallel_two_ranks():
    from oracle import ref_cpu
    out = _run(_worker_inception)
    (l0, s0, d0), (l1, s1, d1) = out[0], out[1]
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k         # replicas stay identical
    for k in d0:
        assert np.array_equal(d0[k], d1[k]), k
    # oracle: nn.DataParallel semantics over the 2 shards
    g = H.load('step_in.npz')
    meta = json.loads(str(g['meta']))
    opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                     lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16)
    ncfg = H.cfg_for(meta['norm'])
    cfg = dict(T=ncfg, S=ncfg, D=ncfg, dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'], lambda_recon=meta['lambda_recon'],
               lambda_distill=meta['lambda_distill'], lambda_gan=1.0, lr=meta['lr'], beta1=opt.beta1)
    S = detfill.fill_state_dict(H.sd_from_shapes(g['student_shapes']), H.SEED_S)
    st = ref_cpu.DistillState(H.teacher_sd(opt), S, H.disc_sd(opt, 6 if meta['dataset_mode'] == 'aligned' else 3), cfg)
    n, s = 4, meta['size']
    for step in range(2):
        A, B = detfill.images((n, 3, s, s), 810 + step), detfill.images((n, 3, s, s), 820 + step)
        ref_losses = ref_cpu.distill_step(st, A, B, n_shards=2)
    for name in ('G_recon', 'G_gan', 'D_fake', 'D_real'):        # means over the gathered batch = mean of the shard means
        got = 0.5 * (l0['G_loss/' + name if name.startswith('G') else 'D_loss/' + name] + l1['G_loss/' + name if name.startswith('G') else 'D_loss/' + name])
        assert abs(got - ref_losses[name]) <= 5e-3 * max(1.0, abs(ref_losses[name])), (name, got, ref_losses[name])
    got = l0['G_loss/G_distill'] + l1['G_loss/G_distill']          # KA terms are summed over shards
    assert abs(got - ref_losses['G_distill']) <= 5e-3 * abs(ref_losses['G_distill'])
    for k, v in s0.items():
        ref = st.S[k].detach().reshape(-1)[:64].numpy()
        assert float(np.abs(v - ref).max()) <= 4 * meta['lr'] + 1e-3 * float(np.abs(ref).max()), k


@pytest.mark.timeout(900)
def test_spade_data_parallel_two_ranks():
    import test_spade_gpu as TS
    from oracle import ref_spade_cpu as R
    out = _run(_worker_spade)
    (l0, s0, d0), (l1, s1, d1) = out[0], out[1]
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    for k in d0:
        assert np.array_equal(d0[k], d1[k]), k
    g, opt, lab, ins, img, sds, cfg = TS.fixture()
    batch = spade_batch(opt, int(g['h']), int(g['w']))
    sem = R.preprocess_input(batch['label'], batch['instance'], opt.input_nc)
    st = R.SpadeState(sds['T'], sds['S'], sds['D'], sds['V'], cfg)
    ref = R.spade_step(st, sem, batch['image'], n_shards=2)
    for k in ('G_gan', 'G_feat', 'G_vgg', 'G_distill', 'D_real', 'D_fake'):
        key = ('D_loss/' if k.startswith('D') else 'G_loss/') + k
        got = 0.5 * (l0[key] + l1[key])            # every SPADE loss is a per-replica value averaged over replicas
        assert abs(got - ref[k]) <= 5e-3 * max(abs(ref[k]), 1e-2), (k, got, ref[k])
    lr = cfg['lr']
    for k, v in s0.items():
        r = st.S[k].detach().reshape(-1)[:64].numpy()
        tol = 1e-3 * float(np.abs(r).max()) + (0 if 'running' in k else 2 * lr)
        assert float(np.abs(v - r).max()) <= tol, (k, float(np.abs(v - r).max()), tol)
