"""CPU, world_size 2 over gloo: the data-parallel path (cat_amd/parallel.py) and the DataParallel loss semantics it has to
reproduce: recon / GAN terms are means over the gathered batch, the KA term is the SUM of per-shard KAs.
Rank-local gradients (computed here with the oracle, KA seed scaled by world_size exactly as
InceptionDistiller.backward_G does) are exchanged with the product's reducer and must equal the oracle's
nn.DataParallel restatement on the full batch."""
import json
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch.nn.functional as F
    from cat_amd import parallel
    from oracle import detfill, ref_cpu
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = parallel.init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    red = parallel.DataParallelReducer()

    # 1. plain bucket reduction: sum + 1/world applied by the consumer
    class FakeOpt:
        grad_scale = 1.0

        def __init__(self):
            self.b = [torch.full((5,), float(rank + 1)), torch.arange(4, dtype=torch.float32) * (rank + 1)]

        def flat_grads(self):
            return self.b
    fo = FakeOpt()
    red.reduce(fo)
    assert torch.equal(fo.b[0], torch.full((5,), 3.0)) and torch.equal(fo.b[1], torch.arange(4, dtype=torch.float32) * 3)
    assert fo.grad_scale == 0.5

    # 1b. several statistics buffers as ONE collective (the lockstep units of fused_spade.prepass): every buffer ends up exactly as after
    #     its own all_reduce_sum_, shapes and views preserved
    bufs = [torch.arange(6, dtype=torch.float32).reshape(2, 3) * (rank + 1), torch.full((5,), 0.25 * (rank + 1)), torch.ones(1) * rank]
    singles = [b.clone() for b in bufs]
    wide = torch.zeros(10)
    view = wide[2:7]
    view.copy_(bufs[1])
    red.all_reduce_sum_many_([bufs[0], view, bufs[2]])
    for t in singles:
        red.all_reduce_sum_(t)
    assert torch.equal(bufs[0], singles[0]) and torch.equal(view, singles[1]) and torch.equal(bufs[2], singles[2])
    assert float(wide[:2].abs().sum()) == 0.0 and float(wide[7:].abs().sum()) == 0.0
    # 1c. adjacent slices of one arena (what fused_spade._drive_many hands out): reduced IN PLACE on the arena span -- no pack / unpack copies
    #     (the arena's storage is what travels: its address must be the message's)
    arena = torch.arange(24, dtype=torch.float32) * (rank + 1)
    parts = [arena[0:8], arena[8:16], arena[16:24]]
    seen = []
    real_all_reduce = parallel.dist.all_reduce

    def spy(t, *a, **k):
        seen.append((t.data_ptr(), t.numel()))
        return real_all_reduce(t, *a, **k)
    parallel.dist.all_reduce = spy
    try:
        red.all_reduce_sum_many_(parts)
    finally:
        parallel.dist.all_reduce = real_all_reduce
    assert seen == [(arena.data_ptr(), 24)], seen
    assert torch.equal(arena, torch.arange(24, dtype=torch.float32) * 3)

    # 2. DataParallel loss semantics on a 2-shard batch (InstanceNorm config: only KA is shard dependent)
    g = H.load('step_in.npz')
    opt = H.make_opt(norm='instance', track=False, ndf=64)
    ncfg = H.cfg_for('instance')
    S = {k: v.clone().requires_grad_(not k.endswith('num_batches_tracked')) for k, v in
         detfill.fill_state_dict(H.sd_from_shapes(g['student_shapes']), H.SEED_S).items()}
    T, D = H.teacher_sd(opt), H.disc_sd(opt, 3)
    A = detfill.images((4, 3, 32, 32), 900)
    batch = parallel.shard_batch({'A': A, 'paths': ['x']}, rank, world)
    assert batch['A'].shape[0] == 2 and torch.equal(batch['A'], A[2 * rank:2 * rank + 2]) and batch['paths'] == ['x']
    lam_recon, lam_distill = 5.0, 1.0
    a = batch['A']
    with torch.no_grad():
        t_out, t_acts = ref_cpu.inception_generator(T, a, ncfg, training=False)
    s_out, s_acts = ref_cpu.inception_generator(S, a, ncfg, training=True)
    recon = F.l1_loss(s_out, t_out)
    gan = ref_cpu.gan_loss('lsgan', ref_cpu.nlayer_discriminator(D, s_out, ncfg, training=True), True, False)
    kas = [ref_cpu.ka(s_acts[n], t_acts[n]) for n in ref_cpu.MAPPING_LAYERS]
    # seeds exactly as cat_amd/distillers/inception_distiller.py backward_G: lambda_gan, lambda_recon, -lambda_distill*world
    torch.autograd.backward([gan, recon] + kas, [torch.tensor(1.0), torch.tensor(lam_recon)] + [torch.tensor(-lam_distill * world)] * 4)
    names = [k for k, v in S.items() if v.requires_grad]
    flat = torch.cat([S[k].grad.reshape(-1) for k in names])
    pend = red.reduce_async([flat])
    pend.wait()
    flat = flat / world
    if rank == 0:
        # nn.DataParallel restatement on the full batch (2 shards)
        S2 = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in S.items()}
        with torch.no_grad():
            tt = [ref_cpu.inception_generator(T, c, ncfg, training=False) for c in A.chunk(2, 0)]
        ss = [ref_cpu.inception_generator(S2, c, ncfg, training=True) for c in A.chunk(2, 0)]
        sf, tf = torch.cat([o[0] for o in ss], 0), torch.cat([o[0] for o in tt], 0)
        loss = lam_recon * F.l1_loss(sf, tf)
        loss = loss + ref_cpu.gan_loss('lsgan', torch.cat([ref_cpu.nlayer_discriminator(D, o[0], ncfg, training=True) for o in ss], 0), True, False)
        for n in ref_cpu.MAPPING_LAYERS:
            loss = loss + lam_distill * sum(-ref_cpu.ka(s[1][n], t[1][n]) for s, t in zip(ss, tt))
        loss.backward()
        ref = torch.cat([S2[k].grad.reshape(-1) for k in names])
        err = float((flat - ref).abs().max() / ref.abs().max())
        q.put(err)
    t = red.max_over_ranks(float(rank + 1))
    assert t == float(world)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_data_parallel_semantics_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    err = q.get(timeout=10)
    assert err < 1e-4, err


def _worker_syncbn(rank, world, port, q):
    """The SynchronizedBatchNorm exchange protocol of cat_amd.ops.SyncBNFn (sums layout, count, clamp, local parameter gradients
    + bucket averaging), restated with torch ops on CPU over gloo, against the oracle's whole-batch sync_bn."""
    from cat_amd import parallel
    from oracle import detfill
    from oracle import ref_spade_cpu as R
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    parallel.init_distributed(backend='gloo')
    red = parallel.DataParallelReducer()
    c, eps = 5, 1e-5
    X = detfill.normal((4, c, 6, 7), 300) * 2 + 1
    X[:, 0] = 0.75                                   # zero-variance channel -> clamp(eps) path
    DY = detfill.normal((4, c, 6, 7), 301)
    gamma, beta = 1 + 0.2 * detfill.normal((c,), 302), 0.1 * detfill.normal((c,), 303)
    x, dy = X.chunk(world, 0)[rank], DY.chunk(world, 0)[rank]
    m = x.shape[0] * x.shape[2] * x.shape[3]
    sums = torch.cat([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))])          # [sum x | sum x^2]  (cat_bn_stats_fwd)
    red.all_reduce_sum_(sums)
    count = m * world
    mean = sums[:c] / count
    var = ((sums[c:] - sums[:c] * mean) / count).clamp_min(0)
    a = var.clamp(eps) ** -0.5                                             # cat_bn_finalize, clamp = 1
    b = -mean * a
    xh = x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    pre = xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    g = dy * (pre > 0).float()                                             # fused ReLU
    local = torch.cat([g.sum((0, 2, 3)), (g * xh).sum((0, 2, 3))])         # cat_bn_stats_bwd
    tot = red.all_reduce_sum_(local.clone())
    dx = (gamma * a).view(1, -1, 1, 1) * (g - (tot[:c] / count).view(1, -1, 1, 1) - xh * (tot[c:] / count).view(1, -1, 1, 1))
    dgamma, dbeta = local[c:].clone(), local[:c].clone()                   # LOCAL sums; the gradient bucket is then averaged
    bucket = torch.cat([dgamma, dbeta])
    red.reduce([bucket])
    bucket = bucket / world
    # non-contiguous parameter broadcast (padded channels_last conv weight)
    w = torch.zeros(3, 2, 2, 4)[..., :3].permute(0, 3, 1, 2)
    w.copy_(detfill.normal((3, 3, 2, 2), 310 + rank))

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(w)
    mod = M()
    red.broadcast_parameters([mod])
    assert torch.equal(mod.w.detach(), detfill.normal((3, 3, 2, 2), 310)) and not mod.w.is_contiguous()
    # oracle: whole batch, loss = mean over replicas of the per-replica loss
    sd = {'n.weight': gamma.clone().requires_grad_(True), 'n.bias': beta.clone().requires_grad_(True),
          'n.running_mean': torch.zeros(c), 'n.running_var': torch.ones(c)}
    xr = X.clone().requires_grad_(True)
    yr = R.sync_bn(sd, 'n', xr, True, True, act='relu')
    ((yr * DY).sum() / world).backward()
    ref_dx = xr.grad.chunk(world, 0)[rank] * world          # each rank back-propagates its UNSCALED replica loss
    e1 = float((dx - ref_dx).abs().max() / ref_dx.abs().max())
    ref_b = torch.cat([sd['n.weight'].grad, sd['n.bias'].grad])
    e2 = float((bucket - ref_b).abs().max() / ref_b.abs().max())
    q.put((rank, e1, e2))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_sync_batchnorm_protocol_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_syncbn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for _ in range(2):
        rank, e1, e2 = q.get(timeout=10)
        assert e1 < 1e-4 and e2 < 1e-4, (rank, e1, e2)
