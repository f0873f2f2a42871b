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
