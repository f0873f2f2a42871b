"""Two gloo ranks on GPU 0 running the data-parallel inception step; prints per-step checksums (teacher output / taps, student output, losses,
weight probes) so that a good and a bad run of a racy schedule can be diffed.   CAT_BRANCH_STREAMS=0 python tools/debug/dp_race.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, q):
    import test_dp_gpu as T
    import helpers as H
    from oracle import detfill
    red = T._init(rank, world, port, 'gloo')
    from cat_amd import parallel
    g = H.load('step_in.npz')
    meta = json.loads(str(g['meta']))
    opt = H.make_opt(norm=meta['norm'], track=meta['track'], ndf=meta['ndf'], dataset_mode=meta['dataset_mode'], gan_mode=meta['gan_mode'],
                     lambda_recon=meta['lambda_recon'], lambda_distill=meta['lambda_distill'], student_ngf=16)
    model = H.build_distiller(opt, g['student_shapes'])
    model.enable_data_parallel(red, overlap=True)
    n, s = 4, meta['size']
    out = []
    for step in range(2):
        A, B = detfill.images((n, 3, s, s), 810 + step), detfill.images((n, 3, s, s), 820 + step)
        model.set_input(parallel.shard_batch({'A': A, 'B': B, 'A_paths': [], 'B_paths': []}, rank, world))
        model.optimize_parameters(step)
        torch.cuda.synchronize()
        rec = {'T': float(model.Tfake_B.double().sum()), 'S': float(model.Sfake_B.double().sum())}
        for k, v in sorted(model.Tacts.items()):
            rec['Ta_' + k[:12]] = float(v.double().sum())
        for k, v in sorted(model.Sacts.items()):
            rec['Sa_' + k[:12]] = float(v.detach().double().sum())
        for k, v in model.get_current_losses().items():
            rec[k] = float(v)
        rec['gradG'] = float(sum(b.double().abs().sum() for b in model.optimizer_G.flat_grads()))
        rec['gradD'] = float(sum(b.double().abs().sum() for b in model.optimizer_D.flat_grads()))
        out.append(rec)
    model.finish_pending()
    torch.cuda.synchronize()
    out.append({'wG': float(sum(p.detach().double().abs().sum() for p in model.netG_student.parameters())),
                'wD': float(sum(p.detach().double().abs().sum() for p in model.netD.parameters()))})
    q.put((rank, out))
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    import test_dp_gpu as T
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(600)
    for r in (0, 1):
        for i, rec in enumerate(res[r]):
            print('rank %d step %d ' % (r, i) + ' '.join('%s=%.6f' % kv for kv in rec.items()))
