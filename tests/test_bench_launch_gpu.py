"""GPU: `bench.py --gpus N` end to end through its OWN launcher path (the re-exec under torch.distributed.run, one process per rank, the
barrier / max-over-ranks timing bracket, rank 0's single JSON line) -- what the driver's 8-GPU node runs first.  The test box has one
MI355X and RCCL refuses two ranks per device, so the two ranks share GPU 0 (CAT_FORCE_DEVICE=0) with gloo as the transport
(CAT_DIST_BACKEND=gloo): the launch line, environment handling, data-parallel schedule (hipGraph segments around the two bucket all-reduces for
the inception distillers, eager SynchronizedBatchNorm exchanges for GauGAN) and the JSON contract are the production ones, only the wire differs.
Reference semantics being scheduled: models/networks.py:157-161 (DataParallel), distillers/inception_distiller.py:136-148 (per-shard KA)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env_extra=None, timeout=1500):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):      # the test process itself may sit under a launcher
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert len(lines) == 1, ('stdout must carry exactly ONE JSON line', r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0]), r.stderr


TWO_RANKS_ONE_GPU = {'CAT_FORCE_DEVICE': '0', 'CAT_DIST_BACKEND': 'gloo'}


def _check_line(j, batch, steps, warmup):
    assert j['n_gpus'] == 2 and j['steps'] == steps and j['warmup'] == warmup
    assert j['unit'] == 'images/sec' and j['higher_is_better'] is True and j['scaling'] == 'weak' and j['data'] == 'synthetic'
    assert j['dtype'] == 'f32' and j['vs_baseline'] is None
    cfg = j['config']
    assert cfg['ranks'] == 2 and cfg['parallelism'] == 'dp2' and cfg['schedule'] == 'data-parallel'
    assert cfg['per_gpu_batch'] == batch and cfg['global_batch'] == 2 * batch
    assert 'all-reduce of the flat gradient buckets' in cfg['collectives'] and cfg['collectives'].startswith('gloo')
    # whole-job aggregate: the units ALL ranks processed / the slowest rank's time
    assert abs(j['value'] - 2 * batch * steps / (j['ms_per_step'] * steps / 1e3)) <= 2e-3 * j['value']
    assert j['value'] > 0 and 'cpu_baseline' not in j


@pytest.mark.timeout(1800)
def test_bench_gpus2_c2_through_its_own_launcher():
    j, err = _bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--batch', '4'], TWO_RANKS_ONE_GPU)
    _check_line(j, 4, 2, 1)
    assert 'hipGraph segments around the collectives' in j['config']['launch'], j['config']['launch']      # not the eager fallback
    assert 'capture failed' not in err
    assert j['roofline'] is not None and j['student_forward'] is not None      # the measurement passes ran on every rank and met at the barrier


@pytest.mark.timeout(1800)
def test_bench_gpus2_c3_through_its_own_launcher():
    j, err = _bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--workload', 'c3', '--batch', '2', '--no-kernel-profile'],
                    TWO_RANKS_ONE_GPU)
    _check_line(j, 2, 2, 1)
    assert 'hipGraph segments around the collectives' in j['config']['launch'], j['config']['launch']
    assert 'capture failed' not in err


@pytest.mark.timeout(1800)
def test_bench_gpus2_spade_through_its_own_launcher():
    j, _ = _bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--workload', 'spade', '--batch', '1', '--no-kernel-profile'],
                  TWO_RANKS_ONE_GPU)
    _check_line(j, 1, 2, 1)
    assert j['config']['launch'] == 'eager'      # SynchronizedBatchNorm exchanges inside the passes: not captured


@pytest.mark.timeout(900)
def test_bench_dp_schedule_on_one_gpu_over_rccl():
    """The N = 1 point of the scaling curve in the data-parallel launch mode: a world_size-1 RCCL group (library load, device_id= init,
    all-reduce of the real buckets), hipGraph segments, deferred Adam G.  This is the `secondary.dp1` leg of the default headline run."""
    j, err = _bench(['--dp-schedule', '1', '--steps', '3', '--warmup', '2', '--no-cpu-baseline', '--no-kernel-profile', '--no-secondary'])
    assert j['n_gpus'] == 1 and j['config']['schedule'] == 'data-parallel' and j['config']['collectives'].startswith('rccl')
    assert 'hipGraph segments around the collectives' in j['config']['launch'], (j['config']['launch'], err[-1500:])
