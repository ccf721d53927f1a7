"""
The N > 1 path of bench.py END TO END on the one-GPU box: `bench.launch_ranks` starts two rank
processes through torch.distributed.run; each supervises one fresh worker per shard mode; the workers (two gloo ranks
sharing device 0 -- BENCH_DEVICE / BENCH_BACKEND, the rehearsal switches) run the sharded model through the HIP library,
all-gather the visibilities, exchange the gradients, and rank 0's self-check compares the gathered visibilities and the
exchanged gradients with the UNSHARDED float64 model: both partitions must agree with it to 1e-5 / 1e-4.  The multi-GPU
runs of the driver execute this same code with one rank per GPU over RCCL.  Replaced pattern: the reference's
single-process device loop, optim.py:1539-1566.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _run(args, world=2, **env):
    import bench
    e = dict(os.environ, BENCH_DEVICE='0', BENCH_BACKEND='gloo', BENCH_MODE_TIMEOUT='500', **env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        e.pop(k, None)
    rc, line = bench.launch_ranks(world, [os.path.join(ROOT, 'bench.py'), '--gpus', str(world)] + args, env=e, timeout=1100)
    return rc, (None if line is None else json.loads(line))


@pytest.mark.parametrize('args, tiles', [
    (['--workload', 'c2', '--nt', '4', '--steps', '2', '--warmup', '1'], False),           # HERA-19: ONE block, contiguous baseline shards
    (['--workload', 'c4', '--nf', '8', '--nt', '2', '--steps', '1', '--warmup', '1'], True),  # HERA-128 + point sources: tile plan
])
def test_two_ranks_both_partitions_match_the_unsharded_float64_model(args, tiles):
    rc, res = _run(args)
    assert rc == 0 and res is not None
    assert res['n_gpus'] == 2 and res['dist']['world_size'] == 2 and res['dist']['backend'] == 'gloo'
    modes = {res['dist']['shard']: res['dist']['selfcheck']}
    assert len(res['alt']) == 1 and 'failed' not in res['alt'][0], res['alt']
    modes[res['alt'][0]['shard']] = res['alt'][0]['selfcheck']
    assert set(modes) == {'freq', 'bl'}
    for mode, ck in modes.items():
        assert ck['ok'] and ck['vis_relmax'] < 1e-5 and ck['grad_relmax'] < 1e-4, (mode, ck)
    if tiles:
        plan = res['dist']['tile_plan_load'] if res['dist']['shard'] == 'bl' else res['alt'][0]['tile_plan_load']
        assert plan is not None and len(plan) == 2


@pytest.mark.parametrize('args', [
    ['--workload', 'c2', '--nt', '4', '--steps', '2', '--warmup', '1'],
    ['--workload', 'c4', '--nf', '8', '--nt', '2', '--steps', '1', '--warmup', '1'],
])
def test_two_ranks_pixel_partition_matches_the_unsharded_float64_model(args):
    """`--shard pix` (round 5; SURVEY 8e's third axis, not in the default pair of modes): every rank contracts every second sky
    pixel / point source for all baselines, times and channels, the partial visibilities are summed by a differentiable
    all-reduce per time chunk and the gradients by all-reduce; same self-check, same tolerances"""
    rc, res = _run(args + ['--shard', 'pix'])
    assert rc == 0 and res is not None
    assert res['n_gpus'] == 2 and res['dist']['shard'] == 'pix' and 'pixel-sharded x2' in res['config']['parallelism']
    ck = res['dist']['selfcheck']
    assert ck['ok'] and ck['vis_relmax'] < 1e-5 and ck['grad_relmax'] < 1e-4, ck


def test_a_wrong_gather_order_fails_the_selfcheck_and_the_other_mode_survives():
    """the self-check is not decorative: with the tile shards' inverse permutation left out (BENCH_BREAK_INVERSE=1, a
    test-only switch) the gathered baselines are in rank order, the `bl` worker fails its self-check (exit code 5) and the
    line is the `freq` mode's, with the failure recorded under `alt`"""
    rc, res = _run(['--workload', 'c4', '--nf', '8', '--nt', '2', '--steps', '1', '--warmup', '1'], BENCH_BREAK_INVERSE='1')
    assert rc == 0 and res['dist']['shard'] == 'freq' and res['dist']['selfcheck']['ok']
    alt = res['alt'][0]
    assert alt['shard'] == 'bl' and alt['failed'] == 5 and not alt['selfcheck']['ok']
    assert alt['selfcheck']['vis_relmax'] > 1e-3


def test_one_rank_over_rccl_runs_the_supervisor_and_the_selfcheck():
    """BENCH_FORCE_DIST=1: the N > 1 code path on ONE rank over RCCL (nccl backend, every collective executed with world
    size 1), through the supervisor, both partitions, with the self-check in the line"""
    import subprocess
    e = dict(os.environ, BENCH_FORCE_DIST='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'BENCH_BACKEND', 'BENCH_DEVICE'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'c2', '--nt', '4', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline'], env=e, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res['n_gpus'] == 1 and res['dist']['backend'] == 'nccl' and res['dist']['selfcheck']['ok']
    assert res['alt'][0]['selfcheck']['ok'] and {res['dist']['shard'], res['alt'][0]['shard']} == {'freq', 'bl'}
    assert res['dist']['collective_timeout_s'] in (60.0, 600.0)


def test_one_rank_under_the_launcher_over_rccl_uses_the_launcher_store():
    """the driver's command shape -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N` -- with N = 1 and BENCH_FORCE_DIST=1: the rank process supervises one worker per
    partition, the workers build their RCCL groups through a PrefixStore on the LAUNCHER's TCPStore (the combination the
    multi-GPU runs use; the gloo rehearsals cover the store, test_one_rank_over_rccl the backend)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, BENCH_FORCE_DIST='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'BENCH_BACKEND', 'BENCH_DEVICE'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--workload', 'c2',
                        '--nt', '4', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], env=e, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res['n_gpus'] == 1 and res['dist']['backend'] == 'nccl' and res['dist']['selfcheck']['ok']
    assert res['alt'][0]['selfcheck']['ok'] and {res['dist']['shard'], res['alt'][0]['shard']} == {'freq', 'bl'}
