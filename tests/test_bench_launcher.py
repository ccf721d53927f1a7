"""
`python bench.py --gpus N` invoked plainly must start N rank processes itself: the launcher
in bench.py is driven here with STUB rank scripts on the CPU -- result relay, exit codes, world-size check,
timeout.  The real ranks need GPUs; the launcher itself never touches one.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body):
    path = tmp_path / 'stub_rank.py'
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_launcher_relays_rank0_result_line(tmp_path):
    stub = _stub(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        assert os.environ['MASTER_ADDR'] == '127.0.0.1'
        print('chatter from rank %d' % rank)
        if rank == 0:
            print(json.dumps(dict(metric='m', value=1.5, n_gpus=world, argv=sys.argv[1:])))
        """)
    out = (tmp_path / 'out.txt').open('w+')
    rc, line = bench.launch_ranks(2, [stub, '--gpus', '2', '--steps', '3'], timeout=120, out=out)
    assert rc == 0
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['argv'] == ['--gpus', '2', '--steps', '3']
    out.seek(0)
    assert out.read() == line + '\n'                    # exactly ONE line on the result stream


def test_launcher_propagates_rank_failure(tmp_path):
    stub = _stub(tmp_path, """
        import json, os, sys
        if int(os.environ['RANK']) == 1:
            sys.exit(7)
        print(json.dumps(dict(metric='m', value=1.0, n_gpus=2)))
        """)
    rc, line = bench.launch_ranks(2, [stub], timeout=120)
    assert rc != 0


def test_launcher_rejects_result_for_another_world_size(tmp_path):
    stub = _stub(tmp_path, """
        import json, os
        if int(os.environ['RANK']) == 0:
            print(json.dumps(dict(metric='m', value=1.0, n_gpus=1)))      # a rank that silently ran alone
        """)
    rc, line = bench.launch_ranks(2, [stub], timeout=120)
    assert rc != 0 and line is None


def test_launcher_needs_a_result_line(tmp_path):
    rc, line = bench.launch_ranks(2, [_stub(tmp_path, "print('nothing useful')")], timeout=120)
    assert rc != 0 and line is None


def test_launcher_kills_ranks_that_hang(tmp_path):
    stub = _stub(tmp_path, """
        import time
        time.sleep(600)
        """)
    rc, line = bench.launch_ranks(2, [stub], timeout=20)
    assert rc == 124 and line is None


def test_bench_main_becomes_the_launcher_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` in a GPU-less container: the parent starts two ranks (which then refuse to run
    without a GPU) and exits non-zero; it must not run a single rank and print n_gpus 1"""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    env['BENCH_LAUNCH_TIMEOUT'] = '300'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip('covered by the GPU rehearsal (BENCH_DEVICE=0 BENCH_BACKEND=gloo)')
    assert r.returncode != 0
    assert r.stdout.strip() == b''                         # no result line, in particular no n_gpus: 1
    err = r.stderr.decode()
    assert 'rank 0 needs GPU' in err or 'rank 1 needs GPU' in err or 'needs a GPU' in err, err[-2000:]


def test_other_workloads_are_child_runs_that_cannot_lose_the_line(tmp_path):
    """after the timed region the N = 1 default run measures configs 3, 2, the config-5 rank share and the
    headline workload without mirror pairs in CHILD processes of the same script and attaches what they report; a child that
    fails, prints nothing or hangs becomes an entry that says so, the budget stops further starts, the environment of the
    no-mirror run reaches its child -- driven here with a stub in place of bench.py"""
    stub = _stub(tmp_path, """
        import argparse, json, os, sys, time
        ap = argparse.ArgumentParser()
        ap.add_argument('--workload'); ap.add_argument('--nf'); ap.add_argument('--steps', type=int, default=5)
        ap.add_argument('--warmup', type=int, default=3)
        ap.add_argument('--no-cpu-baseline', action='store_true'); ap.add_argument('--no-other-workloads', action='store_true')
        a = ap.parse_args()
        assert a.no_cpu_baseline and a.no_other_workloads
        if a.workload == 'c2':
            sys.exit(3)
        if a.workload == 'c5':
            print('no json here'); sys.exit(0)
        k = dict(fringe_ant_fwd_kernel=dict(total_ms=10.0 * a.steps, frac=0.4, useful_frac_of_pipe_peak=0.1),
                 fringe_ant_bwd_kernel=dict(total_ms=12.0 * a.steps, frac=0.5, useful_frac_of_pipe_peak=0.12),
                 reduce_vis_kernel=dict(total_ms=0.1, frac=0.01))
        print(json.dumps(dict(ms_per_step=23.0, value=1e8, unit='vis/s', steps=a.steps, warmup=a.warmup, roofline=dict(kernels=k),
                              config=dict(workload='stub ' + a.workload, antenna_mirror_groups=None if os.environ.get('RIME_MIRROR') == '0' else [[7, 8]],
                                          antenna_pair_blocks=None if '0' in (os.environ.get('RIME_MIRROR'), os.environ.get('RIME_PAIR')) else [[63, 64, 1]]))))
        """)
    res = bench.other_workloads(60.0, script=stub)
    by = {r['workload']: r for r in res}
    assert set(by) == {'c3', 'c2', 'c5 [--nf 64]', 'c4 [RIME_MIRROR=0]', 'c4 [RIME_PAIR=0]', 'c4 [--nf 32]'}
    assert by['c3']['ms_per_step'] == 23.0 and by['c3']['kernels']['fwd'] == dict(kernel='fringe_ant_fwd_kernel', frac=0.4,
                                                                                useful_frac_of_pipe_peak=0.1, ms_per_step=10.0)
    assert by['c3']['kernels']['bwd']['ms_per_step'] == 12.0 and by['c3']['mirror_groups'] == [[7, 8]]
    assert by['c2'] == dict(workload='c2', failed=3) and by['c5 [--nf 64]'] == dict(workload='c5 [--nf 64]', failed=0)
    assert by['c4 [RIME_MIRROR=0]']['mirror_groups'] is None            # the switch reached the child
    assert by['c3']['pair_blocks'] == [[63, 64, 1]] and by['c4 [RIME_PAIR=0]']['pair_blocks'] is None
    assert by['c4 [RIME_PAIR=0]']['mirror_groups'] == [[7, 8]]
    res = bench.other_workloads(5.0, script=stub)                     # no budget: nothing is started
    assert all('skipped' in r for r in res) and len(res) == 6


def test_rank_process_refuses_a_world_size_other_than_gpus():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, timeout=300)
    assert r.returncode != 0 and b'WORLD_SIZE=1' in r.stderr and r.stdout.strip() == b''


# ---------------------------------------------------------------------------------------------------------------------
# N > 1: every rank supervises one fresh worker per shard mode: a mode that crashes, hangs or fails
# its self-check must not lose the other mode's finished result
# ---------------------------------------------------------------------------------------------------------------------
WORKER_STUB = """
    import argparse, json, os, sys, time
    ap = argparse.ArgumentParser()
    ap.add_argument('--shard', action='append')
    ap.add_argument('--worker', action='store_true')
    ap.add_argument('--behave', default='')
    a = ap.parse_args()
    mode = a.shard[-1]                                   # the supervisor's --shard comes last and wins
    assert a.worker and os.environ['BENCH_PG_TAG'].startswith(mode)
    behave = dict(kv.split('=') for kv in a.behave.split(',') if kv)
    what = behave.get(mode, 'ok')
    if what == 'crash':
        sys.exit(7)
    if what.startswith('late'):                          # waits at its rendezvous for a peer held by the previous mode
        time.sleep(float(what[4:]))
        open(os.environ['BENCH_READY_FILE'], 'w').close()
    if what == 'hang':
        time.sleep(600)
    ms = dict(freq=10.0, bl=30.0)[mode] if 'blfast' not in behave else dict(freq=30.0, bl=10.0)[mode]
    if int(os.environ['RANK']) == 0:
        print('chatter')
        print(json.dumps(dict(metric='m', value=1000.0 / ms, ms_per_step=ms, n_gpus=int(os.environ['WORLD_SIZE']),
                              config=dict(parallelism=mode + ' x2'),
                              dist=dict(shard=mode, tile_plan_load=None, selfcheck=dict(ok=what != 'badcheck', vis_relmax=1e-7)))))
    if what == 'badcheck':
        sys.exit(5)
    """


def _supervise(tmp_path, behave, timeout=60):
    import io
    stub = _stub(tmp_path, WORKER_STUB)
    out = io.StringIO()
    rc = bench.supervise_modes(['freq', 'bl'], [stub, '--shard', 'auto', '--behave', behave], 0, 1, timeout, out=out,
                               env=dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29999'))
    lines = out.getvalue().splitlines()
    assert len(lines) <= 1                                   # never more than ONE result line
    return rc, (json.loads(lines[0]) if lines else None)


def test_supervisor_merges_both_modes(tmp_path):
    rc, res = _supervise(tmp_path, '')
    assert rc == 0 and res['dist']['shard'] == 'freq' and res['ms_per_step'] == 10.0
    assert res['alt'] == [dict(shard='bl', parallelism='bl x2', ms_per_step=30.0, value=1000.0 / 30.0, tile_plan_load=None,
                               selfcheck=dict(ok=True, vis_relmax=1e-7))]
    rc, res = _supervise(tmp_path, 'blfast=1')
    assert rc == 0 and res['dist']['shard'] == 'bl' and res['alt'][0]['shard'] == 'freq'     # the faster mode is the line


def test_supervisor_keeps_the_first_result_when_the_second_mode_crashes(tmp_path):
    rc, res = _supervise(tmp_path, 'bl=crash')
    assert rc == 0 and res['dist']['shard'] == 'freq' and res['alt'] == [dict(shard='bl', failed=7)]


def test_supervisor_keeps_the_first_result_when_the_second_mode_hangs(tmp_path):
    rc, res = _supervise(tmp_path, 'bl=hang', timeout=8)
    assert rc == 0 and res['dist']['shard'] == 'freq' and res['alt'] == [dict(shard='bl', failed=124)]


def test_supervisor_starts_a_modes_limit_at_its_rendezvous_not_at_its_start(tmp_path):
    """a rank whose first-mode worker aborted early (here: crashed at once) starts the second mode minutes before
    a peer that is still held by the first; its second worker then WAITS at the rendezvous.  That wait must not count
    against the mode's limit: with an 8-s limit the worker that becomes ready after 12 s and then finishes is kept (the
    old code killed it at 8 s); one that never becomes ready is still killed -- after limit + the time a peer can lag"""
    rc, res = _supervise(tmp_path, 'freq=crash,bl=late12', timeout=8)
    assert rc == 0 and res['dist']['shard'] == 'bl' and res['alt'] == [dict(shard='freq', failed=7)]
    import time
    t0 = time.perf_counter()
    rc, res = _supervise(tmp_path, 'freq=crash,bl=hang', timeout=4)
    assert rc != 0 and res is None and time.perf_counter() - t0 < 4 + (4 + 30) + 25


def test_supervisor_reports_the_second_mode_when_the_first_fails(tmp_path):
    rc, res = _supervise(tmp_path, 'freq=crash')
    assert rc == 0 and res['dist']['shard'] == 'bl' and res['alt'] == [dict(shard='freq', failed=7)]


def test_supervisor_treats_a_failed_selfcheck_as_a_failed_mode(tmp_path):
    rc, res = _supervise(tmp_path, 'bl=badcheck')
    assert rc == 0 and res['dist']['shard'] == 'freq'
    assert res['alt'] == [dict(shard='bl', failed=5, selfcheck=dict(ok=False, vis_relmax=1e-7))]
    rc, res = _supervise(tmp_path, 'freq=badcheck,bl=badcheck')
    assert rc == 5 and res['failed'] == [dict(shard='freq', failed=5), dict(shard='bl', failed=5)]


def test_supervisor_fails_when_every_mode_fails(tmp_path):
    rc, res = _supervise(tmp_path, 'freq=crash,bl=crash')
    assert rc == 7 and res is None


def test_two_ranks_two_modes_each_mode_its_own_process_group_on_the_launcher_store(tmp_path):
    """end to end on the CPU: torch.distributed.run starts two rank processes (bench.launch_ranks), each supervises one
    fresh gloo worker per mode; the workers of a mode rendezvous through bench._init_process_group (a PrefixStore per mode
    on the launcher's TCPStore) and all-reduce; the second mode's rank-1 worker dies before its collective: rank 0's
    worker of that mode fails on its watchdog, the first mode's line is relayed with the failure under `alt`"""
    worker = tmp_path / 'gloo_worker.py'
    worker.write_text(textwrap.dedent("""
        import argparse, datetime, json, os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        ap = argparse.ArgumentParser()
        ap.add_argument('--shard', action='append')
        ap.add_argument('--worker', action='store_true')
        ap.add_argument('--die', default='')
        a = ap.parse_args()
        mode, rank, world = a.shard[-1], int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        assert os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
        bench._init_process_group('gloo', rank, world, None, datetime.timedelta(seconds=60))
        if mode == a.die and rank == 1:
            os._exit(9)
        bench._set_collective_timeout(10)
        x = torch.tensor([float(rank + 1) * (1 if mode == 'freq' else 10)])
        dist.all_reduce(x)
        if rank == 0:
            print(json.dumps(dict(metric='m', value=float(x), ms_per_step=dict(freq=2.0, bl=1.0)[mode], n_gpus=world,
                                  config=dict(parallelism=mode), dist=dict(shard=mode, tile_plan_load=None, selfcheck=None))))
        dist.destroy_process_group()
        """ % ROOT))
    rank_script = tmp_path / 'rank.py'
    rank_script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import bench
        rc = bench.supervise_modes(['freq', 'bl'], [%r] + sys.argv[1:], int(os.environ['RANK']), int(os.environ['WORLD_SIZE']),
                                   120, out=sys.stdout)
        sys.exit(rc)
        """ % (ROOT, str(worker))))
    rc, line = bench.launch_ranks(2, [str(rank_script)], timeout=300)
    assert rc == 0, line
    res = json.loads(line)
    assert res['dist']['shard'] == 'bl' and res['value'] == 30.0                 # both modes ran: sums 1 + 2 and 10 + 20
    assert res['alt'][0]['shard'] == 'freq' and res['alt'][0]['value'] == 3.0
    rc, line = bench.launch_ranks(2, [str(rank_script), '--die', 'bl'], timeout=300)
    assert rc == 0, line
    res = json.loads(line)
    assert res['dist']['shard'] == 'freq' and res['value'] == 3.0
    assert res['alt'][0]['shard'] == 'bl' and res['alt'][0]['failed'] != 0
