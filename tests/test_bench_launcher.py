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


def test_rank_process_refuses_a_world_size_other_than_gpus():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, timeout=300)
    assert r.returncode != 0 and b'WORLD_SIZE=1' in r.stderr and r.stdout.strip() == b''
