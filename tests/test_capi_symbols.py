"""
CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports
every symbol include/rime_hip.h declares (no compute is launched here).
"""
import os
import re
import ctypes

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'rime_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(rime_[a-z0-9_]+)\s*\(', txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from bayeslim_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(_lib.lib, s), 'librime_hip.so does not export ' + s
        assert s in _lib.SIGNATURES, 'no ctypes signature for ' + s
    assert set(_lib.SIGNATURES) == set(syms)
    assert _lib.version().startswith('rime_hip') and 'gfx950' in _lib.version()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    from bayeslim_amd import _lib
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    try:
        _lib._load()
    except _lib.RimeLibraryError as e:
        assert 'no CPU fallback' in str(e)
    else:
        raise AssertionError('expected RimeLibraryError')


def test_ops_have_no_cpu_path():
    import torch
    import pytest
    from bayeslim_amd import ops
    with pytest.raises(RuntimeError):
        ops.FringeGeometry(torch.zeros(2, 3), torch.zeros(1, 3, 64), torch.linspace(1e8, 2e8, 4))


def test_bad_arguments_are_rejected_without_launching():
    """argument validation happens before any HIP call, so it can run on the CPU box"""
    from bayeslim_amd._lib import lib
    off = (ctypes.c_int * 2)(0, 4)
    one = ctypes.c_void_p(8)      # non-null dummy; never dereferenced on a rejected call
    rc = lib.rime_fringe_sum_fwd(0, one, one, one, one, off, None, 4, 1, 8, 100, 1, 1, 0, 1,
                                 1, 1e8, 1e6, 10.0, None, one, one, 0, None)
    assert rc == -1                # Pstride not a multiple of 64
    rc = lib.rime_fringe_sum_fwd(0, one, one, one, one, off, None, 4, 1, 8, 128, 1, 3, 0, 1,
                                 1, 1e8, 1e6, 10.0, None, one, one, 0, None)
    assert rc == -1                # Npp = 3
    rc = lib.rime_interp_gather_fwd(0, 0, one, one, one, 2, 10, 5, 4, one, 3, None)
    assert rc == -1                # out_stride < P
    # workspace = S partial slabs of the vis tensor (forward) -- 0 when the grid is already large
    vis_bytes = 8128 * 64 * 256 * 8
    assert lib.rime_fringe_sum_workspace(0, 8128, 64, 256, 108032, 1, 1, 0, 0) == 0
    ws = lib.rime_fringe_sum_workspace(0, 8128, 4, 256, 108032, 1, 1, 0, 0)
    assert ws % (8128 * 4 * 256 * 8) == 0 and ws > 0
    assert lib.rime_fringe_sum_workspace(0, 3, 2, 33, 9024, 1, 1, 0, 0) > 0


@pytest.mark.parametrize('src', ['alm.hip', 'fringe_mfma.hip'])
def test_no_packed_f32_reader_close_to_an_mfma(src, tmp_path):
    """regression guard for the round-2 defect (rime_common.h, RIME_MFMA_SETTLE): in the gfx950 assembly of the
    matrix-core sources no v_pk_*_f32 instruction reads an MFMA destination register within 24 wait states of the MFMA
    (the failing kernel had 46 such readers, the closest at 13)"""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = str(tmp_path / (src + '.s'))
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=fast', '-S', '--cuda-device-only',
                    os.path.join(root, 'bayeslim_amd', 'csrc', src), '-o', asm], check=True, capture_output=True)
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'scan_packed_readers.py'), asm, '24'],
                         check=True, capture_output=True, text=True).stdout
    assert out.startswith('no packed-f32 reader'), out
