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
    # round-4 entry points
    assert lib.rime_stokes2coh_fwd(0, one, one, 0, 0, 0, 0, 16, one, None) == -1          # R = 0
    assert lib.rime_stokes2coh_fwd(7, one, one, 0, 0, 0, 4, 16, one, None) == -1          # unknown dtype
    assert lib.rime_stokes2coh_bwd(0, one, None, 0, 0, 0, 4, 16, one, None) == -1         # no fractions
    assert lib.rime_stokes2coh_fwd(0, one, one, -1, 0, 0, 4, 16, one, None) == -1         # negative stride
    assert lib.rime_fringe_row_scale_cplx(one, 1, 1, 1, 0, 0, 0, 0, 0, 64, one, None, None, None) == -1    # no rows
    assert lib.rime_fringe_row_scale_cplx(None, 1, 1, 1, 1, 0, 0, 0, 0, 64, one, None, None, None) == -1
    assert lib.rime_interp_scatter_rows_bwd(0, 0, one, 0, one, one, one, 4, 10, one, None) == -1           # row stride 0
    assert lib.rime_interp_scatter_rows_bwd(5, 0, one, 16, one, one, one, 4, 10, one, None) == -1          # unknown dtype
    # round 5: the mirror mask of the block entry points -- a bit at or beyond ceil(Nrows / 16), or a negative mask
    big = 1 << 20
    assert lib.rime_fringe_ant_fwd_block(one, 40, 0, 8, one, one, one, one, None, one, one, 1, 1, 1, 64, 64, 64, 1, 1, 0, one, big, None) == -1
    assert lib.rime_fringe_ant_fwd_block(one, 40, 0, -1, one, one, one, one, None, one, one, 1, 1, 1, 64, 64, 64, 1, 1, 0, one, big, None) == -1
    assert lib.rime_fringe_ant_bwd_block(one, 128, 0, 256, one, one, one, one, one, 1, 1, 1, 64, 64, 64, 1, 1, 0, 0, one, one, big, None) == -1
    # round 5: conjugate-pair blocks -- at most 64 rows, a pixel stride of 1 or 2, a workspace
    assert lib.rime_fringe_pair_fwd_block(one, 65, None, 0, one, one, one, one, None, one, one, 1, 1, 1, 64, 64, 64, 1, 1, one, big, None) == -1
    assert lib.rime_fringe_pair_fwd_block(one, 64, None, 1, one, one, one, one, None, one, one, 1, 1, 1, 64, 64, 64, 3, 1, one, big, None) == -1
    assert lib.rime_fringe_pair_fwd_block(one, 64, None, 1, one, one, one, one, None, None, one, 1, 1, 1, 64, 64, 64, 1, 1, one, big, None) == -1
    assert lib.rime_fringe_pair_fwd_block(one, 64, None, 1, one, one, one, one, None, one, one, 1, 1, 1, 64, 64, 64, 1, 1, one, 4, None) == -2
    assert lib.rime_fringe_pair_bwd_block(one, 0, None, 0, one, one, one, one, one, 1, 1, 1, 64, 64, 64, 1, 1, 0, one, one, big, None) == -1
    assert lib.rime_fringe_pair_bwd_block(one, 64, None, 1, one, one, one, one, one, 1, 1, 1, 64, 64, 64, 1, 1, 0, one, one, 0, None) == -2
    # 16 384 pixels per forward block since round 4: the C4 diffuse launch (98 304 px, 8 x 256 rows) takes 6 slabs
    assert lib.rime_fringe_ant_workspace(8128, 8, 256, 98304) == 6 * 8128 * 8 * 256 * 8
    # workspace = S partial slabs of the vis tensor (forward) -- 0 when the grid is already large
    vis_bytes = 8128 * 64 * 256 * 8
    assert lib.rime_fringe_sum_workspace(0, 8128, 64, 256, 108032, 1, 1, 0, 0) == 0
    ws = lib.rime_fringe_sum_workspace(0, 8128, 4, 256, 108032, 1, 1, 0, 0)
    assert ws % (8128 * 4 * 256 * 8) == 0 and ws > 0
    assert lib.rime_fringe_sum_workspace(0, 3, 2, 33, 9024, 1, 1, 0, 0) > 0


def test_build_scan_refuses_packed_f32_in_the_pair_kernels(tmp_path):
    """round 5: kernels whose blocks share a CU (the conjugate-pair kernels) must hold no v_pk_{add,mul,fma}_f32 at all
    (csrc/fringe_mfma.hip, keep_scalar); tools/scan_packed_readers.py --no-packed=... is what the Makefile runs -- here on two
    synthetic listings, and the record of the shipped build names the kernels it covered"""
    import subprocess, sys
    tool = os.path.join(ROOT, 'tools', 'scan_packed_readers.py')
    body = '_ZN4rime22fringe_pair_fwd_kernelILb1EEEvv:\n\tv_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]\n\ts_nop 15\n\ts_nop 15\n\t%s\n\ts_endpgm\n.Lfunc_end0:\n'
    bad, good = tmp_path / 'bad.s', tmp_path / 'good.s'
    bad.write_text(body % 'v_pk_add_f32 v[40:41], v[42:43], v[44:45]')
    good.write_text(body % 'v_add_f32_e32 v40, v42, v44')
    r = subprocess.run([sys.executable, tool, str(bad), '24', '--fail', '--no-packed=fringe_pair_'], capture_output=True, text=True)
    assert r.returncode == 1 and 'packed f32 instructions in a kernel that must have none' in r.stdout
    r = subprocess.run([sys.executable, tool, str(good), '24', '--fail', '--no-packed=fringe_pair_'], capture_output=True, text=True)
    assert r.returncode == 0 and 'no packed f32 instruction in the 1 kernels' in r.stdout
    r = subprocess.run([sys.executable, tool, str(bad), '24', '--fail'], capture_output=True, text=True)      # the old rule alone: far from the MFMA
    assert r.returncode == 0
    rec = open(os.path.join(ROOT, 'bayeslim_amd', 'lib', 'obj', 'fringe_mfma.scan')).read()
    assert 'no packed f32 instruction in the 18 kernels named *fringe_pair_*' in rec, rec


@pytest.mark.parametrize('src', ['alm', 'fringe_mfma'])
def test_no_packed_f32_reader_close_to_an_mfma(src):
    """regression guard for the round-2 defect (rime_common.h, RIME_MFMA_SETTLE): the Makefile scans the gfx950 assembly
    of the SHIPPED objects (same compiler run, -save-temps) for v_pk_*_f32 readers of fresh MFMA results and fails the
    build on a hit; here: the scan record beside the objects belongs to this build and is clean"""
    import subprocess
    obj = os.path.join(ROOT, 'bayeslim_amd', 'lib', 'obj')
    if not os.path.exists(os.path.join(obj, src + '.scan')):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'bayeslim_amd', 'csrc')], check=True, capture_output=True)
    rec = open(os.path.join(obj, src + '.scan')).read()
    assert rec.startswith('no packed-f32 reader') and 'within 24 wait states' in rec, rec
    assert os.path.getmtime(os.path.join(obj, src + '.scan')) >= os.path.getmtime(os.path.join(obj, src + '.o'))


def test_product_kernels_carry_no_laboratory_code_and_no_scratch():
    """round 5: the product source of the matrix-core fringe kernels holds no RIME_LAB_* / RIME_ABL_* /
    lab-only build switch, the shipped library no default-off kernel (pipelined backward, second forward form), and the
    gfx950 assembly of THIS build of the two matrix-core sources uses no scratch memory (no spill in any kernel);
    the laboratory side is a patch that still applies (tools/lab/)"""
    import subprocess
    src = open(os.path.join(ROOT, 'bayeslim_amd', 'csrc', 'fringe_mfma.hip')).read()
    for word in ('RIME_LAB', 'RIME_ABL', 'RIME_BUILD_FWD_V2', 'RIME_PHASE_MAGIC', 'bwd_pipe', 'fwd_v2', '#if'):
        assert word not in src, word
    obj = os.path.join(ROOT, 'bayeslim_amd', 'lib', 'obj')
    for name in ('fringe_mfma', 'alm'):
        path = os.path.join(obj, name + '-hip-amdgcn-amd-amdhsa-gfx950.s')
        if not os.path.exists(path):
            subprocess.run(['make', '-C', os.path.join(ROOT, 'bayeslim_amd', 'csrc')], check=True, capture_output=True)
        asm = open(path).read()
        kernels = re.findall(r'\.amdhsa_kernel (\S+)', asm)
        assert len(kernels) >= 10
        assert not [k for k in kernels if 'pipe' in k or '_v2_' in k], kernels
        assert not re.findall(r'^\s*scratch_(?:load|store)', asm, flags=re.M)
        sizes = [int(x) for x in re.findall(r'\.amdhsa_private_segment_fixed_size (\d+)', asm)]
        assert len(sizes) == len(kernels) and max(sizes) == 0, sizes
    lib = open(os.path.join(ROOT, 'bayeslim_amd', 'lib', 'librime_hip.so'), 'rb').read()
    assert b'fringe_ant_bwd_pipe_kernel' not in lib and b'fringe_ant_fwd_v2_kernel' not in lib
    assert b'fringe_ant_bwd_kernel' in lib
    subprocess.run([os.path.join(ROOT, 'tools', 'lab', 'make_lab_source.sh'), '--check'], check=True, capture_output=True)


def test_packed_reader_scanner_finds_planted_hazards(tmp_path):
    """the scanner itself: a packed reader 9 wait states behind an MFMA is reported -- directly, through an
    accumulation-register copy, and across a loop back edge; the same reader behind an s_nop 15 margin is not"""
    import subprocess
    import sys
    tool = os.path.join(ROOT, 'tools', 'scan_packed_readers.py')
    filler = '\n'.join('\tv_add_f32 v100, v101, v102' for _ in range(8))

    def run(body):
        path = tmp_path / 'k.s'
        path.write_text('_Z6kernelv:\n' + body + '\n\ts_endpgm\n')
        r = subprocess.run([sys.executable, tool, str(path), '24', '--fail'], capture_output=True, text=True)
        return r.returncode, r.stdout

    mfma = '\tv_mfma_f32_32x32x16_f16 v[0:15], v[20:23], v[24:27], v[0:15]'
    rc, out = run(mfma + '\n' + filler + '\n\tv_pk_fma_f32 v[40:41], v[0:1], v[42:43], v[40:41] op_sel_hi:[1,0,1]')
    assert rc == 1 and 'closest 9' in out, out
    rc, out = run(mfma + '\n' + filler + '\n\ts_nop 15\n\tv_pk_fma_f32 v[40:41], v[0:1], v[42:43], v[40:41]')
    assert rc == 0 and out.startswith('no packed-f32 reader'), out
    amfma = '\tv_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]'
    rc, out = run(amfma + '\n\tv_accvgpr_read_b32 v50, a3\n\tv_accvgpr_read_b32 v51, a4\n' + filler
                  + '\n\tv_pk_mul_f32 v[60:61], v[50:51], v[62:63]')
    assert rc == 1, out
    # loop: the reader sits at the top of the body, the MFMA at the bottom
    rc, out = run('.LBB0_1:\n\tv_pk_fma_f32 v[40:41], v[0:1], v[42:43], v[40:41]\n' + filler + '\n' + mfma
                  + '\n\ts_cbranch_scc1 .LBB0_1')
    assert rc == 1, out
    # a plain (non-packed) reader is the compiler's business, not reported
    rc, out = run(mfma + '\n\tv_fma_f32 v40, v0, v42, v40')
    assert rc == 0, out


def test_integration_md_stubs_match_the_signatures():
    """the reference-side ctypes stubs shown in INTEGRATION.md carry the argument lists of bayeslim_amd/_lib.py
    (one stub was an argument short): every `_lib.<fn>.argtypes = [...]` statement in the document is
    executed against a recording object and compared with _lib.SIGNATURES"""
    from bayeslim_amd import _lib
    txt = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', txt, flags=re.S)
    stmts = []
    for b in blocks:
        # statements may span lines: take from `_lib.x.argtypes =` / `.restype =` to the closing bracket / end of line
        for m in re.finditer(r'^_lib\.(rime_\w+)\.(argtypes|restype)\s*=\s*', b, flags=re.M):
            rest = b[m.end():]
            if m.group(2) == 'restype':
                expr = rest.split('\n', 1)[0]
            else:
                depth, end = 0, None
                for k, ch in enumerate(rest):
                    depth += ch in '[(' 
                    depth -= ch in '])'
                    if depth == 0 and ch in '])':
                        end = k + 1
                        break
                expr = rest[:end]
            stmts.append((m.group(1), m.group(2), expr))
    assert any(kind == 'argtypes' for _, kind, _ in stmts), 'no argtypes stub found in INTEGRATION.md'

    class Fn:
        pass

    class Lib:
        def __getattr__(self, name):
            fn = Fn()
            object.__setattr__(self, name, fn)
            return fn

    lib = Lib()
    ns = dict(ctypes=ctypes, _lib=lib, _vp=ctypes.c_void_p, _i=ctypes.c_int, _d=ctypes.c_double, list=list)
    for name, kind, expr in stmts:
        exec('_lib.%s.%s = %s' % (name, kind, re.sub(r'#.*', '', expr)), ns)
    checked = 0
    for name, kind, _ in stmts:
        res, args = _lib.SIGNATURES[name]
        if kind == 'restype':
            assert getattr(lib, name).restype is res, name
        else:
            got = list(getattr(lib, name).argtypes)
            assert len(got) == len(args), '%s: INTEGRATION.md lists %d arguments, the ABI has %d' % (name, len(got), len(args))
            assert all(g is a for g, a in zip(got, args)), name
            checked += 1
    assert checked >= 2


def test_header_argument_counts_match_the_signatures():
    """include/rime_hip.h and bayeslim_amd/_lib.py agree on the NUMBER of arguments of every entry point"""
    from bayeslim_amd import _lib
    txt = open(os.path.join(ROOT, 'include', 'rime_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'//[^\n]*', '', txt)
    seen = 0
    for m in re.finditer(r'\b(rime_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', txt):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ('', 'void') else params.count(',') + 1
        assert n == len(_lib.SIGNATURES[name][1]), '%s: header %d arguments, _lib.py %d' % (name, n, len(_lib.SIGNATURES[name][1]))
        seen += 1
    assert seen == len(_lib.SIGNATURES)
