"""
ctypes binding of librime_hip.so -- the C-ABI boundary declared in include/rime_hip.h.

The library is built in-tree (bayeslim_amd/lib/librime_hip.so) by `__graft_entry__.build()`
or `make -C bayeslim_amd/csrc`.  There is NO fallback: if the shared object is missing or a
symbol cannot be resolved, importing this module raises.
"""
import ctypes
import os

# torch FIRST: it ships its own HIP runtime (torch/lib/libamdhip64.so); loaded before this library, the library's HIP
# symbols resolve to that copy and both share one runtime.  The other order (this library first, pulling /opt/rocm's copy,
# then torch) leaves two runtimes in the process and every launch from here fails with "no ROCm-capable device is detected"
# (seen with __graft_entry__.build() followed by smoke() in ONE process).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RIME_LIB_PATH') or os.path.join(_HERE, 'lib', 'librime_hip.so')   # override: lab builds only

RIME_F32, RIME_F64 = 0, 1
ERRORS = {-1: 'RIME_EINVAL (bad shape/flag/null pointer)', -2: 'RIME_EWORKSPACE (workspace too small)',
          -3: 'RIME_ELAUNCH (kernel launch failed)', -4: 'RIME_EUNSUPPORTED'}

_vp, _i, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
_ip = ctypes.POINTER(ctypes.c_int)
_llp = ctypes.POINTER(ctypes.c_longlong)
_ll = ctypes.c_longlong

# symbol -> (restype, argtypes); mirrors include/rime_hip.h one to one
SIGNATURES = {
    'rime_version': (ctypes.c_char_p, []),
    'rime_last_error': (ctypes.c_char_p, []),
    'rime_fringe_sum_workspace': (_sz, [_i] * 9),
    'rime_fringe_sum_fwd': (_i, [_i, _vp, _vp, _vp, _vp, _ip, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                                 _i, _d, _d, _d, _llp, _vp, _vp, _sz, _vp]),
    'rime_fringe_sum_bwd': (_i, [_i, _vp, _vp, _vp, _vp, _ip, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                                 _i, _d, _d, _d, _llp, _vp, _vp, _sz, _vp]),
    'rime_fringe_ant_workspace': (_sz, [_i, _i, _i, _i]),
    'rime_fringe_ant_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _ll, _ll, _i, _vp, _vp, _sz, _vp]),
    'rime_fringe_ant_bwd_workspace': (_sz, [_i, _i, _i]),
    'rime_fringe_ant_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _ll, _ll, _i, _vp, _vp, _sz, _vp]),
    'rime_fringe_ant_fwd_block': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _i,
                                       _i, _vp, _sz, _vp]),
    'rime_fringe_row_scale': (_i, [_vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _i, _vp, _vp, _vp]),
    'rime_fringe_row_scale_cplx': (_i, [_vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _i, _vp, _vp, _vp, _vp]),
    'rime_fringe_ant_fwd_finish': (_i, [_vp, _sz, _vp, _i, _i, _i, _i, _vp]),
    'rime_fringe_ant_bwd_prepare': (_i, [_vp, _i, _i, _i, _vp, _sz, _vp]),
    'rime_fringe_ant_bwd_block': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _i,
                                       _i, _i, _vp, _vp, _sz, _vp]),
    'rime_fringe_pair_fwd_block': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _i,
                                        _vp, _sz, _vp]),
    'rime_fringe_pair_bwd_block': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _i, _i,
                                        _vp, _vp, _sz, _vp]),
    'rime_gen_fringe': (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'rime_comm_unique_id': (_i, [_vp]),
    'rime_comm_init': (_i, [ctypes.POINTER(ctypes.c_void_p), _i, _i, _vp]),
    'rime_comm_destroy': (_i, [_vp]),
    'rime_comm_allgather_vis': (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    'rime_comm_reduce_grads': (_i, [_vp, _i, _vp, _sz, _vp]),
    'rime_eq2top': (_i, [_vp, _vp, _i, _vp, _vp, _d, _vp, _vp, _vp]),
    'rime_interp_gather_fwd': (_i, [_i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'rime_interp_scatter_bwd': (_i, [_i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'rime_interp_scatter_rows_bwd': (_i, [_i, _i, _vp, _ll, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'rime_beam_sky_fwd': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'rime_beam_sky_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'rime_beam_sky_bwd': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'rime_jones_apply_fwd': (_i, [_i, _i, _vp, _vp, _vp, _ll, _ll, _vp, _vp]),
    'rime_jones_apply_bwd': (_i, [_i, _i, _vp, _vp, _vp, _vp, _ll, _ll, _vp, _vp, _vp, _vp]),
    'rime_stokes2coh_fwd': (_i, [_i, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _vp, _vp]),
    'rime_stokes2coh_bwd': (_i, [_i, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _vp, _vp]),
    'rime_chisq_workspace': (_sz, []),
    'rime_chisq_fwd': (_i, [_i, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    'rime_chisq_bwd': (_i, [_i, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'rime_apply_cal_fwd': (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _vp, _vp]),
    'rime_apply_cal_bwd': (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _vp, _vp, _vp, _vp]),
    'rime_alm2pix_fwd_workspace': (_sz, [_i, _i, _i, _i]),
    'rime_alm2pix_fwd': (_i, [_i, _vp, _vp, _d, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'rime_alm2pix_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'rime_alm2pix_bwd': (_i, [_i, _vp, _vp, _d, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'rime_alm2pix_packed_bytes': (_sz, [_i, _i, _i]),
    'rime_alm2pix_pack': (_i, [_vp, _d, _i, _i, _i, _vp, _vp]),
    'rime_alm2pix_fwd_packed': (_i, [_vp, _vp, _d, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'rime_alm2pix_bwd_packed': (_i, [_vp, _vp, _d, _i, _i, _i, _vp, _vp, _sz, _vp]),
}


class RimeLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise RimeLibraryError(
            "librime_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C bayeslim_amd/csrc`.  bayeslim_amd has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(code, what):
    if code != 0:
        msg = ERRORS.get(code, 'unknown error %d' % code)
        if code == -3:
            msg += ': ' + lib.rime_last_error().decode()
        raise RimeLibraryError('%s failed: %s' % (what, msg))


def version():
    return lib.rime_version().decode()
