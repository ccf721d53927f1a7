"""Lab: time rime_alm2pix_fwd of a given build of the library (ablation variants of the forward kernel built with
-DRIME_ALM_ABL=1 no split / 2 no A staging / 3 no MFMA; results are wrong by construction, timing only).
usage: python tools/alm_ablate.py [lib.so ...]"""
import ctypes, os, sys
import torch

R, Nc, Npix = 128, 8385, 49152
libs = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bayeslim_amd', 'lib', 'librime_hip.so')]
a = torch.randn(R, Nc, dtype=torch.complex64, device='cuda')
Y = torch.randn(Nc, Npix, dtype=torch.complex64, device='cuda')
out = torch.empty(R, Npix, dtype=torch.float32, device='cuda')
for path in libs:
    lib = ctypes.CDLL(path)
    lib.rime_alm2pix_fwd_workspace.restype = ctypes.c_size_t
    lib.rime_alm2pix_fwd_workspace.argtypes = [ctypes.c_int] * 4
    lib.rime_alm2pix_fwd.restype = ctypes.c_int
    lib.rime_alm2pix_fwd.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    nb = lib.rime_alm2pix_fwd_workspace(0, R, Nc, Npix)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device='cuda')
    ys = 2.0 ** 10

    def run():
        rc = lib.rime_alm2pix_fwd(0, a.data_ptr(), Y.data_ptr(), ys, R, Nc, Npix, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print('%-40s fwd op %.3f ms (median of 7, min %.3f)' % (os.path.basename(path), ts[3], ts[0]), flush=True)
