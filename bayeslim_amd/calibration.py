"""
Gain application on the GPU: the post-RIME calibration step of the forward model, mirroring
calibration.apply_cal / _apply_cal (calibration.py:2330-2487) -- SURVEY section 8(f) item 3.  The
products of complex visibilities with Jones-type gains run in the fused HIP kernels behind
ops.apply_cal (one pass over the visibility tensor, forward and backward); there is no CPU
implementation.  The other branches of the reference function are small-tensor or elementwise
arithmetic and stay torch ops on the GPU: `undo` inverts the (small) gain tensor first (1-pol / 2-pol:
reciprocal of the diagonal, linalg.diag_inv; 4-pol: the 2 x 2 inverse per antenna, time and channel --
the reference's `torch.pinv` call at :2441 does not exist in torch, so that branch cannot run there),
`vis_type='dly'` adds delay differences, `cov` propagates a variance tensor with |g1 g2*|^2.
"""
import torch

from . import ops


def _index_tensor(idx, device):
    if torch.is_tensor(idx) and idx.dtype == torch.int32 and idx.device == device:
        return idx
    return torch.as_tensor(idx, device=device).to(torch.int32).contiguous()


def _invert_gains(gains, polmode, vis_type):
    """the `undo` branch (calibration.py:2430-2443) without the per-antenna Python loop"""
    if polmode in ('1pol', '2pol'):
        if vis_type == 'dly':
            return -gains
        if gains.shape[0] == 1:
            return 1 / gains
        inv = torch.zeros_like(gains)                      # linalg.diag_inv: off-diagonals dropped
        inv[0, 0] = 1 / gains[0, 0]
        inv[1, 1] = 1 / gains[1, 1]
        return inv
    assert vis_type == 'com', 'must have complex vis_type for 4pol mode'
    a, b, c, d = gains[0, 0], gains[0, 1], gains[1, 0], gains[1, 1]
    det = a * d - b * c
    return torch.stack([torch.stack([d / det, -b / det]), torch.stack([-c / det, a / det])])


def _apply_cal(vis, gains, g1_idx, g2_idx, cal_2pol=False, cov=None, vis_type='com', undo=False, inplace=False):
    """
    vis (Npol, Npol, Nbl, Ntimes, Nfreqs); gains (Npol, Npol, Nant, Ntimes | 1, Nfreqs | 1);
    g1_idx / g2_idx: len-Nbl indices into the antenna axis of gains for the two antennas of each
    baseline (int32 GPU tensors are used as they are -- build them once; anything else is converted per
    call).  Returns (new_vis, new_cov) like the reference (calibration.py:2412-2487).
    """
    assert vis.shape[:2] == gains.shape[:2], "vis and gains must have same Npols"
    if not vis.is_cuda:
        raise RuntimeError('bayeslim_amd.calibration needs tensors on the GPU (no CPU implementation)')
    polmode = '1pol' if vis.shape[:2] == (1, 1) else '4pol'
    if cal_2pol and polmode == '4pol':
        polmode = '2pol'
    if undo:
        gains = _invert_gains(gains, polmode, vis_type)
    a1 = _index_tensor(g1_idx, vis.device)
    a2 = _index_tensor(g2_idx, vis.device)
    cov_out = cov
    if vis_type == 'dly':
        assert polmode in ('1pol', '2pol')
        # float delays: V_out = V + tau_1 - tau_2 (:2476)
        return vis + gains.index_select(2, a1.long()) - gains.index_select(2, a2.long()), cov_out
    assert vis_type == 'com'
    # inplace: the reference rebinds its output for complex visibilities, so the flag never changes vis there
    vout = ops.apply_cal(vis, gains.to(vis.dtype) if gains.dtype != vis.dtype else gains, a1, a2,
                         diag=(polmode == '2pol'))
    if cov is not None:
        # variance of the same shape as vis, 1-pol / 2-pol only: cov * |g1 g2*|^2 on the diagonal (:2466-2471)
        assert polmode in ('1pol', '2pol'), 'covariance propagation: 1pol or 2pol mode'
        G = gains.index_select(2, a1.long()) * gains.index_select(2, a2.long()).conj()
        GG = (G * G.conj()).real if torch.is_complex(G) else G * G
        cov_out = torch.zeros_like(cov)
        for p in range(cov.shape[0]):
            cov_out[p, p] = GG[p, p] * cov[p, p]
    return vout, cov_out


def apply_cal(vis, bls, gains, ants, cal_2pol=False, cov=None, vis_type='com', undo=False, inplace=False):
    """
    calibration.apply_cal (calibration.py:2330-2410): bls list of (ant1, ant2), ants list of antenna
    numbers along gains' antenna axis.  Builds the index tensors and calls _apply_cal.
    """
    where = {a: i for i, a in enumerate(ants)}
    g1_idx = torch.as_tensor([where[bl[0]] for bl in bls], dtype=torch.int32, device=vis.device)
    g2_idx = torch.as_tensor([where[bl[1]] for bl in bls], dtype=torch.int32, device=vis.device)
    return _apply_cal(vis, gains, g1_idx, g2_idx, cal_2pol=cal_2pol, cov=cov, vis_type=vis_type,
                      undo=undo, inplace=inplace)
