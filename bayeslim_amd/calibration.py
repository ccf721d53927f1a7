"""
Gain application on the GPU: the post-RIME calibration step of the forward model, mirroring
calibration.apply_cal / _apply_cal (calibration.py:2330-2487) for complex visibilities and
Jones-type gains -- SURVEY section 8(f) item 3.  The products run in the fused HIP kernels behind
ops.apply_cal (one pass over the visibility tensor, forward and backward); there is no CPU
implementation.  Not mirrored here (left in torch upstream of this call): undo (gain inversion),
delay-type visibilities ('dly') and covariance propagation.
"""
import torch

from . import ops


def _index_tensor(idx, device):
    if torch.is_tensor(idx) and idx.dtype == torch.int32 and idx.device == device:
        return idx
    return torch.as_tensor(idx, device=device).to(torch.int32).contiguous()


def _apply_cal(vis, gains, g1_idx, g2_idx, cal_2pol=False, cov=None, vis_type='com', undo=False, inplace=False):
    """
    vis (Npol, Npol, Nbl, Ntimes, Nfreqs) complex; gains (Npol, Npol, Nant, Ntimes | 1, Nfreqs | 1);
    g1_idx / g2_idx: len-Nbl indices into the antenna axis of gains for the two antennas of each
    baseline (int32 GPU tensors are used as they are -- build them once; anything else is converted per
    call).  Returns (new_vis, cov) like the reference; cov is passed through untouched and must be None.
    """
    assert vis.shape[:2] == gains.shape[:2], "vis and gains must have same Npols"
    if vis_type != 'com':
        raise NotImplementedError("only complex visibilities ('com') run on the fused kernels")
    if undo:
        raise NotImplementedError('invert the gains before the call (undo is not fused)')
    if cov is not None:
        raise NotImplementedError('covariance propagation is not fused')
    a1 = _index_tensor(g1_idx, vis.device)
    a2 = _index_tensor(g2_idx, vis.device)
    polmode = '1pol' if vis.shape[:2] == (1, 1) else '4pol'
    if cal_2pol and polmode == '4pol':
        polmode = '2pol'
    # inplace: the reference rebinds its output for complex visibilities, so the flag never changes vis there
    vout = ops.apply_cal(vis, gains, a1, a2, diag=(polmode == '2pol'))
    return vout, cov


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
