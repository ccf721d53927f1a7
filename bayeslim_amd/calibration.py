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

`JonesModel` (calibration.py:416-742) is the module that follows RIME in a forward-model chain: parameters ->
`JonesResponse` (calibration.py:745-875: complex / amplitude / phase / delay / slope gain types, optional linear
bases over time and frequency) -> complex gains -> `_apply_cal` on the fused kernels, with the reference-antenna
phase convention (`rephase_to_refant`, :2490-2608) and the time index cache for minibatches (`IndexCache`, :291-413).
The redundant-calibration degeneracy projections of BaseResponse.setup_projection (abs_amp / phs_slope) and the
CalData export are outside the hot path and not built.
"""
import copy

import numpy as np
import torch

from . import ops, utils


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


# ---------------------------------------------------------------------------------------
# parameter <-> complex gain conversions (calibration.py:215-288)
# ---------------------------------------------------------------------------------------
def params2complex(params, param_type):
    if param_type == 'real':
        return params + 0j
    if param_type == 'amp':
        return torch.exp(params) + 0j
    if param_type == 'phs':
        return torch.exp(1j * params)
    if param_type == 'amp_phs':
        return torch.exp(params[..., 0] + 1j * params[..., 1])
    return params                                            # 'com' (and the types JonesResponse finishes)


def complex2params(data, param_type):
    if param_type == 'real':
        return data.real
    if param_type == 'amp':
        return torch.log(torch.abs(data))
    if param_type == 'phs':
        return torch.angle(data)
    if param_type == 'amp_phs':
        return torch.cat([data.abs().log()[..., None], data.angle()[..., None]], dim=-1)
    return data


def rephase_to_refant(params, param_type, refant_idx, p0=None, mode='rephase', inplace=False):
    """
    Reference-antenna phase convention on (Npol, Npol, Nant, Ntimes, Nfreqs[, 2]) parameters (calibration.py:2490-2608):
    'rephase' divides every antenna by the reference antenna's phasor of params + p0 ('com') or subtracts its
    phase / delay ('phs', 'dly', 'amp_phs'); 'zero' only zeroes the reference antenna's imaginary part / phase.
    """
    if refant_idx is None:
        return None
    if p0 is None:
        p0 = torch.zeros_like(params)
    if not inplace:
        params, p0 = copy.deepcopy(params), copy.deepcopy(p0)
    r = slice(refant_idx, refant_idx + 1)
    if mode == 'rephase':
        if param_type == 'com':
            real_view = not torch.is_complex(params)
            _p, _p0 = (utils.viewcomp(params), utils.viewcomp(p0)) if real_view else (params, p0)
            phasor = torch.exp(1j * torch.angle((_p + _p0)[:, :, r]).detach().clone())
            _p, _p0 = _p / phasor, _p0 / phasor
            params[:] = utils.viewreal(_p) if real_view else _p
            p0[:] = utils.viewreal(_p0) if real_view else _p0
        elif param_type in ('dly', 'phs'):
            params -= params[:, :, r].clone()
            p0 -= p0[:, :, r].clone()
        elif param_type == 'amp_phs':
            params[..., 1] -= params[:, :, r, ..., 1].clone()
            p0[..., 1] -= p0[:, :, r, ..., 1].clone()
    elif mode == 'zero':
        for t in (params, p0):
            if param_type == 'com':
                if torch.is_complex(t):
                    t.imag[:, :, r] = torch.zeros_like(t.imag[:, :, r])
                else:
                    t[:, :, r, ..., 1] = torch.zeros_like(t[:, :, r, ..., 1])
            elif param_type in ('dly', 'phs'):
                t[:, :, r] = torch.zeros_like(t[:, :, r])
            elif param_type == 'amp_phs':
                t[:, :, r, ..., 1] = torch.zeros_like(t[:, :, r, ..., 1])
    if not inplace:
        return params, p0


# ---------------------------------------------------------------------------------------
# response functions (calibration.py:11-212, 745-875)
# ---------------------------------------------------------------------------------------
class BaseResponse:
    """params (Npol, Npol, Nant | Nbl, Ntimes | Ncoeff, Nfreqs | Ncoeff) -> complex tensor over (Ntimes, Nfreqs):
    optional LM, complex view, linear bases along frequency / time, + base0, param-type conversion, projection"""
    def __init__(self, freq_mode='channel', time_mode='channel', param_type='com', device=None, freq_LM=None,
                 time_LM=None, freqs=None, times=None, LM=None, projection_kwargs={}, base0=None):
        self.freq_mode, self.time_mode, self.param_type = freq_mode, time_mode, param_type
        self.device = device
        self.freq_LM, self.time_LM, self.LM = freq_LM, time_LM, LM
        self.freqs, self.times = freqs, times
        self.setup_projection(**projection_kwargs)
        self.base0 = base0
        self._args = dict(freq_mode=freq_mode, time_mode=time_mode, param_type=param_type)

    def setup_projection(self, abs_amp_gain=False, phs_slope_gain=False, wgts_gain=None, refant_idx=None):
        if abs_amp_gain or phs_slope_gain:
            raise NotImplementedError('redundant-calibration degeneracy projections are not built')
        self._proj_refant_idx = refant_idx
        self._projection = refant_idx is not None

    def projection(self, params):
        if self._projection:
            i = self._proj_refant_idx
            params = params / torch.exp(1j * torch.angle(params[:, :, i:i + 1].detach()))
        return params

    def params2complex(self, params):
        return params2complex(params, self.param_type)

    def forward(self, params, **kwargs):
        if not utils.check_devices(params.device, self.device):
            params = params.to(self.device)
        if self.LM is not None:
            params = self.LM(params)
        if self.param_type == 'com' and not torch.is_complex(params):
            params = utils.viewcomp(params)
        if self.freq_mode == 'linear':
            params = self.freq_LM(params)
        if self.time_mode == 'linear':
            params = self.time_LM(params)
        if self.base0 is not None:
            params = params + self.base0
        params = self.projection(self.params2complex(params))
        if isinstance(params, torch.nn.Parameter):
            params = params.view(params.shape)
        return params

    __call__ = forward

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
        if self.base0 is not None:
            self.base0 = utils.push(self.base0, device)
        for lm in (self.LM, self.freq_LM if self.freq_mode == 'linear' else None,
                   self.time_LM if self.time_mode == 'linear' else None):
            if lm is not None and hasattr(lm, 'push'):
                lm.push(device)


class JonesResponse(BaseResponse):
    """gain types 'com', 'real', 'amp', 'phs', 'amp_phs', 'dly' [ns], and the array-gradient types 'dly_slope'
    [ns / m] and 'phs_slope' [rad / m] whose antenna axis holds (EW, NS) (calibration.py:745-875)"""
    def __init__(self, freq_mode='channel', time_mode='channel', param_type='com', vis_type='com', antpos=None,
                 device=None, freq_LM=None, time_LM=None, freqs=None, times=None, LM=None, base0=None):
        super().__init__(freq_mode=freq_mode, time_mode=time_mode, param_type=param_type, device=device,
                         freq_LM=freq_LM, time_LM=time_LM, LM=LM, base0=base0, freqs=freqs, times=times)
        self.vis_type, self.antpos = vis_type, antpos
        assert param_type in ['com', 'amp', 'phs', 'dly', 'real', 'amp_phs', 'phs_slope', 'dly_slope']
        if param_type in ('dly_slope', 'phs_slope'):
            assert antpos is not None, 'need antpos for dly_slope or phs_slope'
            self.antpos_EW = torch.as_tensor([float(antpos[a][0]) for a in antpos], device=device)[None, None, :, None, None]
            self.antpos_NS = torch.as_tensor([float(antpos[a][1]) for a in antpos], device=device)[None, None, :, None, None]
        if 'dly' in param_type:
            assert self.freqs is not None, 'need frequencies for delay gain type'

    def params2complex(self, jones):
        jones = super().params2complex(jones)
        if self.param_type == 'dly' and self.vis_type == 'com':
            return torch.exp(2j * np.pi * jones * torch.as_tensor(self.freqs / 1e9, dtype=jones.dtype, device=jones.device))
        if self.param_type in ('dly_slope', 'phs_slope'):
            tot = jones[:, :, :1] * self.antpos_EW + jones[:, :, 1:] * self.antpos_NS
            if self.param_type == 'phs_slope':
                return torch.exp(1j * tot)
            if self.vis_type == 'com':
                return torch.exp(2j * np.pi * tot * torch.as_tensor(self.freqs, device=tot.device) / 1e9)
            return tot
        return jones

    def push(self, device):
        super().push(device)
        if self.param_type in ('dly_slope', 'phs_slope') and not isinstance(device, torch.dtype):
            self.antpos_EW, self.antpos_NS = self.antpos_EW.to(device), self.antpos_NS.to(device)


class IndexCache:
    """time / baseline index caches for minibatched inputs of shape (..., Nbls, Ntimes, Nfreqs) (calibration.py:291-413)"""
    def __init__(self, times=None, bls=None, atol=1e-5):
        self._times, self._bls, self._atol = times, bls, atol
        self.clear_time_cache()
        self.clear_bl_cache()

    def clear_time_cache(self):
        self.cache_tidx = {}

    def clear_bl_cache(self):
        self.cache_bidx = {}

    def get_time_idx(self, times):
        if times is None or getattr(self, '_times', None) is None:
            return None
        h = utils.arr_hash(times)
        if h not in self.cache_tidx:
            ref = torch.as_tensor(self._times)
            idx = torch.cat([torch.where(torch.isclose(ref, torch.as_tensor(t, dtype=ref.dtype, device=ref.device),
                                                       atol=self._atol, rtol=1e-15))[0] for t in times])
            self.cache_tidx[h] = utils._list2slice(idx)
        return self.cache_tidx[h]

    def get_bl_idx(self, bls):
        if bls is None or getattr(self, '_bls', None) is None:
            return None
        h = utils.arr_hash(bls)
        if h not in self.cache_bidx:
            if isinstance(bls, list):
                idx = [self._bls.index(bl) for bl in bls]
            elif isinstance(bls, torch.Tensor):
                idx = torch.cat([torch.where(self._bls == bl)[0] for bl in bls])
            else:
                idx = np.concatenate([np.where(self._bls == bl)[0] for bl in bls])
            self.cache_bidx[h] = utils._list2slice(idx)
        return self.cache_bidx[h]

    def index_params(self, params, times=None, bls=None):
        for sel, axis in ((self.get_time_idx(times) if times is not None else None, -2),
                          (self.get_bl_idx(bls) if bls is not None else None, -3)):
            if sel is None:
                continue
            if isinstance(sel, slice) and (sel.stop - sel.start) // sel.step == params.shape[axis]:
                continue
            params = params[..., sel, :] if axis == -2 else params[..., sel, :, :]
        return params


class JonesModel(utils.Module, IndexCache):
    """
    Antenna-based, direction-independent Jones term V^d_pq = J_p V^m_pq J_q^dagger applied to a VisData
    (calibration.py:416-742): 1-pol (1, 1), 2-pol (diagonal) and 4-pol (full 2 x 2) parameters of shape
    (Npol, Npol, Nant, Ntimes | Ncoeff, Nfreqs | Ncoeff); forward(vd, undo=False, prior_cache=None, jones=None) -> VisData.
    """
    def __init__(self, params, ants, p0=None, refant=None, R=None, parameter=True, polmode='1pol', single_ant=False,
                 name=None, vis_type='com', atol=1e-5):
        utils.Module.__init__(self, name=name)
        self.params = torch.nn.Parameter(params) if parameter else params
        self.device = params.device
        self.p0 = p0
        self.ants = list(ants)
        self.Nants = len(self.ants)
        self.R = R if R is not None else JonesResponse()
        IndexCache.__init__(self, times=getattr(self.R, 'times', None), atol=atol)
        self.polmode, self.single_ant, self.vis_type = polmode, single_ant, vis_type
        self.set_refant(refant)
        self.clear_cache()
        self._args = dict(refant=refant, polmode=polmode)
        self._args[self.R.__class__.__name__] = getattr(self.R, '_args', None)

    def clear_cache(self):
        self.clear_time_cache()
        self.clear_bl_cache()
        self.cache_aidx = {}
        self._vd = None

    def clear_ant_cache(self):
        self.cache_aidx = {}

    def clear_vd_cache(self):
        self._vd = None

    def get_ant_idx(self, bls):
        """(g1_idx, g2_idx): antenna-axis indices of the two gains of every baseline, cached by the baseline array"""
        h = utils.arr_hash(bls)
        if h not in self.cache_aidx:
            if self.single_ant:
                i1 = i2 = [0] * len(bls)
            else:
                pairs = utils.blnum2ants(bls)
                where = {a: i for i, a in enumerate(self.ants)}
                i1, i2 = [where[b[0]] for b in pairs], [where[b[1]] for b in pairs]
            self.cache_aidx[h] = (torch.as_tensor(i1, dtype=torch.int32, device=self.device),
                                  torch.as_tensor(i2, dtype=torch.int32, device=self.device))
        return self.cache_aidx[h]

    def set_refant(self, refant):
        self.refant, self.refant_idx, self.rephase_mode = refant, None, None
        if refant is not None:
            assert refant in self.ants, "need a valid refant"
            self.refant_idx = self.ants.index(refant)
            channel = self.R.time_mode == 'channel' and self.R.freq_mode == 'channel'
            self.rephase_mode = 'rephase' if channel else 'zero'
            self.fix_refant_phs()

    def fix_refant_phs(self):
        with torch.no_grad():
            rephase_to_refant(self.params, self.R.param_type, self.refant_idx, p0=self.p0, mode=self.rephase_mode,
                              inplace=True)

    def forward(self, vd, undo=False, prior_cache=None, jones=None):
        if self.refant_idx is not None:
            self.fix_refant_phs()
        if getattr(self, '_vd', None) is None:
            self._vd = vd.copy(copydata=True, copymeta=True)
        vout = self._vd.copy(copydata=False, copymeta=False)
        params = self.params if self.p0 is None else self.params + self.p0
        if jones is None:
            jones = self.R(params)
        if getattr(self, '_hook_registry', None) is not None and jones.requires_grad:
            for r in self._hook_registry:
                jones.register_hook(r)
        self.eval_prior(prior_cache, inp_params=self.params, out_params=jones)
        jones = self.index_params(jones, times=vd.times)
        g1_idx, g2_idx = self.get_ant_idx(vd._blnums)
        vout.data, _ = _apply_cal(vd.data, jones, g1_idx, g2_idx, cal_2pol=self.polmode == '2pol',
                                  vis_type=self.vis_type, undo=undo)
        return vout

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.clear_cache()
            self.device = device
            if isinstance(self._times, torch.Tensor):
                self._times = utils.push(self._times, device)
        self.params = utils.push(self.params, device)
        self.R.push(device)
        if self.p0 is not None:
            self.p0 = utils.push(self.p0, device)
        for prs in (self.priors_inp_params, self.priors_out_params):
            for pr in (prs or []):
                if pr is not None:
                    pr.push(device)
