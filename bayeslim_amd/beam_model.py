"""
Primary-beam models with the reference's API (beam_model.py): `PixelBeam` (gen_beam, apply_beam,
:17-567), response functions `PixelResponse` (:570-846), `GaussResponse` (:848), `AiryResponse`
(:902), `UniformResponse` (:991), `YlmResponse` (:1019-1405), and `cut_sky_fov` (:1681).

The interpolation gather / scatter-add and the a_lm -> pixel product go through HIP kernels
(ops.interp_gather, ops.alm2pix).  The small (Nf x P) elementwise pieces (analytic beams, abs /
exp, J B J^dagger per beam-model pair) are torch ops on the GPU, differentiated by autograd, as
SURVEY.md section 2 scopes them.
"""
import math

import numpy as np
import torch

import os

from . import utils, sph_harm, ops
from .utils import _float, _cfloat, D2R

FUSED_JONES = os.environ.get('RIME_FUSED_JONES', '1') != '0'       # 4-pol beam x sky product in one pass (csrc/jones.hip)


class PixelBeam(utils.Module):
    """
    Antenna primary beam on a pixelised / point-source sky: psky = A_p B A_q^dagger.
    Modes by params shape (Npol, Nvec, Nmodel, Nfreqs, Npix) (beam_model.py:46-55):
      1pol: powerbeam (1,1,1,..) or antenna beam (1,Nvec,..); 2pol: powerbeam (2,1,1,..);
      4pol: (2,2,..) Jones.
    """
    def __init__(self, params, freqs, R=None, ant2beam=None, parameter=True, pol=None,
                 powerbeam=True, fov=180, name=None, p0=None, offset=None, skycut_cache=False,
                 skycut_device=None):
        super().__init__(name=name)
        self.params = params
        self.p0 = p0
        self.device = self.params.device
        if parameter:
            self.params = torch.nn.Parameter(self.params)
        self.R = UniformResponse() if R is None else R
        self.powerbeam = powerbeam
        if hasattr(self.R, 'powerbeam'):
            assert self.powerbeam == self.R.powerbeam
        self.Npol, self.Nvec, self.Nmodel = params.shape[:3]
        if self.powerbeam:
            assert self.Nmodel == self.Nvec == 1
        self.freqs = freqs
        self.Nfreqs = len(freqs)
        self.fov = fov
        self.pol = pol
        if ant2beam is None:
            assert params.shape[2] == 1, "only 1 model for default ant2beam"
            self.ant2beam = utils.SimpleIndex()
        else:
            # NB: the reference leaves self.ant2beam unset when a dict is passed
            # (beam_model.py:153-155) and callers assign it afterwards; set it here
            self.ant2beam = ant2beam
        offset = (0, 0) if offset is None else offset
        self.set_pointing_offset(*offset)
        self.skycut_cache = skycut_cache
        self.skycut_device = skycut_device
        self.clear_cache()
        self._args = dict(powerbeam=powerbeam, fov=fov, Npol=self.Npol, Nmodel=self.Nmodel)

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
        self.params = utils.push(self.params, device)
        self.R.push(device)
        self.freqs = utils.push(torch.as_tensor(self.freqs), device)
        if self.p0 is not None:
            self.p0 = utils.push(self.p0, device)
        for prs in (self.priors_inp_params, self.priors_out_params):
            for pr in (prs or []):
                if pr is not None:
                    pr.push(device)
        self.clear_cache()

    # -- FoV cut -------------------------------------------------------------------------
    def fov_cut(self, zen):
        """indices with zen < fov/2 (strict), or slice(None) for fov >= 360 (beam_model.py:221-224)"""
        if self.fov < 360:
            return torch.where(zen < self.fov / 2)[0]
        return slice(None)

    def gen_beam(self, zen, az, prior_cache=None, out_stride=None):
        """
        beam (Npol, Nvec, Nmodel, Nf, P), cut, zen[cut], az[cut] (beam_model.py:197-271).
        `out_stride` (build extension): pad the pixel axis of `beam` with zeros for the fused
        fringe kernel; only honoured by interpolating responses.
        """
        zen_hash = getattr(zen, '_arr_hash', None)
        cut = self.query_cache(zen) if self.skycut_cache else None
        if cut is None:
            cut = self.fov_cut(zen)
            if self.skycut_cache:
                self.set_skycut_cache(zen, cut, device=self.skycut_device)
        zen, az = zen[cut], az[cut]
        if zen_hash:
            zen._arr_hash = zen_hash
        p = self.params if self.p0 is None else self.params + self.p0
        tx, ty = getattr(self, 'theta_x', 0), getattr(self, 'theta_y', 0)
        if tx > 0 or ty > 0:
            nz, na = pointing_offset(utils.tensor2numpy(zen) * D2R, utils.tensor2numpy(az) * D2R, tx, ty)
            new_zen = torch.as_tensor(nz, device=zen.device) / D2R
            new_az = torch.as_tensor(na, device=zen.device) / D2R
        else:
            new_zen, new_az = zen, az
        if out_stride is not None and getattr(self.R, 'supports_out_stride', False):
            beam = self.R(p, new_zen, new_az, self.freqs, out_stride=out_stride)
        else:
            beam = self.R(p, new_zen, new_az, self.freqs)
        if getattr(self, '_hook_registry', None) is not None:
            bc = getattr(self.R, 'beam_cache', None)
            if bc is not None and bc.requires_grad:
                for r in self._hook_registry:
                    bc.register_hook(r)
        self.eval_prior(prior_cache)
        return beam, cut, zen, az

    def eval_response(self, zen, az, prior_cache=None):
        """
        The part of gen_beam() after the FoV cut (beam_model.py:238-269): params (+p0) ->
        pointing offset -> R(p, zen, az, freqs) -> gradient hooks -> prior.  RIME calls it once
        per forward on the concatenated, already-cut angles of all time steps of the minibatch,
        so one interpolation launch serves every time step.
        """
        p = self.params if self.p0 is None else self.params + self.p0
        tx, ty = getattr(self, 'theta_x', 0), getattr(self, 'theta_y', 0)
        if tx > 0 or ty > 0:
            h = getattr(zen, '_arr_hash', None)
            nz, na = pointing_offset(utils.tensor2numpy(zen) * D2R, utils.tensor2numpy(az) * D2R, tx, ty)
            zen = torch.as_tensor(nz, device=zen.device) / D2R
            az = torch.as_tensor(na, device=zen.device) / D2R
            if h is not None:
                zen._arr_hash = h
        beam = self.R(p, zen, az, self.freqs)
        if getattr(self, '_hook_registry', None) is not None:
            bc = getattr(self.R, 'beam_cache', None)
            if bc is not None and bc.requires_grad:
                for r in self._hook_registry:
                    bc.register_hook(r)
        self.eval_prior(prior_cache)
        return beam

    def response_map_and_stencil(self, zen, az, prior_cache=None):
        """
        For RIME's fused psky builder: the forwarded beam map (PixelResponse.beam_cache) and the
        interpolation stencil at (zen, az) instead of the interpolated beam -- same side effects as
        eval_response (beam cache, gradient hooks, prior).  None when the fused form does not apply:
        anything but a 1-pol, single-model, real power beam on a plain PixelResponse without
        pointing offsets.
        """
        R = self.R
        if type(R).__call__ is not PixelResponse.__call__ or getattr(R, 'Rchi', None) is not None:
            return None
        if not (self.powerbeam and self.Npol == 1 and getattr(self, 'Nvec', 1) == 1):
            return None
        if getattr(self, 'theta_x', 0) > 0 or getattr(self, 'theta_y', 0) > 0:
            return None
        p = self.params if self.p0 is None else self.params + self.p0
        if p.dim() != 5 or p.shape[2] != 1:
            return None
        if R.beam_cache is None:
            R.set_beam_cache(p)
        bc = R.beam_cache
        if bc.is_complex() or bc.dim() != 5 or tuple(bc.shape[:3]) != (1, 1, 1):
            return None
        if getattr(self, '_hook_registry', None) is not None and bc.requires_grad:
            for r in self._hook_registry:
                bc.register_hook(r)
        self.eval_prior(prior_cache)
        return bc, R.get_stencil(zen, az)

    # -- beam x sky ----------------------------------------------------------------------
    def modelpairs(self, bls):
        """sorted unique (model1, model2) pairs and the pair index per baseline (beam_model.py:303-305)"""
        bls = utils.blnum2ants(bls)
        if isinstance(bls, tuple):
            bls = [bls]
        a2b = self.ant2beam
        pairs = [(a2b[b[0]], a2b[b[1]]) for b in bls]
        uniq = sorted(set(pairs))
        lut = {p: i for i, p in enumerate(uniq)}
        return uniq, [lut[p] for p in pairs]

    def apply_beam_mp(self, beam, sky, modelpairs):
        """
        psky per beam-model pair, (Npol, Npol|1, Nmp, Nf, P): the arithmetic of
        beam_model.py:313-363 without the final Nmp -> Nbl expansion, which the fringe kernel
        performs by indexing (it never materialises (.., Nbl, Nf, P)).
        """
        if not utils.check_devices(beam.device, self.device):
            beam = beam.to(self.device)
        if not utils.check_devices(sky.device, self.device):
            sky = sky.to(self.device)
        select = len(modelpairs) > 1 or beam.shape[2] > 1
        if select:
            # index tensors cached per (pairs, device): building them per call is an H2D copy + sync
            ckey = (tuple(modelpairs), str(beam.device))
            cache = self.__dict__.setdefault('_mp_index_cache', {})
            if ckey not in cache:
                cache[ckey] = (torch.as_tensor([mp[0] for mp in modelpairs], device=beam.device),
                               torch.as_tensor([mp[1] for mp in modelpairs], device=beam.device))
            i1, i2 = cache[ckey]
        beam1 = beam.index_select(2, i1) if select else beam
        if not self.powerbeam:
            beam2 = beam.index_select(2, i2) if select else beam
        if sky.ndim == 4:
            sky = sky[:, :, None]
        if self.Npol == 1 and self.Nvec == 1:
            assert tuple(sky.shape[:2]) == (1, 1)
            return beam1 * sky if self.powerbeam else (beam1 * beam2.conj()) * sky
        if self.powerbeam:
            assert self.Npol == 2 and self.Nvec == 1
            assert tuple(sky.shape[:2]) == (1, 1)
            return beam1[:, :1] * sky[0:1, 0:1]                       # (2, 1, Nmp, Nf, P)
        assert tuple(sky.shape[:2]) == (2, 2)
        if (sky.is_complex() and sky.is_cuda and FUSED_JONES and sky.shape[2] in (1, beam1.shape[2])
                and tuple(sky.shape[-2:]) == tuple(beam1.shape[-2:])          # a sky that broadcasts over Nf / P: torch path
                and beam1.dtype in (sky.dtype, sky.real.dtype) and beam2.dtype == beam1.dtype):
            # J_p S J_q^dagger in one pass (csrc/jones.hip) instead of two broadcast products with 8x temporaries
            return ops.jones_apply(beam1, beam1 if (beam2 is beam1) else beam2, sky)
        dt = torch.promote_types(beam1.dtype, sky.dtype)
        b1, b2c, sk = beam1.to(dt), beam2.conj().to(dt), sky.to(dt)
        # out[a,d] = sum_{b,c} b1[a,b] sky[b,c] conj(b2[d,c])   (J_p B J_q^dagger)
        t = (b1[:, :, None] * sk[None]).sum(1)                         # (a, c, ...)
        return (t[:, None] * b2c[None]).sum(2)                         # (a, d, ...)

    def apply_beam(self, beam, bls, sky):
        """psky (Npol, Npol|1, Nbl, Nf, P) with the reference's layout (beam_model.py:273-372)"""
        pairs, idx = self.modelpairs(bls)
        psky = self.apply_beam_mp(beam, sky, pairs)
        if len(pairs) > 1:
            return psky.index_select(2, torch.as_tensor(idx, device=psky.device))
        return psky.expand(psky.shape[:2] + (len(idx),) + psky.shape[3:])

    def forward(self, sky_comp, telescope, time, bls, prior_cache=None, **kwargs):
        """perceived sky of one sky component at one time (beam_model.py:374-421)"""
        zen, az = telescope.eq2top(time, sky_comp.angs[0], sky_comp.angs[1], store=False)
        beam, cut, zen, az = self.gen_beam(zen, az, prior_cache=prior_cache)
        sky = cut_sky_fov(sky_comp.data, cut)
        return dict(sky=self.apply_beam(beam, bls, sky), angs=cut_sky_fov(sky_comp.angs, cut),
                    zenaz=torch.vstack([zen, az]))

    def eval_prior(self, prior_cache, inp_params=None, out_params=None):
        """priors on params and on the forwarded beam map (beam_model.py:423-469)"""
        if prior_cache is None or self.name in prior_cache:
            return
        val = torch.as_tensor(0.0)
        if self.priors_inp_params is not None:
            inp = self.params if inp_params is None else inp_params
            for pr in self.priors_inp_params:
                if pr is not None:
                    val = val + pr(inp)
        if self.priors_out_params is not None:
            if out_params is None and hasattr(self.R, 'beam_cache'):
                if self.R.beam_cache is None:
                    self.R.set_beam_cache(self.params if self.p0 is None else self.params + self.p0)
                out_params = self.R.beam_cache
            for pr in self.priors_out_params:
                if pr is not None:
                    val = val + pr(out_params)
        prior_cache[self.name] = val

    def clear_graph_tensors(self):
        if hasattr(self.R, 'clear_beam_cache'):
            self.R.clear_beam_cache()

    def set_pointing_offset(self, theta_x=0, theta_y=0):
        self.theta_x, self.theta_y = theta_x, theta_y

    def set_skycut_cache(self, zen, cut, device=None):
        h = utils.arr_hash(zen)
        if h not in self.cache:
            if isinstance(cut, torch.Tensor) and device is not None and not utils.check_devices(cut.device, device):
                cut = cut.to(device)
            self.cache[h] = cut

    def query_cache(self, zen):
        return self.cache.get(utils.arr_hash(zen))

    def clear_cache(self):
        self.cache = {}


class PixelResponse(utils.PixInterp):
    """
    Pixelised beam map (params) -> beam at (zen, az) by interpolation (beam_model.py:570-846).
    forward(): LM -> complex view -> freq_LM -> .real -> exp | abs -> + beam0 -> / |norm_pix|.
    The forwarded map is kept as `beam_cache` for all time steps of one RIME forward.
    """
    supports_out_stride = True

    def __init__(self, freqs, pixtype, beam0=None, comp_params=False, interp_mode='nearest',
                 theta=None, phi=None, theta_grid=None, phi_grid=None, freq_mode='channel',
                 freq_LM=None, nside=None, device=None, log=False, powerbeam=True, realbeam=True,
                 Rchi=None, interp_cache_depth=None, taper_kwargs=None, LM=None, norm_pix=None):
        super().__init__(pixtype, interp_mode=interp_mode, nside=nside, device=device,
                         theta_grid=theta_grid, phi_grid=phi_grid,
                         interp_cache_depth=interp_cache_depth)
        assert isinstance(comp_params, bool)
        self.beam0 = beam0
        self.theta, self.phi = theta, phi
        self.powerbeam = powerbeam
        self.realbeam = True if powerbeam else realbeam
        self.freqs = freqs
        self.comp_params = comp_params
        self.log = log
        self.freq_mode = freq_mode
        self.freq_ax = 3
        self.Rchi = Rchi
        self.clear_beam_cache()
        self.taper_kwargs = taper_kwargs
        self.LM = LM
        self.norm_pix = norm_pix
        self.freq_LM = freq_LM
        self._args = dict(interp_mode=interp_mode, freq_mode=freq_mode)

    def _setup(self, **kwargs):
        pass

    def push(self, device):
        super().push(device)
        self.freqs = utils.push(torch.as_tensor(self.freqs), device)
        for k in ('theta', 'phi', 'beam0'):
            v = getattr(self, k, None)
            if isinstance(v, torch.Tensor):
                setattr(self, k, utils.push(v, device))
        for lm in (self.LM, self.freq_LM):
            if lm is not None and hasattr(lm, 'push'):
                lm.push(device)
        self.clear_beam_cache()

    def _post(self, p):
        if self.realbeam and torch.is_complex(p):
            p = p.real
        if self.log:
            p = torch.exp(p)
        elif self.powerbeam:
            p = torch.abs(p)
        if self.beam0 is not None:
            p = p + self.beam0
        if getattr(self, 'taper_kwargs', None) is not None:
            p = p * beam_edge_taper(self.theta, device=p.device, **self.taper_kwargs)
        if self.norm_pix is not None:
            p = p / p[..., self.norm_pix:self.norm_pix + 1].detach().abs()
        return p

    def forward(self, params):
        if not utils.check_devices(params.device, self.device):
            params = params.to(self.device)
        if self.LM is not None:
            params = self.LM(params)
        if self.comp_params and not torch.is_complex(params):
            params = utils.viewcomp(params)
        p = params if self.freq_mode == 'channel' else self.freq_LM(params)
        return self._post(p)

    def __call__(self, params, zen, az, *args, out_stride=None):
        if self.beam_cache is None:
            self.set_beam_cache(params)
        b = self.interp(self.beam_cache, zen, az, out_stride=out_stride)
        return self.apply_Rchi(b)

    def clear_beam_cache(self):
        self.beam_cache = None

    def __getstate__(self):
        # pickle / deepcopy: the forwarded map is a graph tensor of the LAST forward (the reference clears it through
        # clear_graph_tensors before the next one, utils.py:1306-1320); a copy starts without it -- torch refuses to
        # deep-copy a non-leaf tensor, and RIME.forward rebuilds the map at its start anyway
        sup = getattr(super(), '__getstate__', None)
        state = dict(sup()) if sup is not None else dict(self.__dict__)
        state['beam_cache'] = None
        return state

    def set_beam_cache(self, params):
        self.beam_cache = self.forward(params)
        return self.beam_cache

    def apply_Rchi(self, beam):
        if self.Rchi is None:
            return beam
        raise NotImplementedError        # as the reference (beam_model.py:842)


class GaussResponse:
    """exp(-0.5((l/sig_ew)^2 + (m/sig_ns)^2)); params (Npol,Nvec,Nmodel,Nf,2) (beam_model.py:848-899)"""
    def __init__(self, powerbeam=True):
        self.freq_mode = 'channel'
        self.freq_ax = 3
        self.powerbeam = powerbeam

    def _setup(self):
        pass

    def __call__(self, params, zen, az, freqs):
        zr = torch.as_tensor(zen, device=params.device) * D2R
        ar = torch.as_tensor(az, device=params.device) * D2R
        srad = torch.where(zr > math.pi / 2, torch.ones_like(zr), torch.sin(zr))
        l, m = srad * torch.sin(ar), srad * torch.cos(ar)
        beam = torch.exp(-0.5 * ((l / params[..., 0:1]) ** 2 + (m / params[..., 1:2]) ** 2))
        return beam if self.powerbeam else torch.sqrt(beam)

    def push(self, device):
        pass


def airy_disk(zen, az, Dew, freqs, Dns=None, freq_ratio=1.0, square=True, **kwargs):
    """[2 J1(x)/x]^(2|1), x = pi D nu sin(zen)/c; zen, az in radians (beam_model.py:1418-1482)"""
    zen = torch.as_tensor(zen)
    az = torch.as_tensor(az, device=zen.device)
    freqs = torch.as_tensor(freqs, device=zen.device)
    z = torch.clamp(zen, max=math.pi / 2)
    diameter = Dew if Dns is None else Dns + torch.abs(torch.sin(az)) ** 2 * (Dew - Dns)
    x = (diameter * torch.sin(z) * math.pi * freqs.reshape(-1, 1) * freq_ratio / 2.99792458e8).clip(1e-10)
    b = 2.0 * torch.special.bessel_j1(x) / x
    return b ** 2 if square else b


class AiryResponse:
    """Airy-disk beam; params (Npol,Nvec,Nmodel,1,1|2) aperture diameter(s) [m] (beam_model.py:902-988)"""
    def __init__(self, freq_ratio=1.0, powerbeam=True, brute_force=False, Ntau=100, taper_kwargs=None):
        self.freq_ratio = freq_ratio
        self.freq_mode = 'other'
        self.freq_ax = None
        self.powerbeam = powerbeam
        self.taper_kwargs = taper_kwargs

    def _setup(self):
        pass

    def __call__(self, params, zen, az, freqs):
        Dew = params[..., 0:1]
        Dns = params[..., 1:2] if params.shape[-1] > 1 else None
        zen = torch.as_tensor(zen, device=params.device)
        az = torch.as_tensor(az, device=params.device)
        beam = airy_disk(zen * D2R, az * D2R, Dew, torch.as_tensor(freqs, device=params.device), Dns,
                         self.freq_ratio, square=self.powerbeam)
        if self.taper_kwargs is not None:
            beam = beam * beam_edge_taper(zen, device=beam.device, **self.taper_kwargs)
        return beam

    def push(self, device):
        pass


class UniformResponse:
    """all-ones beam (beam_model.py:991-1016)"""
    def __init__(self, freqs=None, device=None, taper_kwargs=None):
        self.freqs = freqs
        self.taper_kwargs = taper_kwargs
        self.device = device

    def _setup(self):
        pass

    def __call__(self, params, zen, az, freqs):
        out = torch.ones(params.shape[:3] + (len(freqs), len(zen)), dtype=_float(),
                         device=self.device if self.device is not None else params.device)
        if self.taper_kwargs is not None:
            out = out * beam_edge_taper(zen, device=out.device, **self.taper_kwargs)
        return out

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device


class YlmResponse(PixelResponse, sph_harm.AlmModel):
    """
    a_lm beam: params (Npol, Nvec, Nmodel, Ndeg, Ncoeff) -> pixel beam via AlmModel.forward_alm,
    then interpolated like PixelResponse ('interpolate' mode) or evaluated exactly at the
    requested angles ('generate') (beam_model.py:1019-1405).
    """
    def __init__(self, l, m, freqs, pixtype='healpix', beam0=None, comp_params=False,
                 mode='interpolate', device=None, interp_mode='nearest', theta=None, phi=None,
                 theta_grid=None, phi_grid=None, nside=None, powerbeam=True, realbeam=True,
                 log=False, freq_mode='channel', freq_LM=None, Ylm_kwargs=None, Rchi=None,
                 separable=False, interp_cache_depth=None, taper_kwargs=None, LM=None,
                 norm_pix=None):
        realbeam = True if powerbeam else realbeam
        PixelResponse.__init__(self, freqs, pixtype, nside=nside, beam0=beam0,
                               interp_mode=interp_mode, theta=theta, phi=phi, freq_mode=freq_mode,
                               comp_params=comp_params, freq_LM=freq_LM, Rchi=Rchi,
                               theta_grid=theta_grid, phi_grid=phi_grid, norm_pix=norm_pix,
                               interp_cache_depth=interp_cache_depth, powerbeam=powerbeam,
                               realbeam=realbeam, device=device)
        sph_harm.AlmModel.__init__(self, l, m, default_kw=Ylm_kwargs, real_output=realbeam, LM=LM)
        self.mode = mode
        self.beam_cache = None
        self.separable = separable
        self.device = device
        self.log = log
        self.taper_kwargs = taper_kwargs
        self._args = dict(mode=mode, interp_mode=interp_mode, freq_mode=freq_mode)

    def forward(self, params, zen, az, *args):
        if not utils.check_devices(params.device, self.device):
            params = params.to(self.device)
        if self.LM is not None:
            params = self.LM(params)
        if self.comp_params and not torch.is_complex(params):
            params = utils.viewcomp(params)
        p = params if self.freq_mode == 'channel' else self.freq_LM(params)
        Ylm, alm_mult = self.get_Ylm(zen, az, h=utils.arr_hash(zen), separable=self.separable)
        beam = self.forward_alm(p, Ylm=Ylm, alm_mult=alm_mult, ignoreLM=True)
        if self.log:
            beam = torch.exp(beam)
        elif self.powerbeam:
            beam = torch.abs(beam)
        if self.beam0 is not None:
            beam = beam + self.beam0
        if self.taper_kwargs is not None:
            beam = beam * beam_edge_taper(zen, device=beam.device, **self.taper_kwargs)
        if self.norm_pix is not None:
            beam = beam / beam[..., self.norm_pix:self.norm_pix + 1].detach().abs()
        return beam

    def __call__(self, params, zen, az, *args, out_stride=None):
        if self.mode == 'generate':
            return self.forward(params, zen, az)
        if self.beam_cache is None:
            self.set_beam_cache(params)
        return self.interp(self.beam_cache, zen, az, out_stride=out_stride)

    @property
    def supports_out_stride(self):
        return self.mode != 'generate'

    def set_beam_cache(self, params):
        if self.separable:
            self.beam_cache = self.forward(params, self.theta_grid, self.phi_grid)
        else:
            self.beam_cache = self.forward(params, self.theta, self.phi)
        return self.beam_cache

    def push(self, device):
        PixelResponse.push(self, device)
        sph_harm.AlmModel.push(self, device)


def cut_sky_fov(sky, cut):
    """sky[..., cut] (index_select for integer cuts) (beam_model.py:1681-1698)"""
    if isinstance(cut, slice):
        return sky[..., cut]
    cut = torch.as_tensor(cut)
    if not utils.check_devices(cut.device, sky.device):
        cut = cut.to(sky.device)
    return sky.index_select(-1, cut)


def beam_edge_taper(zen, mode='gauss', fov=180, device=None, mu=85, sigma=2.5, alpha=0.1):
    """Gaussian roll-off of the beam beyond zen = mu [deg] (beam_model.py:1701-1735, 'gauss' mode)"""
    zen = torch.as_tensor(zen, device=device)
    taper = torch.ones(len(zen), device=device, dtype=_float())
    if mode != 'gauss':
        raise NotImplementedError("only mode='gauss' is provided")
    s = zen >= mu
    taper[s] = torch.exp(-0.5 * (zen[s] - mu) ** 2 / sigma ** 2).to(taper.dtype)
    return taper


def pointing_offset(zen, az, theta_x, theta_y):
    """small-angle rotation of (zen, az) [rad] about x-hat then y-hat (non-differentiable)"""
    s = np.array([np.sin(zen) * np.sin(az), np.sin(zen) * np.cos(az), np.cos(zen)])
    cx, sx, cy, sy = np.cos(theta_x), np.sin(theta_x), np.cos(theta_y), np.sin(theta_y)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    x, y, z = Ry @ Rx @ s
    return np.arccos(np.clip(z, -1, 1)), np.mod(np.arctan2(x, y), 2 * np.pi)
