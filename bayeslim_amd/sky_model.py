"""
Sky models with the reference's API (sky_model.py): `PointSky` (:154-280), `PointSkyResponse`
(:283-380), `PixelSky` (:383-500), `PixelSkyResponse` (:503-720), `CompositeModel` and
`Stokes2Coherency` (:1160-1300).  Each forward returns a `MapData` whose `.data` is
(Nstokes, 1, Nfreqs, Npix) and `.angs` is (2, Npix) (ra, dec) in degrees.

The only heavy piece, the a_lm -> pixel transform of `spatial_mode='alm'`, runs in HIP through
`sph_harm.AlmModel`; frequency responses are cheap elementwise torch ops kept in autograd.
"""
import numpy as np
import torch

from . import utils, dataset, ops
from .utils import _float, _cfloat


class DefaultResponse:
    def __init__(self):
        self.freq_mode = 'channel'
        self.freqs = None

    def __call__(self, params):
        return params

    def push(self, device):
        pass


class SkyBase(utils.Module):
    """common params / p0 / R handling of sky models (sky_model.py:13-150)"""
    def __init__(self, params, R=None, name=None, parameter=True, p0=None):
        super().__init__(name=name)
        self.params = params
        self.device = self.params.device
        self.p0 = p0
        if parameter:
            self.params = torch.nn.Parameter(self.params)
        self.R = DefaultResponse() if R is None else R
        self._args = dict(name=name)

    def _push(self, device, attrs=[]):
        if not isinstance(device, torch.dtype):
            self.device = device
        self.params = utils.push(self.params, device)
        for a in attrs:
            if hasattr(self, a):
                setattr(self, a, getattr(self, a).to(device))
        self.R.push(device)
        if self.p0 is not None:
            self.p0 = utils.push(self.p0, device)
        if isinstance(self.angs, torch.Tensor):
            self.angs = utils.push(self.angs, device)
        else:
            self.angs = tuple(utils.push(a, device) for a in self.angs)
        for prs in (self.priors_inp_params, self.priors_out_params):
            for pr in (prs or []):
                if pr is not None:
                    pr.push(device)

    def _forward_params(self, params):
        params = self.params if params is None else params
        return params if self.p0 is None else params + self.p0

    def _emit(self, sky, prior_cache):
        if getattr(self, '_hook_registry', None) is not None and sky.requires_grad:
            for r in self._hook_registry:
                sky.register_hook(r)
        self.eval_prior(prior_cache, inp_params=self.params, out_params=sky)
        out = dataset.MapData()
        out.setup_meta(name=getattr(self, 'name', None))
        angs = torch.vstack(list(self.angs)) if isinstance(self.angs, (tuple, list)) else torch.as_tensor(self.angs)
        freqs = self.R.freqs
        if getattr(self.R, '_freq_idx', None) is not None:
            freqs = freqs[self.R._freq_idx]
        return out, freqs, angs


class PointSky(SkyBase):
    """point sources at fixed (ra, dec) with parameterised flux density (sky_model.py:154-280)"""
    def __init__(self, params, angs, R=None, name=None, parameter=True, p0=None):
        super().__init__(params, R=R, name=name, parameter=parameter, p0=p0)
        self.angs = angs

    def forward(self, params=None, prior_cache=None, **kwargs):
        sky = self.R(self._forward_params(params))
        out, freqs, angs = self._emit(sky, prior_cache)
        out.setup_data(freqs=freqs, data=sky, angs=angs)
        return out

    def push(self, device, **kwargs):
        self._push(device, **kwargs)


class PointSkyResponse:
    """flux vs frequency: 'channel' | 'linear' | 'powerlaw' (sky_model.py:283-380)"""
    def __init__(self, freqs, freq_mode='linear', log=False, device=None, LM=None, freq_LM=None,
                 f0=None):
        self.log = log
        self.freqs = freqs
        self.freq_mode = freq_mode
        self.device = device
        self.LM = LM
        self.freq_LM = freq_LM
        self.f0 = f0
        self._args = dict(freq_mode=freq_mode)

    def __call__(self, params):
        if not utils.check_devices(params.device, self.device):
            params = params.to(self.device)
        if self.LM is not None:
            params = self.LM(params)
        if self.freq_mode == 'linear':
            params = self.freq_LM(params)
        elif self.freq_mode == 'powerlaw':
            amp = params[..., 0:1, :]
            if self.log:
                amp = torch.exp(amp)
            f = torch.as_tensor(self.freqs, device=params.device)
            params = amp * (f[:, None] / self.f0) ** params[..., 1:2, :]
        if self.log and self.freq_mode in ('channel', 'linear'):
            params = torch.exp(params)
        if getattr(self, '_freq_idx', None) is not None:
            params = params[..., self._freq_idx, :]
        return params

    def set_freq_index(self, idx=None):
        self._freq_idx = idx

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
        self.freqs = utils.push(torch.as_tensor(self.freqs), device)
        if isinstance(self.f0, torch.Tensor):
            self.f0 = utils.push(self.f0, device)
        for lm in (self.LM, self.freq_LM):
            if lm is not None and hasattr(lm, 'push'):
                lm.push(device)


class PixelSky(SkyBase):
    """pixelised sky brightness; output is flux density = R(params) * px_area (sky_model.py:383-500)"""
    def __init__(self, params, angs, px_area, R=None, name=None, parameter=True, p0=None):
        super().__init__(params, R=R, name=name, parameter=parameter, p0=p0)
        self.angs = angs
        self.px_area = torch.as_tensor(px_area)

    def forward(self, params=None, prior_cache=None, **kwargs):
        sky = self.R(self._forward_params(params))
        out, freqs, angs = self._emit(sky, prior_cache)
        if self.px_area.device != sky.device:
            self.px_area = self.px_area.to(sky.device)      # once: a per-call H2D copy is a host sync
        out.setup_data(freqs=freqs, data=sky * self.px_area, angs=angs)
        return out

    def push(self, device, **kwargs):
        self._push(device, **kwargs)
        self.px_area = utils.push(self.px_area, device)


class PixelSkyResponse:
    """
    params (Nstokes, 1, Nfreq_coeff, Npix_coeff) -> sky (Nstokes, 1, Nfreqs, Npix)
    (sky_model.py:503-720).  spatial_mode 'pixel' | 'linear' | 'alm' (spat_LM = LinearModel /
    AlmModel); freq_mode 'channel' | 'linear' | 'powerlaw'.  ('bessel' needs the cosmology
    module -- out of scope.)
    """
    def __init__(self, freqs, comp_params=False, spatial_mode='pixel', freq_mode='channel',
                 device=None, transform_order=0, cosmo=None, spat_LM=None, freq_LM=None, f0=None,
                 gln=None, kbins=None, log=False, real_output=True, abs_output=False, LM=None,
                 sky0=None):
        if freq_mode == 'bessel':
            raise NotImplementedError("freq_mode='bessel' is outside the RIME hot path")
        self.freqs = freqs
        self.comp_params = comp_params
        self.Nfreqs = len(freqs)
        self.spatial_mode = spatial_mode
        self.freq_mode = freq_mode
        self.device = device
        self.transform_order = transform_order
        self.cosmo = cosmo
        self.log = log
        self.LM = LM
        self.real_output = real_output
        self.abs_output = abs_output
        self.sky0 = sky0
        self.freq_LM = freq_LM
        self.spat_LM = spat_LM
        self.f0 = f0
        self._args = dict(freq_mode=freq_mode, spatial_mode=spatial_mode)

    def spatial_transform(self, params):
        if self.comp_params and not torch.is_complex(params):
            params = utils.viewcomp(params)
        if self.spatial_mode == 'pixel':
            return params
        return self.spat_LM(params)

    def freq_transform(self, params):
        if self.comp_params and not torch.is_complex(params):
            params = utils.viewcomp(params)
        if self.freq_mode == 'channel':
            return params
        if self.freq_mode == 'linear':
            return self.freq_LM(params)
        if self.freq_mode == 'powerlaw':
            f = torch.as_tensor(self.freqs, device=params.device)
            return params[..., 0:1, :] * (f[:, None] / self.f0) ** params[..., 1:2, :]
        raise ValueError(self.freq_mode)

    def __call__(self, params):
        if not utils.check_devices(params.device, self.device):
            params = params.to(self.device)
        if self.LM is not None:
            params = self.LM(params)
        if self.transform_order == 0:
            params = self.freq_transform(self.spatial_transform(params))
        else:
            params = self.spatial_transform(self.freq_transform(params))
        if self.real_output and torch.is_complex(params):
            params = params.real
        if self.log:
            params = torch.exp(params)
        if getattr(self, '_freq_idx', None) is not None:
            params = params[..., self._freq_idx, :]
        if self.sky0 is not None:
            params = params + self.sky0
        if self.abs_output:
            params = params.abs()
        return params

    def set_freq_index(self, idx=None):
        self._freq_idx = idx

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
        self.freqs = utils.push(torch.as_tensor(self.freqs), device)
        for lm in (self.spat_LM, self.freq_LM, self.LM):
            if lm is not None and hasattr(lm, 'push'):
                lm.push(device)
        if isinstance(self.sky0, torch.Tensor):
            self.sky0 = utils.push(self.sky0, device)
        if isinstance(self.f0, torch.Tensor):
            self.f0 = utils.push(self.f0, device)


class CompositeModel(utils.Module):
    """several sky models evaluated together; forward returns a list of MapData
    (sky_model.py:CompositeModel.forward)"""
    def __init__(self, models, name=None):
        super().__init__(name=name)
        self.models = list(models)
        for k, m in models.items():
            self.add_module(k, m)
        self.device = getattr(next(iter(models.values())), 'device', None)

    def forward(self, *args, prior_cache=None, **kwargs):
        return [getattr(self, k)(prior_cache=prior_cache) for k in self.models]

    def push(self, device):
        for k in self.models:
            getattr(self, k).push(device)
        if not isinstance(device, torch.dtype):
            self.device = device


class Stokes2Coherency(utils.Module):
    """
    Stokes (I, fQ, fU, fV) -> coherency [[I+Q, U-iV], [U+iV, I-Q]], Q = I fQ etc.
    Input (1, 1, ...) Stokes I with fractional-pol `params` (<=3, 1, ...), or (4, 1, ...) /
    (2, 2, ...) full Stokes (sky_model.py:1160-1300).
    """
    def __init__(self, params=None, parameter=False):
        super().__init__()
        self.params = params
        if parameter and isinstance(params, torch.Tensor):
            self.params = torch.nn.Parameter(params)

    def forward(self, sky_comp, prior_cache=None):
        if isinstance(sky_comp, dataset.MapData):
            sky_comp.data = self.forward(sky_comp.data, prior_cache=prior_cache)
            return sky_comp
        S = sky_comp
        if len(S) == 1:
            I = S[0, 0]
            if self.params is None:
                z = torch.zeros_like(I)
                return torch.stack([torch.stack([I, z]), torch.stack([z, I])])
            fr = self.params if isinstance(self.params, torch.Tensor) else self.params().data
            fr = fr.to(I.device)
            if len(fr) == 3 and I.is_cuda:
                # one fused pass each way (ops.stokes2coherency, csrc/jones.hip) instead of ~15 elementwise / stack kernels
                coh = ops.stokes2coherency(I, fr)
                if coh is not None:
                    return coh
            Q = I * fr[0, 0]
            U = I * fr[1, 0] if len(fr) > 1 else torch.zeros_like(I)
            V = I * fr[2, 0] if len(fr) > 2 else None
        else:
            flat = S.reshape((4,) + tuple(S.shape[2:])) if tuple(S.shape[:2]) == (2, 2) else S[:, 0]
            I = flat[0]
            Q, U, V = I * flat[1], I * flat[2], I * flat[3]
        if V is None:
            return torch.stack([torch.stack([I + Q, U]), torch.stack([U, I - Q])])
        return torch.stack([torch.stack([I + Q, U - 1j * V]), torch.stack([U + 1j * V, I - Q])])

    def push(self, device):
        if isinstance(self.params, torch.Tensor):
            self.params = utils.push(self.params, device)
        elif self.params is not None and hasattr(self.params, 'push'):
            self.params.push(device)
