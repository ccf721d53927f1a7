"""
Minimal data containers with the reference's attribute layout: what RIME.forward emits
(`VisData`, dataset.py:289-411) and what sky models emit (`MapData`, dataset.py:1867-1936).
Only the tensor layout and metadata fields are mirrored -- HDF5 IO, selection, averaging etc.
are outside the hot path.
"""
import numpy as np
import torch

from . import utils


class TensorData:
    def __init__(self):
        self.data = None
        self.flags = None
        self.cov = None
        self.icov = None
        self.cov_axis = None

    def set_cov(self, cov, cov_axis, icov=None):
        self.cov, self.cov_axis, self.icov = cov, cov_axis, icov

    def push(self, device):
        for k in ('data', 'flags', 'cov', 'icov'):
            v = getattr(self, k, None)
            if isinstance(v, torch.Tensor):
                setattr(self, k, utils.push(v, device))


class VisData(TensorData):
    """visibilities of shape (Npol, Npol, Nbl, Ntimes, Nfreqs) + metadata (dataset.py:289)"""
    def __init__(self):
        super().__init__()
        self.atol = 1e-10
        self._file = None
        self.setup_meta()

    def setup_meta(self, telescope=None, antpos=None):
        self.telescope = telescope
        if antpos is not None and not isinstance(antpos, utils.AntposDict):
            antpos = utils.AntposDict(list(antpos.keys()), list(antpos.values()))
        self.antpos = antpos
        self.ants = antpos.ants if antpos is not None else None

    def setup_data(self, bls, times, freqs, pol=None, data=None, flags=None, cov=None,
                   cov_axis=None, icov=None, history='', file=None):
        self.data = data
        self._set_bls(bls)
        self.times = torch.as_tensor(times)
        self.Ntimes = len(times)
        self.freqs = torch.as_tensor(freqs)
        self.Nfreqs = len(freqs)
        self.pol = pol
        if isinstance(pol, str):
            assert pol.lower() in ['ee', 'nn'], "pol must be 'ee' or 'nn' for 1pol mode"
        self.Npol = 2 if pol is None else 1
        self.flags = flags
        self.set_cov(cov, cov_axis, icov=icov)
        self.history = history
        self._file = file

    def _set_bls(self, bls):
        if isinstance(bls, torch.Tensor):
            bls = bls.cpu().numpy()
        if isinstance(bls, np.ndarray) and bls.ndim == 1:
            self.blnums = bls
        else:
            self.blnums = np.asarray(utils.ants2blnum([tuple(b) for b in bls])) if len(bls) else np.array([])
        self._blnums = torch.as_tensor(self.blnums)
        self.Nbls = len(self.blnums)

    @property
    def bls(self):
        return utils.blnum2ants(self.blnums)

    def push(self, device, return_obj=False):
        super().push(device)
        self.freqs = utils.push(self.freqs, device)
        if return_obj:
            return self

    def _inflate_by_redundancy(self, new_bls, red_bl_inds, try_view=False):
        """
        new VisData whose baseline axis is self's indexed by red_bl_inds (one redundant-group index per
        baseline of new_bls): data, flags, cov and icov alike (dataset.py:1568-1602).  The gather is a
        differentiable index_select on the tensors' own device.
        """
        idx = torch.as_tensor(red_bl_inds, dtype=torch.int64)

        def take(x):
            if x is None:
                return None
            return x.index_select(2, idx.to(x.device))

        cov = self.cov if (self.cov is None or self.cov_axis not in (None, 'bl')) else take(self.cov)
        icov = self.icov if (self.icov is None or self.cov_axis not in (None, 'bl')) else take(self.icov)
        out = VisData()
        out.setup_meta(telescope=self.telescope, antpos=self.antpos)
        out.setup_data(new_bls, self.times, self.freqs, pol=self.pol, data=take(self.data), flags=take(self.flags),
                       cov=cov, cov_axis=self.cov_axis, icov=icov, history=self.history)
        return out

    def inflate_by_redundancy(self, bls, bl2red):
        """copy every redundant type held here over to the physical baselines `bls` of the same type
        (bl2red: baseline -> redundant-group index, e.g. ArrayModel.bl2red; dataset.py:1604-1640)"""
        mine = {bl2red[b]: i for i, b in enumerate(self.bls)}
        keep = [b for b in bls if bl2red[b] in mine]
        return self._inflate_by_redundancy(keep, [mine[bl2red[b]] for b in keep])


class RedVisInflate(utils.Module):
    """VisData redundant-inflation block of a forward-model chain (dataset.py:3699-3735): the step that
    follows RIME when only one baseline per redundant group was simulated"""
    def __init__(self, new_bls, red_bl_inds):
        super().__init__()
        self.new_bls = new_bls
        self.red_bl_inds = torch.as_tensor(red_bl_inds, dtype=torch.int64)
        self.device = None

    def __call__(self, vd, **kwargs):
        return vd._inflate_by_redundancy(new_bls=self.new_bls, red_bl_inds=self.red_bl_inds)

    def forward(self, vd, **kwargs):
        return self(vd, **kwargs)

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.red_bl_inds = utils.push(self.red_bl_inds, device)
            self.device = device


class MapData(TensorData):
    """sky map of shape (Npol, 1, Nfreqs, Npix) with pixel angles (dataset.py:1867)"""
    def __init__(self):
        super().__init__()
        self.atol = 1e-10
        self.setup_meta()

    def setup_meta(self, name=None):
        self.name = name

    def setup_data(self, freqs, df=None, pols=None, data=None, angs=None, flags=None, cov=None,
                   cov_axis=None, icov=None, norm=None, history=''):
        self.freqs = freqs
        self.df = df
        self.angs = angs
        self.pols = pols
        self.data = data
        self.flags = flags
        self.norm = norm
        self.set_cov(cov, cov_axis, icov=icov)
        self.history = history


def concat_VisData(vds, axis):
    """concatenate VisData along 'bl' or 'time' (dataset.py:3739), metadata from the first"""
    assert axis in ('bl', 'time')
    out = VisData()
    out.setup_meta(vds[0].telescope, vds[0].antpos)
    if axis == 'bl':
        data = torch.cat([v.data for v in vds], dim=2)
        bls = utils.flatten([v.bls for v in vds])
        times = vds[0].times
    else:
        data = torch.cat([v.data for v in vds], dim=3)
        bls = vds[0].bls
        times = torch.cat([torch.as_tensor(v.times) for v in vds])
    out.setup_data(bls, times, vds[0].freqs, pol=vds[0].pol, data=data, history=vds[0].history)
    return out
