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
