"""
Minimal data containers with the reference's attribute layout: what RIME.forward emits
(`VisData`, dataset.py:289-411) and what sky models emit (`MapData`, dataset.py:1867-1936).
The tensor layout, the metadata fields and the index selection (`get_inds` / `get_data` / `get_cov`, what
imaging.VisMapper reads visibilities and weights through) are mirrored -- HDF5 IO, averaging etc. are outside
the hot path.
"""
import copy as _copy

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
        self.cov_ndim = None
        self.cov_logdet = torch.tensor(0.0)

    def set_cov(self, cov, cov_axis, icov=None):
        """covariance (data-shaped variances for cov_axis None, a matrix for 'full', matrices along the last two axes
        otherwise) with the log-determinant and dimension the likelihood normalisation uses (dataset.py:70-124)"""
        logdet = None
        if isinstance(cov, torch.Tensor):
            if cov_axis is None:
                logdet = torch.sum(torch.log(cov))
            elif cov_axis == 'full':
                logdet = torch.slogdet(cov).logabsdet
            else:
                logdet = torch.slogdet(cov.reshape((-1,) + tuple(cov.shape[-2:]))).logabsdet.sum()
            if torch.is_complex(logdet):
                logdet = logdet.real
        elif isinstance(icov, torch.Tensor) and cov_axis is None:
            logdet = torch.sum(-torch.log(icov))
        self.cov = cov.clone() if isinstance(cov, torch.Tensor) else cov
        self.icov = icov.clone() if isinstance(icov, torch.Tensor) else icov
        self.cov_axis = cov_axis
        self.cov_ndim = int(np.prod(self.data.shape)) if self.data is not None else None
        self.cov_logdet = logdet if logdet is not None else torch.tensor(0.0)

    def get_data(self, **kwargs):
        return self.data

    def get_flags(self, **kwargs):
        return self.flags

    def get_cov(self, **kwargs):
        return self.cov

    def get_icov(self, **kwargs):
        return self.icov

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

    def copy(self, copydata=False, copymeta=False, detach=True):
        """new VisData sharing (or cloning) the tensors; copymeta re-instantiates telescope / antpos / axes, which
        drops the telescope's conversion cache (dataset.py:556-605)"""
        vd = VisData()
        telescope, antpos = self.telescope, self.antpos
        times, freqs, blnums = self.times, self.freqs, self.blnums
        flags, cov, icov, data = self.flags, self.cov, self.icov, self.data
        if copydata and data is not None:
            data = (data.detach() if (data.requires_grad and detach) else data).clone()
        if copymeta:
            if telescope is not None:
                telescope = telescope.__class__(telescope.location, tloc=getattr(telescope, 'tloc', None),
                                                device=telescope.device)
            if antpos is not None:
                antpos = antpos.__class__(_copy.deepcopy(antpos.ants), antpos.antvecs.clone())
            times, freqs, blnums = _copy.deepcopy(times), _copy.deepcopy(freqs), _copy.deepcopy(blnums)
            flags, cov, icov = [x.clone() if isinstance(x, torch.Tensor) else x for x in (flags, cov, icov)]
        vd.setup_meta(telescope=telescope, antpos=antpos)
        vd.setup_data(blnums, times, freqs, pol=self.pol, data=data, flags=flags, cov=cov, cov_axis=self.cov_axis,
                      icov=icov, history=self.history)
        return vd

    # ---- selection (dataset.py:607-1042): one fancy-indexed axis at most, everything else slices
    def _bl2ind(self, bl):
        if isinstance(bl, (list, np.ndarray, torch.Tensor)):
            if isinstance(bl, torch.Tensor):
                bl = bl.cpu().numpy()
            elif isinstance(bl, list):
                bl = utils.ants2blnum(bl)
            return [self._bl2ind(b) for b in bl]
        if isinstance(bl, tuple):
            bl = utils.ants2blnum(bl)
        idx = np.where(self.blnums == bl)[0]
        if len(idx) == 0:
            raise ValueError("Couldn't find bl {}".format(bl))
        return idx[0]

    def _time2ind(self, time, atol=None):
        if isinstance(time, (list, np.ndarray)) or (isinstance(time, torch.Tensor) and time.ndim == 1):
            return np.concatenate([self._time2ind(t, atol) for t in time]).tolist()
        atol = atol if atol is not None else self.atol          # rtol 1e-13: 0.03 s on a Julian date
        return np.where(np.isclose(np.asarray(self.times.cpu()), float(time), atol=atol, rtol=1e-13))[0].tolist()

    def _freq2ind(self, freq, atol=None):
        if isinstance(freq, (list, np.ndarray)) or (isinstance(freq, torch.Tensor) and freq.ndim == 1):
            return np.concatenate([self._freq2ind(f, atol) for f in freq]).tolist()
        atol = atol if atol is not None else self.atol
        return torch.where(torch.isclose(self.freqs, torch.as_tensor(freq, dtype=self.freqs.dtype,
                                                                     device=self.freqs.device), atol=atol))[0].tolist()

    def _pol2ind(self, pol, data=None):
        if isinstance(pol, list):
            assert len(pol) == 1
            pol = pol[0]
        assert isinstance(pol, str)
        if self.pol is not None:
            if pol.lower() != self.pol.lower():
                raise ValueError("cannot index pol from 1pol {}".format(self.pol))
            return (slice(0, 1), slice(0, 1))
        if pol.lower() == 'ee':
            return (slice(0, 1), slice(0, 1))
        if pol.lower() == 'nn':
            data = data if data is not None else self.data
            return (slice(1, 2), slice(0, 1)) if tuple(data.shape[:2]) == (2, 1) else (slice(1, 2), slice(1, 2))
        raise ValueError("only 'ee' / 'nn' can be indexed")

    def get_inds(self, bl=None, times=None, freqs=None, pol=None, bl_inds=None, time_inds=None, freq_inds=None,
                 data=None, atol=None):
        """5 index objects (pol, pol, bl, time, freq) for a selection by value or by index (dataset.py:776-862)"""
        data = data if data is not None else self.data
        if bl is not None:
            assert bl_inds is None
            bl_inds = self._bl2ind(bl)
        elif bl_inds is None:
            bl_inds = slice(None)
        if times is not None:
            assert time_inds is None
            time_inds = self._time2ind(times, atol=atol)
        elif time_inds is None:
            time_inds = slice(None)
        if freqs is not None:
            assert freq_inds is None
            freq_inds = self._freq2ind(freqs, atol=atol)
        elif freq_inds is None:
            freq_inds = slice(None)
        pol_inds = self._pol2ind(pol, data=data) if pol is not None else (slice(None), slice(None))
        inds = tuple(utils._list2slice(i) for i in (pol_inds[0], pol_inds[1], bl_inds, time_inds, freq_inds))
        assert sum(isinstance(i, slice) for i in inds) > 3, "cannot fancy index more than 1 axis"
        return inds

    def _take(self, x, squeeze, try_view, **sel):
        if x is None:
            return None
        inds = self.get_inds(data=x, **sel)
        x = x[inds]
        if not try_view and all(isinstance(i, slice) for i in inds):
            x = x.clone()
        return x.squeeze() if squeeze else x

    def get_data(self, bl=None, times=None, freqs=None, pol=None, bl_inds=None, time_inds=None, freq_inds=None,
                 squeeze=True, data=None, try_view=False, **kwargs):
        """slice of the data tensor (dataset.py:864-911)"""
        return self._take(self.data if data is None else data, squeeze, try_view, bl=bl, times=times, freqs=freqs,
                          pol=pol, bl_inds=bl_inds, time_inds=time_inds, freq_inds=freq_inds, **kwargs)

    def get_flags(self, bl=None, times=None, freqs=None, pol=None, bl_inds=None, time_inds=None, freq_inds=None,
                  squeeze=True, flags=None, try_view=False, **kwargs):
        return self._take(self.flags if flags is None else flags, squeeze, try_view, bl=bl, times=times, freqs=freqs,
                          pol=pol, bl_inds=bl_inds, time_inds=time_inds, freq_inds=freq_inds, **kwargs)

    def get_cov(self, bl=None, times=None, freqs=None, pol=None, bl_inds=None, time_inds=None, freq_inds=None,
                squeeze=True, cov=None, try_view=False, atol=None, **kwargs):
        """slice of the covariance (dataset.py:954-1035): data-shaped (cov_axis None) or one covariance matrix
        along the named axis"""
        cov = self.cov if cov is None else cov
        if cov is None:
            return None
        if self.cov_axis is None:
            return self._take(cov, squeeze, try_view, bl=bl, times=times, freqs=freqs, pol=pol, bl_inds=bl_inds,
                              time_inds=time_inds, freq_inds=freq_inds)
        if self.cov_axis == 'full':
            raise NotImplementedError
        inds = self.get_inds(bl=bl, times=times, freqs=freqs, pol=pol, bl_inds=bl_inds, time_inds=time_inds,
                             freq_inds=freq_inds, data=self.data)
        axis = self.cov_axis
        if bl is not None or bl_inds is not None:
            cov = cov[..., inds[2]][..., inds[2]] if axis == 'bl' else cov[:, :, inds[2]]
        elif times is not None or time_inds is not None:
            cov = (cov[..., inds[3]][..., inds[3]] if axis == 'time' else
                   (cov[:, :, inds[3]] if axis == 'bl' else cov[:, :, :, inds[3]]))
        elif freqs is not None or freq_inds is not None:
            cov = cov[..., inds[4]][..., inds[4]] if axis == 'freq' else cov[:, :, :, inds[4]]
        elif pol is not None:
            cov = cov[inds[0], inds[0]]
        if squeeze:
            cov = cov.squeeze()
        if not try_view and all(isinstance(i, slice) for i in inds):
            cov = cov.clone()
        return cov

    def get_icov(self, bl=None, icov=None, try_view=False, **kwargs):
        return self.get_cov(bl=bl, cov=self.icov if icov is None else icov, try_view=try_view, **kwargs)

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


def pass_data(obj, copy=False, **kwargs):
    """read function of an in-memory Dataset (dataset.py:4127-4133)"""
    import copy as _c
    return _c.deepcopy(obj) if copy else obj


class Dataset(torch.utils.data.Dataset):
    """the minibatches of target (or input) data a LogProb iterates over, one data object per batch
    (dataset.py:3611-3648); objects held in memory unless a read function is given"""
    def __init__(self, data, read_fn=None, read_kwargs={}):
        if isinstance(data, (str, TensorData)):
            data = [data]
        self.data = data
        self.Ndata = len(data)
        self.read_fn = read_fn if read_fn is not None else pass_data
        self.read_kwargs = [read_kwargs for _ in data] if isinstance(read_kwargs, dict) else read_kwargs

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.read_fn(self.data[idx], **self.read_kwargs[idx])
