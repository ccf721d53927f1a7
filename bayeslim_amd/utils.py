"""
Host-side utilities mirroring the pieces of the reference's `utils.py` that the RIME path and
its callers rely on (same names, argument meaning and error behaviour; reference lines cited
per item, paths relative to /root/reference/bayeslim/).  Setup-time index arithmetic lives here;
per-forward arithmetic goes through the HIP kernels in `ops`.
"""
import math

import numpy as np
import torch

from . import ops, healpix

viewreal = torch.view_as_real
viewcomp = torch.view_as_complex

D2R = math.pi / 180.0


def _float(numpy=False):
    """default real dtype, following torch.set_default_dtype (utils.py:52-63)"""
    ft = torch.get_default_dtype()
    if not numpy:
        return ft
    return {torch.float16: np.float16, torch.float32: np.float32, torch.float64: np.float64}[ft]


def _cfloat(float_type=None, numpy=False):
    """complex dtype paired with the default real dtype (utils.py:66-82)"""
    ft = float_type if float_type is not None else torch.get_default_dtype()
    if not numpy:
        return {torch.float64: torch.complex128, torch.float32: torch.complex64,
                torch.float16: torch.complex32}[ft]
    return {torch.float64: np.complex128, torch.float32: np.complex64}[ft]


def colat2lat(theta, deg=True):
    """co-latitude <-> latitude (utils.py:110)"""
    return (90.0 - theta) if deg else (math.pi / 2 - theta)


# ---------------------------------------------------------------------------------------
# hashing / device helpers
# ---------------------------------------------------------------------------------------
def arr_hash(arr, pntr=False):
    """
    Weak array 'hash' = hash((first, last, len)), cached on the object as `_arr_hash`
    (utils.py:1643-1680).  Kept weak on purpose: the reference's caches key on it and RIME
    overrides it with (sky name, Npix, time) tuples (rime_model.py:345-357).
    """
    if pntr:
        return id(arr)
    if hasattr(arr, '_arr_hash'):
        return arr._arr_hash
    if isinstance(arr, torch.Tensor):
        h = hash((arr[0].cpu().item(), arr[-1].cpu().item(), len(arr)))
        arr._arr_hash = h
        return h
    return hash((arr[0], arr[-1], len(arr)))


def parse_device(d):
    if d is None:
        return 'cpu'
    if isinstance(d, torch.device):
        return d.type if d.index is None else '%s:%d' % (d.type, d.index)
    return d


def check_devices(d1, d2):
    """True when two device specs name the same device; None == cpu (utils.py:1785)"""
    a, b = parse_device(d1), parse_device(d2)
    if a == 'cuda':
        a = 'cuda:0'
    if b == 'cuda':
        b = 'cuda:0'
    return a == b


def push(tensor, device, parameter=False):
    """move (or re-type, when `device` is a dtype) a tensor, keeping Parameter-ness (utils.py:1683)"""
    if tensor is None:
        return None
    if isinstance(device, torch.dtype):
        dt = device
        if tensor.is_complex() and not dt.is_complex:
            dt = _cfloat(dt)
        elif not tensor.is_floating_point() and not tensor.is_complex():
            return tensor
        device = None
    else:
        dt = None
    is_param = parameter or isinstance(tensor, torch.nn.Parameter)
    h = getattr(tensor, '_arr_hash', None)
    out = tensor.detach().to(device=device, dtype=dt) if device is not None else tensor.detach().to(dt)
    if is_param:
        out = torch.nn.Parameter(out)
    if h is not None:
        out._arr_hash = h
    return out


def tensor2numpy(tensor, clone=True):
    if isinstance(tensor, torch.Tensor):
        t = tensor.detach().cpu()
        return t.clone().numpy() if clone else t.numpy()
    return tensor


def flatten(arr, Nelem=None):
    """flatten one nesting level of a list of lists (utils.py:2038)"""
    out = []
    for a in arr:
        out.extend(list(a))
    return out


def split_into_groups(arr, Nelem=None, Ngroup=None, interleave=False):
    """split a list/array into chunks of Nelem (or Ngroup chunks) (utils.py:1976-2013)"""
    N = len(arr)
    if Nelem is not None:
        assert Ngroup is None
    if interleave:
        if Ngroup is None:
            Ngroup = int(np.ceil(N / Nelem))
        groups = [arr[i::Ngroup] for i in range(Ngroup)]
    else:
        if Nelem is None:
            Nelem = int(np.ceil(N / Ngroup))
        groups = [arr[i:i + Nelem] for i in range(0, N, Nelem)]
    return [g for g in groups if len(g) > 0]


class SimpleIndex:
    """returns `value` for any key: the default ant2beam (utils.py:1965)"""
    def __init__(self, value=0):
        self.value = value

    def __getitem__(self, k):
        return self.value


# ---------------------------------------------------------------------------------------
# baselines / antennas
# ---------------------------------------------------------------------------------------
def _list2slice(inds):
    """an index list / tuple / tensor / range / int as a slice when it is an increasing arithmetic run, else
    unchanged (utils.py:2108-2133)"""
    if isinstance(inds, range) and inds.step > 0:
        return slice(inds.start, inds.stop, inds.step)
    if isinstance(inds, (int, np.integer)):
        return slice(int(inds), int(inds) + 1)
    if isinstance(inds, (list, tuple, torch.Tensor, np.ndarray)):
        if len(inds) == 0:
            return inds
        if len(inds) == 1:
            return slice(int(inds[0]), int(inds[0]) + 1, 1)
        steps = set(np.diff(inds.cpu() if isinstance(inds, torch.Tensor) else inds).tolist())
        if len(steps) == 1:
            step = int(steps.pop())
            if step > 0:
                return slice(int(inds[0]), int(inds[-1]) + step, step)
    return inds


def _slice2tensor(obj, device=None):
    """a slice as the integer tensor it selects (utils.py:2136-2144)"""
    if isinstance(obj, slice):
        obj = torch.arange(obj.start if obj.start is not None else 0, obj.stop,
                           obj.step if obj.step is not None else 1, device=device)
    return obj


def _idx2ten(idx, device=None):
    """a 1-D index list / array (boolean masks included) as an integer tensor on `device`; slices and ints pass
    through (utils.py:2147-2163)"""
    if isinstance(idx, (list, np.ndarray, tuple)):
        if isinstance(idx, np.ndarray) and idx.dtype == np.dtype(bool):
            idx = torch.as_tensor(np.where(idx)[0])
        else:
            idx = torch.as_tensor(idx, dtype=torch.long)
    if isinstance(idx, torch.Tensor) and idx.dtype == torch.bool:
        idx = torch.where(idx)[0]
    if device is not None and isinstance(idx, torch.Tensor):
        idx = idx.to(device)
    return idx


def blnum2ants(blnum, separate=False):
    """baseline integer(s) 1000*(a1+100)+(a2+100) -> antenna pair(s) (utils.py:2352)"""
    if isinstance(blnum, tuple):
        return blnum
    if isinstance(blnum, list) and len(blnum) and isinstance(blnum[0], tuple):
        return list(zip(*blnum)) if separate else blnum
    if isinstance(blnum, (int, np.integer)):
        a1 = int(blnum) // 1000
        return (a1 - 100, int(blnum) - a1 * 1000 - 100)
    if isinstance(blnum, torch.Tensor):
        blnum = blnum.cpu().numpy()
    b = np.asarray(blnum).astype(np.int64)
    a1 = b // 1000
    a2 = b - a1 * 1000 - 100
    a1 = a1 - 100
    if separate:
        return a1.tolist(), a2.tolist()
    return list(zip(a1.tolist(), a2.tolist()))


def ants2blnum(antnums, separate=False, tensor=False):
    """antenna pair(s) -> baseline integer(s) (utils.py:2416)"""
    if isinstance(antnums, tuple) and not separate:
        return int((antnums[0] + 100) * 1000 + antnums[1] + 100)
    if separate:
        a1, a2 = np.asarray(antnums[0]), np.asarray(antnums[1])
    else:
        a = np.asarray(antnums)
        a1, a2 = a[:, 0], a[:, 1]
    out = (a1 + 100) * 1000 + (a2 + 100)
    return torch.as_tensor(out) if tensor else out.tolist()


def conjbl(bl):
    if isinstance(bl, tuple):
        return bl[::-1]
    a1, a2 = blnum2ants(int(bl))
    return ants2blnum((a2, a1))


class AntposDict:
    """
    dict-like antenna number -> ENU position [m], positions held in one (Nants, 3) tensor
    (utils.py:2280-2349)
    """
    def __init__(self, ants, antvecs):
        self.ants = list(ants)
        self._ant_idx = {a: i for i, a in enumerate(self.ants)}
        try:
            self.antvecs = torch.as_tensor(antvecs)
        except (ValueError, TypeError):
            self.antvecs = torch.vstack([torch.as_tensor(v) for v in antvecs])
        if self.antvecs.dtype not in (torch.float32, torch.float64):
            self.antvecs = self.antvecs.to(_float())

    def keys(self):
        return (a for a in self.ants)

    def values(self):
        return (v for v in self.antvecs)

    def items(self):
        return zip(self.ants, self.antvecs)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.antvecs[self._ant_idx[key]]
        if isinstance(key, torch.Tensor):
            key = key.tolist()
        return self.antvecs[[self._ant_idx[k] for k in key]]

    def __setitem__(self, key, value):
        self.antvecs[self._ant_idx[key]] = value

    def __repr__(self):
        return 'Antpos{{{}}}'.format(self.ants)

    def __len__(self):
        return len(self.ants)

    def __contains__(self, key):
        return key in self._ant_idx

    def __iter__(self):
        return self.keys()

    def push(self, device):
        self.antvecs = push(self.antvecs, device)

    def select(self, new_ants):
        return AntposDict(new_ants, self.antvecs[[self._ant_idx[a] for a in new_ants]])


def _make_hex(N, D=15):
    """
    Hexagonal array with N antennas on a side (2N-1 rows), spacing D [m], centred on the
    origin, numbered row by row from the south-west; returns (ants, (Nant, 3) ENU array)
    (utils.py:1943-1962).
    """
    xs, ys = [], []
    for row in range(2 * N - 1):
        n_in_row = N + (row if row < N else 2 * N - 2 - row)
        x0 = -0.5 * (n_in_row - N)
        for j in range(n_in_row):
            xs.append(x0 + j)
            ys.append(row * math.sin(math.pi / 3))
    x = np.array(xs) - np.mean(xs)
    y = np.array(ys) - np.mean(ys)
    vecs = np.vstack([x, y, np.zeros_like(x)]).T * D
    return list(range(len(xs))), vecs


# ---------------------------------------------------------------------------------------
# Module base (utils.py:1123-1320) and attribute helpers (utils.py:1414-1557)
# ---------------------------------------------------------------------------------------
def has_model_attr(model, name):
    parts = name.split('.') if isinstance(name, str) else list(name)
    obj = model
    for p in parts:
        if not hasattr(obj, p):
            return False
        obj = getattr(obj, p)
    return True


def get_model_attr(model, name, pop=0):
    parts = name.split('.') if isinstance(name, str) else list(name)
    if pop > 0:
        parts = parts[:-pop]
    obj = model
    for p in parts:
        obj = getattr(obj, p)
    return obj


def set_model_attr(model, name, value, clobber_param=False, no_grad=True, idx=None, add=False,
                   fill=None):
    """
    Assign `value` to model.<dotted name> with the reference's semantics (utils.py:1453-1545):
    an existing Parameter is stripped to its data first and re-wrapped afterwards unless
    clobber_param; idx/add/fill modify the existing tensor in place.  Assigning a non-leaf graph
    tensor (what optim.LogProb.set_main_params does) therefore works on every module here.
    """
    parts = name.split('.') if isinstance(name, str) else list(name)
    if len(parts) > 1:
        return set_model_attr(get_model_attr(model, parts[:-1]), parts[-1], value,
                              clobber_param=clobber_param, no_grad=no_grad, idx=idx, add=add, fill=fill)
    attr = parts[0]
    ctx = torch.no_grad() if no_grad else _nullcontext()
    with ctx:
        cur = getattr(model, attr, None)
        if cur is None:
            setattr(model, attr, value)
            return
        was_param = isinstance(cur, torch.nn.Parameter)
        if clobber_param or was_param:
            data = cur.data
            delattr(model, attr)
            setattr(model, attr, data)
            cur = getattr(model, attr)
        if isinstance(value, torch.Tensor) and not check_devices(cur.device, value.device):
            value = value.to(cur.device)
        if fill is not None:
            cur.data[:] = fill.to(cur.dtype) if isinstance(fill, torch.Tensor) else fill
        if add:
            if idx is None:
                cur += value
            else:
                cur[idx] += value
        elif idx is None:
            setattr(model, attr, value)
        else:
            cur[idx] = value
        if was_param and not clobber_param:
            setattr(model, attr, torch.nn.Parameter(getattr(model, attr)))


def del_model_attr(model, name):
    parts = name.split('.') if isinstance(name, str) else list(name)
    obj = get_model_attr(model, parts[:-1]) if len(parts) > 1 else model
    delattr(obj, parts[-1])


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class Module(torch.nn.Module):
    """
    Thin torch.nn.Module with the reference's extras: a `name`, dotted get/set/del of
    attributes (module['sky.params']), priors on input/output params evaluated into a
    prior_cache, response gradient hooks (utils.py:1123-1320).
    """
    def __init__(self, name=None):
        super().__init__()
        self.__version__ = '0.1.0'
        self.set_priors()
        self._name = name

    @property
    def name(self):
        return self._name if self._name is not None else self.__class__.__name__

    @property
    def named_params(self):
        return [k for k, _ in self.named_parameters()]

    def forward(self, inp=None, prior_cache=None, **kwargs):
        raise NotImplementedError

    def __getitem__(self, name):
        return get_model_attr(self, name)

    def __setitem__(self, name, value):
        with torch.no_grad():
            set_model_attr(self, name, value)

    def __delitem__(self, name):
        del_model_attr(self, name)

    def update(self, pdict, clobber_param=False):
        for k, v in pdict.items():
            set_model_attr(self, k, v, clobber_param=clobber_param)

    def unset_param(self, name):
        if isinstance(name, list):
            for n in name:
                self.unset_param(n)
            return
        p = self[name].detach()
        del self[name]
        self[name] = p

    def set_param(self, name):
        if isinstance(name, list):
            for n in name:
                self.set_param(n)
            return
        p = self[name]
        if not isinstance(p, torch.nn.Parameter):
            self[name] = torch.nn.Parameter(p)

    def set_priors(self, priors_inp_params=None, priors_out_params=None):
        as_list = lambda p: p if (p is None or isinstance(p, (list, tuple))) else [p]
        self.priors_inp_params = as_list(priors_inp_params)
        self.priors_out_params = as_list(priors_out_params)

    def eval_prior(self, prior_cache, inp_params=None, out_params=None):
        """sum of log-priors into prior_cache[self.name], once per key (utils.py:1237-1287)"""
        if prior_cache is None or self.name in prior_cache:
            return
        val = torch.as_tensor(0.0)
        if inp_params is None and hasattr(self, 'params'):
            inp_params = self.params
        if self.priors_inp_params is not None and inp_params is not None:
            for pr in self.priors_inp_params:
                if pr is not None:
                    val = val + pr(inp_params)
        if self.priors_out_params is not None:
            if out_params is None and hasattr(self, 'params') and hasattr(self, 'R'):
                p = self.params
                if getattr(self, 'p0', None) is not None:
                    p = p + self.p0
                out_params = self.R(p)
            if out_params is not None:
                for pr in self.priors_out_params:
                    if pr is not None:
                        val = val + pr(out_params)
        prior_cache[self.name] = val

    def register_response_hooks(self, registry=None):
        if registry is not None and not isinstance(registry, (list, tuple)):
            registry = [registry]
        self._hook_registry = registry

    def clear_graph_tensors(self):
        pass


class Sequential(Module):
    """
    Chain of Modules evaluated in order, output of one feeding the next, with the
    minibatch protocol (Nbatch / batch_idx forwarded to the first model) (utils.py:1323-1411).
    """
    def __init__(self, models):
        super().__init__()
        self._models = list(models)
        for k, m in models.items():
            self.add_module(k, m)

    def forward(self, inp=None, pdict=None, prior_cache=None, **kwargs):
        if pdict is not None:
            self.update(pdict)
        for i, k in enumerate(self._models):
            m = getattr(self, k)
            inp = m(inp, prior_cache=prior_cache, **kwargs)
        return inp

    @property
    def Nbatch(self):
        return getattr(getattr(self, self._models[0]), 'Nbatch', 1)

    @property
    def batch_idx(self):
        return getattr(getattr(self, self._models[0]), 'batch_idx', 0)

    @batch_idx.setter
    def batch_idx(self, val):
        getattr(self, self._models[0]).batch_idx = val

    def push(self, device):
        for k in self._models:
            getattr(self, k).push(device)


# ---------------------------------------------------------------------------------------
# PixInterp (utils.py:684-878)
# ---------------------------------------------------------------------------------------
_S2D = {'nearest': 0, 'linear': 1, 'quadratic': 2, 'cubic': 3}


def clear_cache_depth(cache, depth):
    """FIFO-trim an (insertion-ordered) dict to `depth` entries (utils.py:881)"""
    if depth is None:
        return
    for k in list(cache.keys())[:max(0, len(cache) - depth)]:
        del cache[k]


def _stencil_start(t, n, N, wrap):
    # first node of the n nearest grid nodes around fractional index t, ascending; the
    # reference gets the same set from argsort(|grid - x|) (utils.py:1003-1004)
    if n % 2 == 0:
        s = torch.ceil(t).to(torch.int64) - n // 2
    else:
        s = torch.ceil(t - 0.5).to(torch.int64) - n // 2
    return s if wrap else s.clamp(0, N - n)


def _lagrange(u, n):
    cols = []
    for j in range(n):
        w = torch.ones_like(u)
        for k in range(n):
            if k != j:
                w = w * ((u - k) / (j - k))
        cols.append(w)
    return torch.stack(cols, dim=-1)


def bipoly_interp_weights(xgrid, ygrid, xnew, ynew, degree, wrapx=True):
    """
    Closed-form replacement of bipoly_grid_index + setup_bipoly_interp (utils.py:949-1116) on
    a uniform grid: stencil = (degree+1) nearest nodes per axis (x periodic), weights =
    tensor-product Lagrange basis (what `Anew @ pinv(A^T A) A^T` evaluates to).  Returns
    inds (P, Nx*Ny) flat indices into the x-fastest raveled grid and wgts (P, Nx*Ny),
    stencil ordered y-slow / x-fast.  float64 arithmetic regardless of the default dtype.
    """
    nx, ny = degree[0] + 1, degree[1] + 1
    xg, yg = xgrid.to(torch.float64), ygrid.to(torch.float64)
    xn, yn = xnew.to(torch.float64), ynew.to(torch.float64)
    Nx, Ny = len(xg), len(yg)
    tx = (xn - xg[0]) / (xg[1] - xg[0])
    ty = (yn - yg[0]) / (yg[1] - yg[0])
    sx = _stencil_start(tx, nx, Nx, wrapx)
    sy = _stencil_start(ty, ny, Ny, False)
    wx = _lagrange(tx - sx.to(tx.dtype), nx)
    wy = _lagrange(ty - sy.to(ty.dtype), ny)
    ar_x = torch.arange(nx, device=sx.device)
    ar_y = torch.arange(ny, device=sx.device)
    ix = (sx[:, None] + ar_x) % Nx if wrapx else (sx[:, None] + ar_x).clamp(0, Nx - 1)
    iy = sy[:, None] + ar_y
    inds = (ix[:, None, :] + Nx * iy[:, :, None]).reshape(len(xn), -1)
    wgts = (wx[:, None, :] * wy[:, :, None]).reshape(len(xn), -1)
    return inds, wgts


class PixInterp:
    """
    Interpolation of a pixelised map at arbitrary (zen, az): a weighted sum of nearest
    neighbours whose (indices, weights) are cached per angle set (utils.py:684-878).
    'rect': bi-polynomial on a uniform (theta, phi) grid, modes nearest / linear / quadratic /
    cubic or 'az_mode,zen_mode'.  'healpix': RING bilinear (own implementation of the HEALPix
    scheme healpy.get_interp_weights follows -- parity unpinned, healpy is not available).
    The gather itself and its adjoint run in HIP (ops.interp_gather).
    """
    def __init__(self, pixtype, nside=None, interp_mode='nearest', theta_grid=None, phi_grid=None,
                 device=None, interp_cache_depth=None):
        assert pixtype in ('healpix', 'rect'), "pixtype must be 'healpix' or 'rect'"
        self.pixtype = pixtype
        self.nside = nside
        self.interp_cache = {}
        self.interp_mode = interp_mode
        self.theta_grid = theta_grid
        self.phi_grid = phi_grid
        self.device = device
        self.interp_cache_depth = interp_cache_depth

    def clear_cache(self, depth=None):
        if depth is None:
            self.interp_cache = {}
        else:
            clear_cache_depth(self.interp_cache, depth)

    def _npix_map(self):
        if self.pixtype == 'healpix':
            return 12 * self.nside ** 2
        return len(self.theta_grid) * len(self.phi_grid)

    def _compute_interp(self, zen, az):
        if self.pixtype == 'healpix':
            inds, wgts = healpix.get_interp_weights(self.nside, tensor2numpy(zen) * D2R,
                                                    tensor2numpy(az) * D2R)
            return torch.as_tensor(inds.T.copy()), torch.as_tensor(wgts.T.copy())
        mode = self.interp_mode
        deg = [mode, mode] if ',' not in mode else [s.strip() for s in mode.split(',')]
        deg = [_S2D[d] for d in deg]
        az, zen = torch.as_tensor(az), torch.as_tensor(zen)
        return bipoly_interp_weights(torch.as_tensor(self.phi_grid).to(az.device),
                                     torch.as_tensor(self.theta_grid).to(az.device),
                                     az, zen, deg, wrapx=True)

    def get_stencil(self, zen, az):
        """cached ops.InterpStencil for this angle set (keyed like the reference: arr_hash(zen))"""
        h = arr_hash(zen)
        st = self.interp_cache.get(h)
        if st is None:
            inds, wgts = self._compute_interp(zen, az)
            dev = self.device if self.device is not None else getattr(zen, 'device', 'cpu')
            if not str(dev).startswith('cuda'):
                raise RuntimeError("PixInterp needs a GPU device (got %r): bayeslim_amd has no CPU "
                                   "interpolation path" % (dev,))
            st = ops.InterpStencil(inds.to(dev), wgts.to(_float()).to(dev), self._npix_map())
            if self.interp_cache_depth is None or self.interp_cache_depth > 0:
                self.interp_cache[h] = st
                if self.interp_cache_depth is not None:
                    self.clear_cache(depth=self.interp_cache_depth)
        return st

    def get_interp(self, zen, az):
        """(inds (P, Nnn) int, wgts (P, Nnn)) as the reference returns them (utils.py:742-813)"""
        st = self.get_stencil(zen, az)
        return st.inds.to(torch.int64), st.wgts

    def interp(self, m, zen, az, out_stride=None):
        """m (..., Npix_map) -> (..., P [padded to out_stride]) (utils.py:815-861)"""
        st = self.get_stencil(zen, az)
        if not m.is_cuda:
            raise RuntimeError('PixInterp.interp: map must live on the GPU')
        return ops.interp_gather(m, st, out_stride)

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
        self.interp_cache = {}          # stencils are rebuilt on the new device / dtype
        if self.theta_grid is not None:
            self.theta_grid = push(torch.as_tensor(self.theta_grid), device)
        if self.phi_grid is not None:
            self.phi_grid = push(torch.as_tensor(self.phi_grid), device)
