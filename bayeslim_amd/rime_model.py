"""
`RIME`: the radio interferometric measurement equation V_pq = sum_pix A_p B A_q^dagger K_pq,
with the reference's constructor, attributes and minibatch protocol (rime_model.py:13-482).

What differs from the reference is HOW forward() computes:
  * the reference loops over times in Python and, per time, materialises an (Nbl, Nf, P)
    complex fringe and the (.., Nbl, Nf, P) product before summing (rime_model.py:334-365,
    423-429).  Here every time step of the minibatch is prepared (FoV cut, beam interpolation,
    beam x sky per beam-model pair) and ONE fused HIP kernel (ops.fringe_sum) produces
    vis[pp, bl, t, f] for all baselines, times and channels; backward is one fused kernel too.
  * per-time geometry (pointing vectors, FoV cut, interpolation stencil) is cached under the
    same keys the reference uses: (sky name, Npix, time).
Output: `VisData` with data (Npol, Npol, Nbl_data, Ntimes, Nfreqs), exactly as the reference.
"""
from datetime import datetime

import numpy as np
import torch

from . import utils, ops, beam_model, dataset, telescope_model
from .dataset import VisData


class RIME(utils.Module):
    def __init__(self, sky, telescope, beam, array, sim_bls, times, freqs, data_bls=None,
                 device=None, cache_eq2top=True, name=None, verbose=False):
        super().__init__(name=name)
        self.sky = sky
        self.telescope = telescope
        self.beam = beam
        self.array = array
        self.device = device
        self.cache_eq2top = cache_eq2top
        self.fuse_beam_sky = True       # 1-pol power beam: ops.beam_sky_product instead of interp + cut + product
        self.verbose = verbose
        self.clear_geometry_cache()
        self.setup_freqs(freqs)
        self.setup_sim_bls(sim_bls, data_bls)
        self.setup_sim_times(times=times)

    # -- device / caches -----------------------------------------------------------------
    def clear_geometry_cache(self):
        self._zenaz_cache = {}      # key -> (zen, az) float64 on the compute device
        self._npix_cache = {}       # key -> pixels surviving the FoV cut
        self._geom_cache = {}       # (bl group, time group, sky name, Npix) -> FringeGeometry
        self._ant_like = {}         # bl group -> first FringeGeometry built for it (shares its pair tables)
        self._mp_cache = {}         # bl group -> (modelpairs, pair index per baseline)

    # derived state that is rebuilt on demand and must not travel with a pickled / deep-copied model (io.py:50-66 pickles whole
    # models, optim.py:1517-1523 and notebook users deepcopy them): geometry entries hold ctypes tables and device buffers and
    # are keyed on addresses (`data_ptr()`, `id()`) of the ORIGINAL's tensors -- in a copy those keys describe nothing
    _DERIVED = ('_zenaz_cache', '_npix_cache', '_geom_cache', '_ant_like', '_mp_cache', '_inflate_cache', '_blnum_cache')

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._DERIVED:
            if k in state:
                state[k] = {}
        return state

    def push(self, device):
        self.sim_blvec_groups = {k: v.to(device) for k, v in self.sim_blvec_groups.items()}
        if not isinstance(device, torch.dtype):
            self.device = device
            for k, v in self._sim2data.items():
                if v is not None:
                    self._sim2data[k] = utils.push(v, device)
        self.clear_geometry_cache()

    @property
    def Ntimes_all(self):
        return len(self.all_sim_times)

    @property
    def Nbls_all(self):
        return len(self.all_sim_bls)

    def setup_freqs(self, freqs):
        self.freqs = freqs
        self.Nfreqs = len(freqs)
        self._geom_cache = {}

    # -- baseline groups (rime_model.py:148-226) -------------------------------------------
    def setup_sim_bls(self, sim_bls, data_bls=None):
        self.bl_group_id = 0
        if isinstance(sim_bls, dict):
            groups = {k: [tuple(b) for b in v] for k, v in sim_bls.items()}
        else:
            ints = (int, np.integer)
            assert isinstance(sim_bls[0][0], ints) or isinstance(sim_bls[0][0][0], ints), \
                "sim_bls must be list of 2-tuples or list of list of 2-tuples"
            if isinstance(sim_bls[0][0], ints):
                groups = {0: [tuple(b) for b in sim_bls]}
            else:
                groups = {i: [tuple(b) for b in g] for i, g in enumerate(sim_bls)}
        if data_bls is not None:
            data_bls = [tuple(b) for b in data_bls]
        self.sim_bl_groups = groups
        self.all_sim_bls = utils.flatten(groups.values())
        self.Nbl_groups = len(groups)
        self.sim_blvec_groups = {k: self.array.get_blvecs(v) for k, v in groups.items()}
        if data_bls is None:
            self.data_bl_groups = self.sim_bl_groups
            self._sim2data = {k: None for k in groups}
        else:
            self._sim2data, self.data_bl_groups = {}, {}
            b2r = self.array.bl2red
            for k, blg in groups.items():
                sim_red = [b2r[bl] for bl in blg]
                keep = set(sim_red)
                dbl = [bl for bl in data_bls if b2r[bl] in keep]
                data_red = [b2r[bl] for bl in dbl]
                assert set(sim_red) == set(data_red), "non-overlapping bl type(s) in data_bls and sim_bls"
                # redundant baselines must be contiguous in data_bls
                assert len(np.where(np.diff(data_red) != 0)[0]) == len(blg) - 1
                self.data_bl_groups[k] = dbl
                pos = {r: i for i, r in reversed(list(enumerate(sim_red)))}
                self._sim2data[k] = torch.as_tensor([pos[r] for r in data_red], device=self.device)
        self._geom_cache, self._mp_cache, self._ant_like = {}, {}, {}
        self._set_group()

    # -- time groups (rime_model.py:228-251) -------------------------------------------------
    def setup_sim_times(self, times):
        self.time_group_id = 0
        if not isinstance(times, dict):
            if isinstance(times, list) or (isinstance(times, (np.ndarray, torch.Tensor)) and times.ndim > 1):
                times = {k: np.asarray(t) for k, t in enumerate(times)}
            else:
                times = {0: np.asarray(times)}
        self.sim_time_groups = times
        self.all_sim_times = np.asarray(utils.flatten(times.values()))
        self.Ntime_groups = len(times)
        self._geom_cache = {}
        self._set_group()

    @property
    def Nbatch(self):
        if hasattr(self, 'sim_bl_groups') and hasattr(self, 'sim_time_groups'):
            return len(self.sim_bl_groups) * len(self.sim_time_groups)
        return None

    @property
    def batch_idx(self):
        """time_group_id * Nbl_groups + bl_group_id: baselines vary fastest (rime_model.py:55-57)"""
        if hasattr(self, 'bl_group_id') and hasattr(self, 'time_group_id'):
            return self.time_group_id * len(self.sim_bl_groups) + self.bl_group_id
        return None

    @batch_idx.setter
    def batch_idx(self, val):
        assert 0 <= val < self.Nbatch
        self.time_group_id = int(val // len(self.sim_bl_groups))
        self.bl_group_id = int(val % len(self.sim_bl_groups))
        self._set_group()

    def _set_group(self):
        if hasattr(self, 'sim_bl_groups'):
            self.sim_bls = self.sim_bl_groups[self.bl_group_id]
            self.sim_blvecs = self.sim_blvec_groups[self.bl_group_id]
            self.Nsim_bls = len(self.sim_bls)
            self.data_bls = self.data_bl_groups[self.bl_group_id]
            self.Ndata_bls = len(self.data_bls)
        if hasattr(self, 'sim_time_groups'):
            self.sim_times = self.sim_time_groups[self.time_group_id]
            self.Ntimes = len(self.sim_times)

    # -- forward ---------------------------------------------------------------------------
    def _current_blvecs(self):
        """baseline vectors of the current group; re-derived from the antenna positions when those carry a
        gradient (the reference derives them once at setup, rime_model.py:193, so its graph to antvecs survives a
        single backward only), else the tensor cached at setup"""
        if getattr(self.array.antvecs, 'requires_grad', False):
            return self.array.get_blvecs(self.sim_bls)
        return self.sim_blvecs

    def _compute_device(self, sky):
        dev = sky.device
        if dev.type != 'cuda':
            raise RuntimeError("RIME.forward: the sky model lives on '%s'; bayeslim_amd computes on the "
                               "GPU only (push the models to 'cuda')" % dev)
        return dev

    def _zenaz(self, key, time, ra, dec, dev):
        src = self.telescope.conv_cache.get(key)
        hit = self._zenaz_cache.get(key)
        if hit is not None and hit[0] is src and src is not None:
            return hit[1]
        angs = self.telescope.eq2top(time, ra, dec, store=self.cache_eq2top, key=key)
        zen = torch.as_tensor(angs[0]).to(device=dev, dtype=torch.float64)
        az = torch.as_tensor(angs[1]).to(device=dev, dtype=torch.float64)
        # remembered together with the conv_cache entry it came from (the OBJECT, held alive here, compared with
        # `is`: an id() alone can be handed to a new entry after clear_cache()): replacing that entry (new
        # coordinates for the same key) is seen on the next forward
        self._zenaz_cache[key] = (self.telescope.conv_cache.get(key), (zen, az))
        return zen, az

    def _batch_geometry(self, name, ra, dec, Npix, dev, pairs, bl_mp):
        """
        Geometry of one (baseline group, time group, sky component), built once and cached:
        per time step the FoV cut (strict zen < fov/2, beam_model.py:221-224), the cut angles and
        the float64 pointing vectors (telescope_model.py:337-343), padded to a common stride Ps
        (multiple of 64) and concatenated over the Nt time steps of the minibatch.
        """
        # everything the cached entry bakes in is part of its key: the FoV cut, the antenna positions (tensor
        # identity + in-place version), the channel grid and the matrix-core grouping
        av = self.array.antvecs
        fq = self.freqs
        cc = self.telescope.conv_cache
        gkey = (self.bl_group_id, self.time_group_id, name, Npix, float(self.beam.fov), av.data_ptr(), av._version,
                (fq.data_ptr(), fq._version) if isinstance(fq, torch.Tensor) else id(fq),
                getattr(self, 'mfma_group', None), getattr(self, 'mfma_mode', 'auto'),
                tuple(id(cc.get((name, Npix, float(t)))) for t in self.sim_times))      # replaced (zen, az) entries
        bg = self._geom_cache.get(gkey)
        if bg is not None:
            return bg
        for k in [k for k in self._geom_cache if k[:4] == gkey[:4]]:        # superseded entries of this minibatch
            del self._geom_cache[k]
        # the reference's fringe uses array.freqs[array._freq_idx] (telescope_model.py:350); here RIME.freqs
        # feeds the kernels, so the two must describe the same channels
        af = self.array._freqs_active() if getattr(self.array, 'freqs', None) is not None else None
        if af is not None:
            a64 = torch.as_tensor(af).detach().to('cpu', torch.float64)
            r64 = torch.as_tensor(self.freqs).detach().to('cpu', torch.float64)
            if a64.shape != r64.shape or not torch.allclose(a64, r64, rtol=1e-6, atol=0.0):
                raise ValueError('RIME.freqs and the ArrayModel\'s active frequencies (array.freqs[_freq_idx]) differ: '
                                 'the fringe is evaluated at RIME.freqs')
        keys = [(name, Npix, float(t)) for t in self.sim_times]
        za = [self._zenaz(k, t, ra, dec, dev) for k, t in zip(keys, self.sim_times)]
        cuts = []
        for k, (zen, az) in zip(keys, za):
            cut = self.beam.fov_cut(zen)
            cuts.append(torch.arange(Npix, device=dev) if isinstance(cut, slice) else cut.to(dev))
            self._npix_cache[k] = int(cuts[-1].numel())
        Nt = len(keys)
        Ps = ops.pad_to_tile(max(max(c.numel() for c in cuts), 1))
        zen_all = torch.zeros(Nt * Ps, dtype=torch.float64, device=dev)
        az_all = torch.zeros(Nt * Ps, dtype=torch.float64, device=dev)
        cut_all = torch.full((Nt * Ps,), Npix, dtype=torch.int64, device=dev)
        sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64, device=dev)
        for j, ((zen, az), cut) in enumerate(zip(za, cuts)):
            P = cut.numel()
            zc, ac = zen[cut], az[cut]
            zen_all[j * Ps:j * Ps + P] = zc
            az_all[j * Ps:j * Ps + P] = ac
            cut_all[j * Ps:j * Ps + P] = cut
            sdir[j, :, :P] = telescope_model.pointing_vectors(zc, ac)
        # cache key for the interpolation stencil / Ylm of this angle set (rime_model.py:345-357
        # uses (sky name, Npix, time) per time step; here the whole time group is one set)
        zen_all._arr_hash = ('rime-batch', name, Npix, tuple(float(t) for t in self.sim_times), Ps)
        # antenna positions + antenna-index pairs let the fringe op take its matrix-core path
        idx = self.array._ant_idx
        bl_ants = [(idx[b[0]], idx[b[1]]) for b in self.sim_bls]
        # pair tables of the matrix-core path depend on the baseline group only: built once, shared by
        # the geometries of all time minibatches
        # (keyed on the antenna positions too: the blocks hold them)
        lkey = (self.bl_group_id, av.data_ptr(), av._version, getattr(self, 'mfma_group', None), getattr(self, 'mfma_mode', 'auto'))
        like = self._ant_like.get(lkey)
        geom = ops.FringeGeometry(self._current_blvecs().detach().to(dev), sdir, self.freqs, bl_mp=bl_mp,
                                  Nmp=len(pairs), npix=[c.numel() for c in cuts],
                                  antpos=self.array.antvecs, bl_ants=bl_ants, ant_like=like, mp_pairs=pairs,
                                  group=getattr(self, 'mfma_group', None), mfma=getattr(self, 'mfma_mode', 'auto'))
        if like is None:
            for k in [k for k in self._ant_like if k[0] == self.bl_group_id and k[3:] == lkey[3:]]:      # tables of superseded positions
                del self._ant_like[k]
            self._ant_like[lkey] = geom
        # for the fused psky builder: int32 cut and its inverse per time step
        pos = torch.full((Nt, Npix), -1, dtype=torch.int32, device=dev)
        for j, cut in enumerate(cuts):
            pos[j, cut] = torch.arange(cut.numel(), dtype=torch.int32, device=dev)
        # `alive`: the objects whose addresses the key is made of stay referenced for as long as the entry lives, so
        # neither CPython nor the caching allocator can hand the same id() / data_ptr() to a successor
        bg = dict(zen=zen_all, az=az_all, cut=cut_all, geom=geom, Nt=Nt, Ps=Ps,
                  cut32=cut_all.to(torch.int32), pos=pos,
                  alive=(av, fq, tuple(cc.get((name, Npix, float(t))) for t in self.sim_times)))
        self._geom_cache[gkey] = bg
        return bg

    def forward(self, *args, prior_cache=None, **kwargs):
        """sky -> beam -> fused fringe sum -> VisData (rime_model.py:291-389)"""
        self._set_group()
        comps = self.sky.forward(prior_cache=prior_cache)
        if not isinstance(comps, list):
            comps = [comps]
        Npol = self.beam.Npol
        pol = '{0}{0}'.format(self.beam.pol) if Npol == 1 else None
        if hasattr(self.beam.R, 'clear_beam_cache'):
            self.beam.R.clear_beam_cache()
        self.beam.skycut_device = self.sky.device
        if self.bl_group_id not in self._mp_cache:
            self._mp_cache[self.bl_group_id] = self.beam.modelpairs(self.sim_bls)
        pairs, bl_mp = self._mp_cache[self.bl_group_id]
        start = datetime.now().timestamp()

        vis = None
        node_major = {}
        for i, comp in enumerate(comps):
            sky = comp.data
            dev = self._compute_device(sky)
            ra, dec = comp.angs
            Npix = len(ra)
            bg = self._batch_geometry(comp.name, ra, dec, Npix, dev, pairs, bl_mp)
            if self.verbose:
                log('{} times for {}/{} sky model | {} elapsed'.format(
                    len(self.sim_times), i + 1, len(comps), elapsed_time(start)), verbose=True)
            Nt, Ps = bg['Nt'], bg['Ps']
            fused = None
            if self.fuse_beam_sky and len(pairs) == 1 and tuple(sky.shape[:2]) == (1, 1) and not sky.is_complex():
                fused = self.beam.response_map_and_stencil(bg['zen'], bg['az'], prior_cache=prior_cache)
                if fused is not None and fused[0].dtype != sky.dtype:
                    fused = None
            if fused is not None:
                # 1-pol power beam: interpolation, FoV cut and beam x sky in one pass (ops.beam_sky_product)
                bc, st = fused
                # leading axes of length 1 are dropped by reshape (a view in the backward too: indexing them away
                # costs a zero-fill + copy of the whole map / sky gradient per axis, 0.5 ms per C4 step)
                b2 = bc.reshape(bc.shape[-2:]) if bc.numel() == bc.shape[-2] * bc.shape[-1] else bc[0, 0, 0]
                # the node-major copy the builder gathers from: made once per forward and shared by the sky components (the
                # beam cache is one tensor object for the whole forward)
                if node_major.get('of') is not bc:
                    node_major.update(of=bc, bT=ops.node_major(b2))
                ps = ops.beam_sky_product(b2, sky.reshape(sky.shape[-2:]), st, bg['cut32'], bg['pos'], Nt, Ps,
                                          node_major_map=node_major['bT'])
                ps = ps.reshape(1, 1, 1, ps.shape[0], Nt * Ps)
            else:
                # beam at the FoV-cut angles of ALL time steps: one response evaluation / gather launch
                beam = self.beam.eval_response(bg['zen'], bg['az'], prior_cache=prior_cache)
                # beam_model.cut_sky_fov for all times at once, as a one-node gather (index = the cut, weight 1, 0 for the
                # padding of a time step): no zero-extended copy of the sky, and its adjoint is the CSR gather of the
                # interpolation kernels -- a sky pixel seen at several time steps is summed in a fixed order (the
                # scatter-add of index_select's backward uses atomics: the sky gradient differed in the last bits from run
                # to run)
                st = bg.get('cut_stencil')
                if st is None:
                    cut = bg['cut']
                    st = bg['cut_stencil'] = ops.InterpStencil(cut.clamp(max=Npix - 1).reshape(-1, 1),
                                                               (cut < Npix).to(torch.float64).reshape(-1, 1), Npix)
                cut_sky = ops.interp_gather(sky, st)
                ps = self.beam.apply_beam_mp(beam, cut_sky, pairs)   # (n1, n2, Nmp, Nf, Nt*Ps)
            n1, n2, Nmp, Nf = ps.shape[:4]
            # -> (Nt, Nmp, Npp, Nf, Ps) as a strided VIEW: the fringe kernels take the strides
            ps = ps.reshape(n1 * n2, Nmp, Nf, Nt, Ps).permute(3, 1, 0, 2, 4)
            blv = self._current_blvecs()
            v = ops.fringe_sum(ps, bg['geom'], blv.to(dev) if blv.requires_grad else None)   # (Npp, Nbl, Nt, Nf)
            v = v.reshape(n1, n2, v.shape[1], Nt, Nf)
            vis = v if vis is None else vis + v

        idx = self._sim2data[self.bl_group_id]
        if idx is not None:                                          # rime_model.py:436-437
            idx = idx.to(vis.device)
            st = self.__dict__.setdefault('_inflate_cache', {}).get(self.bl_group_id)
            if st is None or st[0] is not self._sim2data[self.bl_group_id]:
                dup = bool(idx.numel() > torch.unique(idx).numel())
                st = (self._sim2data[self.bl_group_id],
                      ops.InterpStencil(idx.reshape(-1, 1), torch.ones(idx.numel(), 1, dtype=torch.float64, device=idx.device),
                                        vis.shape[2]) if dup and vis.is_cuda else None)
                self._inflate_cache[self.bl_group_id] = st
            if st[1] is not None:
                # redundant inflation (one simulated baseline feeds several data baselines): a one-node gather whose adjoint
                # sums the copies in a fixed order (index_select's scatter-add backward uses atomics)
                vis = ops.interp_gather(vis.movedim(2, -1), st[1]).movedim(-1, 2).contiguous()
            else:
                vis = vis.index_select(2, idx)
        if self.device is not None and not utils.check_devices(vis.device, self.device):
            vis = vis.to(self.device)

        vd = VisData()
        tel = self.telescope.__class__(self.telescope.location, tloc=getattr(self.telescope, 'tloc', None),
                                       device=self.telescope.device)
        vd.setup_meta(tel, self.array.to_antpos())
        # the baseline integers of this group, converted once (8128 tuples -> numpy costs 1 ms of host time per forward)
        bn = self.__dict__.setdefault('_blnum_cache', {}).get(self.bl_group_id)
        if bn is None or bn[0] is not self.data_bls:
            bn = (self.data_bls, np.asarray(utils.ants2blnum([tuple(b) for b in self.data_bls]), dtype=np.int64)
                  if len(self.data_bls) else np.array([], dtype=np.int64))
            self._blnum_cache[self.bl_group_id] = bn
        vd.setup_data(bn[1].copy(), self.sim_times, self.freqs, pol=pol, data=vis, flags=None,
                      cov=None, history=self.describe())
        return vd

    def describe(self):
        """one-line model description stored as VisData.history (io.get_model_description analogue)"""
        return 'RIME(sky={}, beam={}[{}], Nbls={}, Ntimes={}, Nfreqs={})'.format(
            getattr(self.sky, 'name', type(self.sky).__name__), getattr(self.beam, 'name', 'beam'),
            type(self.beam.R).__name__, self.Nsim_bls, self.Ntimes, self.Nfreqs)

    def run_batches(self, concat=True):
        """forward() over all minibatches, concatenated over baselines then times (rime_model.py:442-482)"""
        per_time, per_bl = [], []
        for i in range(self.Nbatch):
            self.batch_idx = i
            per_bl.append(self.forward())
            if self.Nbatch == 1:
                per_time.append(per_bl[-1])
            elif self.bl_group_id == self.Nbl_groups - 1:
                if concat:
                    per_time.append(dataset.concat_VisData(per_bl, 'bl'))
                else:
                    per_time.extend(per_bl)
                per_bl = []
        out = dataset.concat_VisData(per_time, 'time') if concat else per_time
        self.batch_idx = 0
        return out


def log(message, verbose=False, style=1):
    if verbose:
        if style == 1:
            print(message)
        elif style == 2:
            print('{}\n{}'.format(message, '-' * 30))
        else:
            print('\n{}\n{}\n{}'.format('-' * 30, message, '-' * 30))


def elapsed_time(start):
    t = datetime.now().timestamp() - start
    unit = 'sec'
    if t > 60000:
        t, unit = t / 3600, 'hrs'
    elif t > 1000:
        t, unit = t / 60, 'min'
    return '{:.3f} {}'.format(t, unit)
