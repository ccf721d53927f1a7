"""
torch.autograd.Function wrappers over the C-ABI of librime_hip.so (include/rime_hip.h).

These are the only callers of the native library.  Every function requires CUDA (ROCm)
tensors and raises otherwise -- there is no CPU path in the product.

  fringe_sum   : vis = sum_pix psky * exp(+-2 pi i nu/c b.s)     rime_model.py:423-429 +
                                                                 telescope_model.py:310-358
  interp_gather: out[..., p] = sum_k w[p,k] m[..., inds[p,k]]    utils.py:815-861
  alm2pix      : out = Re(alm @ Ylm)                             sph_harm.py:1342-1372
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, RIME_F32, RIME_F64

TILE = 64        # pixel-axis padding granule required by the fringe kernels (rime::TP)

# bench.py sets this to a list to collect (kernel name, start event, end event, fringe elements)
# around every fringe-sum launch; None (default) records nothing.
PROFILE = None


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("bayeslim_amd ops need tensors on the GPU (got device '%s'); "
                               "there is no CPU implementation" % t.device)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _real_dtype(t):
    if t.dtype in (torch.float32, torch.complex64):
        return RIME_F32, torch.float32
    if t.dtype in (torch.float64, torch.complex128):
        return RIME_F64, torch.float64
    raise TypeError('unsupported dtype %s' % t.dtype)


def pad_to_tile(n):
    return ((int(n) + TILE - 1) // TILE) * TILE


class FringeGeometry:
    """
    Everything the fringe kernels need besides psky: baseline vectors, per-time pointing
    vectors, frequencies and the baseline -> beam-model-pair grouping.  Built once per RIME
    minibatch and reused by forward and backward.

    blvecs (Nbl, 3) [m]; sdir (Nt, 3, Pstride) unit vectors, zero-padded past each time's
    pixel count; freqs (Nf,) [Hz]; bl_mp: optional beam-model-pair index per baseline (Nmp pairs,
    mp_pairs = their (model1, model2) tuples: needed for the matrix-core path with several models);
    antpos (Nant, 3) + bl_ants (antenna-index pairs) enable the antenna-factored matrix-core kernels;
    group: antennas per block group of that path (default 128; 32 / 64 for rank-local tile shards).
    """
    def __init__(self, blvecs, sdir, freqs, bl_mp=None, Nmp=1, conj=False, npix=None,
                 antpos=None, bl_ants=None, mfma='auto', ant_like=None, mp_pairs=None, group=None):
        _require_cuda(blvecs, sdir)
        dev = blvecs.device
        self.blvecs = blvecs.detach().to(torch.float64).contiguous()
        self.sdir = sdir.detach().to(torch.float64).contiguous()
        f = torch.as_tensor(freqs, dtype=torch.float64)
        fh = f.detach().cpu().numpy()
        self.freqs = f.to(dev).contiguous()
        self.Nbl, self.Nt, self.Nf = self.blvecs.shape[0], self.sdir.shape[0], len(fh)
        self.Pstride = self.sdir.shape[2]
        assert self.sdir.shape[1] == 3 and self.Pstride % TILE == 0
        self.sign = -1 if conj else 1
        # algorithmic fringe elements per launch: Nbl x Nf x (valid pixels summed over times)
        nvalid = int(sum(npix)) if npix is not None else self.Nt * self.Pstride
        self.elements = self.Nbl * self.Nf * nvalid
        self.max_blen = float(torch.linalg.norm(self.blvecs, dim=1).max().item())
        # channel grid: 1 = exactly uniform (rotation recurrence), 2 = uniform up to tiny residuals
        # (float32-rounded linspace: recurrence + first-order correction), 0 = arbitrary
        self.f0 = float(fh[0])
        if len(fh) > 1:
            self.df = float((fh[-1] - fh[0]) / (len(fh) - 1))
            eps = np.abs(fh - (self.f0 + self.df * np.arange(len(fh)))).max()
            if eps <= 1e-9 * max(abs(self.df), 1.0):
                self.uniform = 1
            elif 2 * np.pi * eps * max(self.max_blen, 1.0) / 2.99792458e8 < 2e-3 and self.df != 0:
                self.uniform = 2
            else:
                self.uniform = 0
        else:
            self.uniform, self.df = 1, 0.0
        # antenna factorisation (matrix-core path): baselines given as antenna-index pairs
        self.ant = None
        if mfma is not False and ant_like is not None and ant_like.ant is not None and ant_like.Nbl == self.Nbl:
            # same baseline set as an existing geometry (another time minibatch): share its pair tables
            per16 = self.Nt * self.Nf * (self.Pstride // 16) * 32768
            self.ant = dict(ant_like.ant, mfma_flops_fwd=per16 * ant_like.ant['mfma_fwd'],
                            mfma_flops_bwd=per16 * ant_like.ant['mfma_bwd'])
            if self.Nt > 65535:
                self.ant = None
        elif antpos is not None and bl_ants is not None and mfma in ('auto', True):
            self._setup_antenna_path(antpos, bl_ants, force=(mfma is True), bl_mp=bl_mp if int(Nmp) > 1 else None,
                                     mp_pairs=mp_pairs, group=group or MFMA_GROUP)
        # model-pair grouping
        self.Nmp = int(Nmp)
        if bl_mp is None or self.Nmp == 1:
            self.Nmp = 1
            self.mp_offsets = (ctypes.c_int * 2)(0, self.Nbl)
            self.bl_order = None
        else:
            bl_mp = np.asarray(bl_mp, dtype=np.int64)
            order = np.argsort(bl_mp, kind='stable')
            counts = np.bincount(bl_mp, minlength=self.Nmp)
            offs = np.concatenate([[0], np.cumsum(counts)])
            self.mp_offsets = (ctypes.c_int * (self.Nmp + 1))(*[int(o) for o in offs])
            self.bl_order = torch.as_tensor(order, dtype=torch.int32, device=dev)


def _env_int(name, default):
    """C atoi() of an environment variable (leading integer, else 0), as csrc/ reads its switches"""
    v = os.environ.get(name)
    if v is None:
        return default
    import re
    m = re.match(r'\s*([+-]?\d+)', v)
    return int(m.group(1)) if m else 0


MFMA_MIN_ANTS = int(os.environ.get('RIME_MFMA_MIN_ANTS', '16'))   # 'auto' threshold (see _setup_antenna_path)
MFMA_GROUP = 128          # antennas per group of the matrix-core path (4 x 4 tiles of 32)
MFMA_MAX_ANTS = 2048      # table memory only: 136 blocks x 128 KB at 2048 antennas
# complex psky, forward: diagonal blocks whose baselines all have one orientation run as
# triangular self-cross blocks in ONE pass (RIME_SELF_BLOCKS=0: the two real-plane passes of the diagonal kernel)
SELF_BLOCKS = os.environ.get('RIME_SELF_BLOCKS', '1') != '0'
# arrays of 33..48 antennas: the forward kernel with the second row tile's re / im planes packed into one operand
# (csrc/fringe_mfma.hip, fringe_ant_fwd_packed_kernel); RIME_FWD_PACKED=0 keeps the generic two-tile kernel (A/B)
FWD_PACKED = _env_int('RIME_FWD_PACKED', 1) != 0         # parsed as the library parses it (atoi)


def _group_capacity(n, group):
    """rows a group of n antennas occupies in a cross block: 32, 64 or 128 (the kernels' shapes)"""
    if group <= 32 or n <= 32:
        return 32
    return 64 if (group <= 64 or n <= 64) else 128


def _antenna_blocks(bl_ants, Nant, bl_mp=None, ant_model=None, group=MFMA_GROUP):
    """
    Block decomposition of the pair matrix for the matrix-core path (see include/rime_hip.h).
    Antennas are ordered by (beam model, index) and cut into groups of <= `group` antennas that share
    a beam model; every (group pair, beam-model pair) that holds a baseline becomes one block with its
    own psky plane `mp`:
      diagonal block: one group against itself (upper-triangular 32 x 32 tiles),
      cross block   : group I x group J, rows padded to a supported shape (32, 32), (32, 64), (64, 64)
                      or (128, 128) antenna rows.
    Returns a list of dicts {ants_i, ants_j | None, rows_i, rows_j, mp, direct, conj, cpass}, tables
    int32 [128, 128] of baseline slots indexed by LOCAL antenna indices, or None when a pair occurs
    twice (not representable).  cpass: +1 the block holds direct entries only, -1 conj entries only,
    0 both (a complex psky then takes one pass per real plane).
    """
    if ant_model is None:
        ant_model = [0] * Nant
    order = sorted(range(Nant), key=lambda a: (ant_model[a], a))
    groups, gid, lid = [], {}, {}
    for a in order:
        if not groups or ant_model[groups[-1][0]] != ant_model[a] or len(groups[-1]) >= group:
            groups.append([])
        gid[a], lid[a] = len(groups) - 1, len(groups[-1])
        groups[-1].append(a)
    tabs = {}

    def get(key):
        if key not in tabs:
            tabs[key] = (np.full((MFMA_GROUP, MFMA_GROUP), -1, dtype=np.int32),
                         np.full((MFMA_GROUP, MFMA_GROUP), -1, dtype=np.int32))
        return tabs[key]

    for b, (a1, a2) in enumerate(bl_ants):
        mp = 0 if bl_mp is None else int(bl_mp[b])
        g1, l1, g2, l2 = gid[a1], lid[a1], gid[a2], lid[a2]
        if g1 == g2:
            direct, conj = get((g1, g1, mp))
            if l1 // 32 <= l2 // 32:
                tab, i, j = direct, l1, l2
            else:
                tab, i, j = conj, l2, l1
        elif g1 < g2:
            tab, i, j = get((g1, g2, mp))[0], l1, l2           # baseline I_i -> J_j: V[i, j]
        else:
            tab, i, j = get((g2, g1, mp))[1], l2, l1           # baseline J_j -> I_i: conj(V[i, j])
        if tab[i, j] >= 0:
            return None
        tab[i, j] = b
    blocks = []

    def compact(tab, rows, cols):
        out = np.full((MFMA_GROUP, MFMA_GROUP), -1, dtype=np.int32)
        out[:len(rows), :len(cols)] = tab[np.ix_(rows, cols)]
        return out

    for (gi, gj, mp) in sorted(tabs):
        direct, conj = tabs[(gi, gj, mp)]
        nd, nc = int((direct >= 0).sum()), int((conj >= 0).sum())
        # antennas of the groups that hold no baseline of this block are dropped (a rank-local shard of a
        # block, a sparse pair set): fewer generated rows, possibly a smaller kernel shape
        used = (direct >= 0) | (conj >= 0)
        if gi == gj:
            keep = np.nonzero(used.any(0) | used.any(1))[0]
            # tile membership (index // 32) decides direct vs conj inside a diagonal block: only drop whole
            # trailing / unused 32-antenna tiles' worth when the order of the rest is unchanged
            if len(keep) < len(groups[gi]) and np.array_equal(keep // 32, np.arange(len(keep)) // 32):
                direct, conj = compact(direct, keep, keep), compact(conj, keep, keep)
                ants_i = [groups[gi][k] for k in keep]
            else:
                ants_i = groups[gi]
            blk = dict(ants_i=ants_i, ants_j=None, rows_i=len(ants_i), rows_j=0, mp=mp,
                       cpass=(1 if nc == 0 else (-1 if nd == 0 else 0)))
        else:
            ri, cj_ = np.nonzero(used.any(1))[0], np.nonzero(used.any(0))[0]
            direct, conj = compact(direct, ri, cj_), compact(conj, ri, cj_)
            blk = dict(ants_i=[groups[gi][k] for k in ri], ants_j=None, rows_i=len(ri), rows_j=0, mp=mp,
                       cpass=(1 if nc == 0 else (-1 if nd == 0 else 0)))
            gi_ants, gj_ants = blk['ants_i'], [groups[gj][k] for k in cj_]
        if gi != gj:
            ci, cj = _group_capacity(len(gi_ants), group), _group_capacity(len(gj_ants), group)
            if (ci, cj) not in ((32, 32), (32, 64), (64, 32), (64, 64), (128, 128)):
                ci = cj = 128                                # unequal groups: padded to the full shape
            ai, aj = gi_ants, gj_ants
            if ci > cj:
                # the kernels take the smaller group as I: swap the groups (V[j, i] = conj(V[i, j]))
                ai, aj, ci, cj = aj, ai, cj, ci
                direct, conj = conj.T.copy(), direct.T.copy()
                blk['cpass'] = -blk['cpass']
            blk.update(ants_i=ai, ants_j=aj, rows_i=ci, rows_j=cj)
        blk['direct'], blk['conj'] = direct, conj
        blocks.append(blk)
    return blocks


# MIRROR PAIRS (round 5; csrc/fringe_mfma.hip, "MIRROR PAIRS"): antennas with r' - c = -(r - c) have conjugate phasors.
MIRROR = _env_int('RIME_MIRROR', 1) != 0        # RIME_MIRROR=0: no search (every row evaluated, rounds 1-4)
MIRROR_TOL = 1e-9                               # [m] mismatch allowed in r + r' = 2 c (7e-10 turn of phase at 200 MHz)


def _mirror_pairs(P, tol=MIRROR_TOL):
    """
    Point symmetry of a set of antenna positions P (n, 3): the centre c shared by the largest number of pairs
    (r_a + r_b = 2 c within `tol`, all three coordinates) and a maximal pairing about it.  Returns (c, pairs, singles) --
    pairs of row indices (a, b), singles the rows without a partner (an antenna AT the centre is its own mirror: a single) --
    or None when fewer than two pairs exist.  O(n^2) on the host, n <= 128 per block.
    """
    P = np.asarray(P, dtype=np.float64)
    n = len(P)
    if n < 4:
        return None
    ia, ib = np.triu_indices(n, 1)
    sums = P[ia] + P[ib]
    # vote for 2 c on a 1-micrometre grid (a symmetric set of m antennas puts ~m / 2 of its n (n - 1) / 2 pair sums in one cell;
    # twice, with the grid shifted by half a cell, so that sums sitting on a cell edge are not split)
    best = None
    for shift in (0.0, 0.5):
        q = np.floor(sums * 1e6 + shift).astype(np.int64)
        cells, inv, cnt = np.unique(q, axis=0, return_inverse=True, return_counts=True)
        k = int(np.argmax(cnt))
        if best is None or cnt[k] > best[0]:
            best = (int(cnt[k]), sums[np.asarray(inv).reshape(-1) == k].mean(axis=0))
    if best[0] < 2:
        return None
    c2 = best[1]
    D = np.abs(P[:, None, :] + P[None, :, :] - c2).max(-1) <= tol
    partner = -np.ones(n, dtype=np.int64)
    for a in range(n):
        if partner[a] >= 0:
            continue
        for b in np.nonzero(D[a])[0]:
            if b != a and partner[b] < 0:
                partner[a], partner[b] = b, a
                break
    pairs = [(a, int(partner[a])) for a in range(n) if partner[a] > a]
    singles = [a for a in range(n) if partner[a] < 0]
    if len(pairs) < 2:
        return None
    # the centre the pairs actually share (mean of their sums: the kernels use E' = conj(E) EXACTLY for a pair)
    c = np.mean([P[a] + P[b] for a, b in pairs], axis=0) / 2
    if max(np.abs(P[a] + P[b] - 2 * c).max() for a, b in pairs) > tol:
        return None
    return c, pairs, singles


def _mirror_order(P, packed=None):
    """
    Row order of a diagonal block that puts mirror pairs into the octet pairs of 16-row groups: rows 16 g + i (i < 8) and
    16 g + 8 + i of the groups g < Gm hold the two antennas of a pair (or a single and an empty row); the other antennas
    fill plain groups behind them.  Returns (rows, mask, centre): rows[r] = index into P or -1 (empty row), mask = bit per
    mirror group -- or None when no order with at least one mirror group fits into the block's row capacity
    32 ceil(n / 32) (the kernel shape of the plain order must not grow).  `packed`: 33..48 antennas on the packed forward
    kernel -- 48 rows, mirror groups in the first row tile only.
    """
    found = _mirror_pairs(P)
    if found is None:
        return None
    c, pairs, singles = found
    n = len(P)
    packed = (32 < n <= 48 and FWD_PACKED) if packed is None else packed
    G = 3 if packed else (32 * ((n + 31) // 32)) // 16              # 16-row groups available
    Gcap = 2 if packed else G                                       # groups that may be mirror groups
    for Gm in range(min(Gcap, (len(pairs) + 7) // 8), 0, -1):
        slots = 8 * Gm
        pin = min(len(pairs), slots)
        sin = min(len(singles), slots - pin)
        if n - 2 * pin - sin <= 16 * (G - Gm):
            break
    else:
        return None
    rows = [-1] * (16 * G)
    for k, (a, b) in enumerate(pairs[:pin]):
        rows[16 * (k // 8) + k % 8], rows[16 * (k // 8) + 8 + k % 8] = a, b
    for k, a in enumerate(singles[:sin], start=pin):
        rows[16 * (k // 8) + k % 8] = a                             # its mirror row stays empty
    rest = [x for ab in pairs[pin:] for x in ab] + list(singles[sin:])
    rows[16 * Gm:16 * Gm + len(rest)] = rest
    while rows and rows[-1] < 0:
        rows.pop()
    return rows, (1 << Gm) - 1, c


def _mirror_block(blk, P, dev):
    """the mirrored form of a built diagonal block (see _mirror_order): positions measured from the centre of symmetry in
    the new row order, pair tables rebuilt for it, `mirror` mask; None when the block has no usable symmetry"""
    n = int(blk['nrows'])
    if blk['cross'] != 0 or n < 4:
        return None
    found = _mirror_order(P)
    if found is None:
        return None
    rows, mask, c = found
    newrow = -np.ones(n, dtype=np.int64)
    for r, a in enumerate(rows):
        if a >= 0:
            newrow[a] = r
    assert (newrow >= 0).all() and len(rows) <= 32 * ((n + 31) // 32)
    pos = np.zeros((len(rows), 3))
    for r, a in enumerate(rows):
        if a >= 0:
            pos[r] = P[a] - c
    direct = np.full((MFMA_GROUP, MFMA_GROUP), -1, dtype=np.int32)
    conj = np.full((MFMA_GROUP, MFMA_GROUP), -1, dtype=np.int32)
    od, oc = blk['direct'].reshape(MFMA_GROUP, MFMA_GROUP).cpu().numpy(), blk['conj'].reshape(MFMA_GROUP, MFMA_GROUP).cpu().numpy()
    # direct[i, j] = b: baseline b runs from antenna i to antenna j; conj[i, j] = b: from j to i (_antenna_blocks)
    for tab, swap in ((od, False), (oc, True)):
        for i, j in zip(*np.nonzero(tab >= 0)):
            a1, a2 = (j, i) if swap else (i, j)
            r1, r2 = newrow[a1], newrow[a2]
            if r1 // 32 <= r2 // 32:
                direct[r1, r2] = tab[i, j]
            else:
                conj[r2, r1] = tab[i, j]
    nd, nc = int((direct >= 0).sum()), int((conj >= 0).sum())
    assert nd + nc == int((od >= 0).sum()) + int((oc >= 0).sum())
    return dict(blk, pos=torch.as_tensor(pos, device=dev).contiguous(), nrows=len(rows), mirror=int(mask), rows=list(rows),
                cpass=(1 if nc == 0 else (-1 if nd == 0 else 0)), fwd_cpass=0, self_pos=None, mf_self=0,
                direct=torch.as_tensor(direct.reshape(-1), device=dev), conj=torch.as_tensor(conj.reshape(-1), device=dev))


# CONJUGATE-PAIR FORM (round 5; csrc/fringe_mfma.hip, "CONJUGATE-PAIR FORM"): a diagonal block of a point-symmetric array
# contracted from the phasors of one antenna of every mirror pair.
PAIR = _env_int('RIME_PAIR', 1) != 0            # RIME_PAIR=0: such blocks keep the mirror-pair kernels (A/B measurements)
PAIR_ROWS = 64                                  # rows of the pair kernels (firsts + antennas without a partner)
PAIR_CPLX = _env_int('RIME_PAIR_CPLX', 1) != 0  # complex psky on pair blocks, one pass per real plane (0: the plain blocks; A/B)


def _pair_layout(P, rows=PAIR_ROWS, hub_ok=True):
    """
    Rows of the conjugate-pair form for antenna positions P (n, 3): (firsts, partner, hub, centre) -- firsts[k] the antenna
    of row k, partner[k] its mirror antenna or -1, hub the antenna AT the centre that is served outside the rows (or None)
    -- or None when the set has no point symmetry or does not fit: at most `rows` rows, plus (hub_ok) the hub when the rows
    are full.
    """
    found = _mirror_pairs(P)
    if found is None:
        return None
    c, pairs, singles = found
    hub = None
    if len(pairs) + len(singles) > rows:
        at_c = [a for a in singles if np.abs(np.asarray(P[a]) - c).max() <= MIRROR_TOL]
        if not hub_ok or not at_c or len(pairs) + len(singles) - 1 > rows:
            return None
        hub = at_c[0]
        singles = [a for a in singles if a != hub]
    firsts = [a for a, _ in pairs] + list(singles)
    partner = [b for _, b in pairs] + [-1] * len(singles)
    return firsts, partner, hub, c


def _pair_block(blk, P, dev):
    """the conjugate-pair form of a built diagonal block (include/rime_hip.h, rime_fringe_pair_fwd_block): positions of the
    rows from the centre of symmetry, the pair tables of the virtual 128-row block (row k: firsts[k], row 64 + k: its
    mirror), the hub's slot table; None when the block does not qualify.  More than 64 antennas: up to 64 rows + the hub (two
    row tiles: 26 MFMAs per K step against 63 / 100 of the three- / four-tile kernels); 33..64 antennas: up to 32 rows (one row
    tile: 7 against 16 / 26); up to 32 antennas keep the one-tile kernels (the same 7 MFMAs; their generation is already
    halved by the mirror pairs)."""
    n = int(blk['nrows'])
    if blk['cross'] != 0 or n <= 32:
        return None
    lay = _pair_layout(P) if n > 64 else _pair_layout(P, rows=32, hub_ok=False)
    if lay is None:
        return None
    firsts, partner, hub, c = lay
    vrow = -np.ones(n, dtype=np.int64)
    for k, (a, b) in enumerate(zip(firsts, partner)):
        vrow[a] = k
        if b >= 0:
            vrow[b] = 64 + k
    direct = np.full((MFMA_GROUP, MFMA_GROUP), -1, dtype=np.int32)
    conj = np.full((MFMA_GROUP, MFMA_GROUP), -1, dtype=np.int32)
    centre = np.full((2, MFMA_GROUP), -1, dtype=np.int32)
    od, oc = blk['direct'].reshape(MFMA_GROUP, MFMA_GROUP).cpu().numpy(), blk['conj'].reshape(MFMA_GROUP, MFMA_GROUP).cpu().numpy()
    # direct[i, j] = b: baseline b runs from antenna i to antenna j; conj[i, j] = b: from j to i (_antenna_blocks)
    for tab, swap in ((od, False), (oc, True)):
        for i, j in zip(*np.nonzero(tab >= 0)):
            a1, a2 = (j, i) if swap else (i, j)
            if a1 == hub and a2 == hub:
                return None                                      # the hub's autocorrelation: not a column sum of the image
            if a1 == hub:
                centre[0, vrow[a2]] = tab[i, j]
            elif a2 == hub:
                centre[1, vrow[a1]] = tab[i, j]
            else:
                r1, r2 = vrow[a1], vrow[a2]
                if r1 // 32 <= r2 // 32:
                    direct[r1, r2] = tab[i, j]
                else:
                    conj[r2, r1] = tab[i, j]
    nslots = int((direct >= 0).sum()) + int((conj >= 0).sum()) + int((centre >= 0).sum())
    assert nslots == int((od >= 0).sum()) + int((oc >= 0).sum())
    pos = np.asarray(P)[firsts] - c
    # a coplanar array measured from a centre in its plane: z = 0 within the tolerance of the pairing itself -> exactly 0 and
    # the `flat` licence (the kernels skip that term of the phase)
    flat = int(np.abs(pos[:, 2]).max() <= MIRROR_TOL)
    if flat:
        pos[:, 2] = 0.0
    return dict(blk, pos=torch.as_tensor(pos, device=dev).contiguous(), nrows=len(firsts), mirror=0, pair=1,
                firsts=list(firsts), partner=list(partner), hub=hub, flat=flat,
                centre=None if hub is None else torch.as_tensor(centre.reshape(-1), device=dev),
                cpass=0, fwd_cpass=0, self_pos=None, mf_self=0, mf_fwd=26 if len(firsts) > 32 or hub is not None else 7,
                mf_bwd_real=30 if len(firsts) > 32 else 9,
                direct=torch.as_tensor(direct.reshape(-1), device=dev), conj=torch.as_tensor(conj.reshape(-1), device=dev))


def _dense_strides(t):
    """element strides (time, model pair, pol product, channel) of a (Nt,Nmp,Npp,Nf,P) tensor whose
    pixel axis is contiguous, or None when the tensor cannot be passed as is"""
    if t.stride(-1) != 1:
        return None
    st = t.stride()
    P = t.shape[-1]
    if t.shape[3] > 1 and st[3] < P:
        return None
    if any(s <= 0 and n > 1 for s, n in zip(st[:4], t.shape[:4])):
        return None
    return (ctypes.c_longlong * 4)(*[int(max(s, 1)) for s in st[:4]])


def _pow2_scale(amax):
    """power of two s with amax * s in [2^13, 2^14]; 1 where amax == 0 (exact to apply and undo)"""
    safe = torch.where(amax > 0, amax, torch.ones_like(amax))
    e = torch.floor(torch.log2(16384.0 / safe)).clamp(-100.0, 100.0)      # denormal-sized rows: no inf scale
    return torch.where(amax > 0, torch.exp2(e), torch.ones_like(amax))


def _fringe_ant_call(geom, backward, inp, out, strides, Npp, cplx):
    """
    antenna-factored kernels: one launch per block of the pair matrix (ops._antenna_blocks), each on
    the psky plane of its beam-model pair.  Real psky: one pass per polarisation product.  Complex
    psky (V is linear in psky: V[ar + i ai] = V[ar] + i V[ai]; d/d(ai) = Re(conj(F) (-i g))): blocks
    whose entries all have one orientation take ONE complex pass (forward: cross blocks; backward:
    every block, the imaginary plane is a second lane-local contraction of the same products); the
    others take one pass per real plane.  Returns the MFMA flops executed.
    """
    a = geom.ant
    # real psky: mirror / pair forms where the array has them; complex psky: the plain blocks (single-pass forms where a
    # block's baselines allow), except where a block has the conjugate-pair form -- V is linear in psky, so that block takes
    # one pair pass per real plane (2 x 26 MFMAs per K step against 120 of the one-pass self block of 128 antennas)
    blocks = a.get('blocks_cplx', a['blocks']) if cplx else a.get('blocks_real', a['blocks'])
    tp_mask = a.get('two_pass_mask_cplx', a['two_pass_mask']) if cplx else a['two_pass_mask']
    m = 2 if cplx else 1                                     # floats per psky element
    st_t, st_mp, st_pp, st_f = (int(strides[k]) * m for k in range(4))
    Nbl, Nt, Nf, Nmp = geom.Nbl, geom.Nt, geom.Nf, geom.Nmp
    dev = inp.device
    geo = (_ptr(geom.sdir), _ptr(geom.freqs))
    shape = (Nbl, Nt, Nf, geom.Pstride, st_t, st_f, m, geom.sign)
    per16 = Nt * Nf * (geom.Pstride // 16) * 32768
    flops = 0
    if not backward:
        # inp: psky (Nt, Nmp, Npp, Nf, Ps[, 2]) float32 view; out: vis (Npp, Nbl, Nt, Nf, 2) float32
        if cplx and inp.stride(-1) == 1 and inp.stride(-2) == 2 and all(inp.stride(k) % 2 == 0 for k in range(4)):
            # interleaved complex rows: scale from max(|re|, |im|) and the minimum of each plane in ONE pass (round 4; torch's
            # abs + amax + amin passes were 1.1 ms of a C5 rank step)
            scale = torch.empty((Nmp, Npp, Nt, Nf), dtype=torch.float32, device=dev)
            rowmin = [torch.empty_like(scale), torch.empty_like(scale)]
            check(lib.rime_fringe_row_scale_cplx(_ptr(inp), Nmp, Npp, Nt, Nf, inp.stride(1) // 2, inp.stride(2) // 2,
                                                 inp.stride(0) // 2, inp.stride(3) // 2, inp.shape[-2], _ptr(scale),
                                                 _ptr(rowmin[0]), _ptr(rowmin[1]), _stream()), 'rime_fringe_row_scale_cplx')
        elif cplx:
            amax = inp.abs().amax(dim=(-1, -2))                                    # (Nt, Nmp, Npp, Nf)
            lo = inp.amin(dim=-2)                                                  # (Nt, Nmp, Npp, Nf, 2): per plane
            rowmin = [lo[..., c].permute(1, 2, 0, 3).contiguous() for c in range(2)]
            scale = _pow2_scale(amax.permute(1, 2, 0, 3)).contiguous()             # (Nmp, Npp, Nt, Nf)
        elif inp.stride(-1) == 1:
            # one launch: power-of-two scale and minimum of every (mp, pp, t, f) row, through the view's strides
            scale = torch.empty((Nmp, Npp, Nt, Nf), dtype=torch.float32, device=dev)
            rowmin = [torch.empty_like(scale)]
            check(lib.rime_fringe_row_scale(_ptr(inp), Nmp, Npp, Nt, Nf, inp.stride(1), inp.stride(2), inp.stride(0),
                                            inp.stride(3), inp.shape[-1], _ptr(scale), _ptr(rowmin[0]), _stream()),
                  'rime_fringe_row_scale')
        else:
            lo, hi = torch.aminmax(inp, dim=-1)                                    # one pass, no |psky| temporary
            amax = torch.maximum(hi, -lo)
            rowmin = [lo.permute(1, 2, 0, 3).contiguous()]
            scale = _pow2_scale(amax.permute(1, 2, 0, 3)).contiguous()             # (Nmp, Npp, Nt, Nf)
        nbytes = lib.rime_fringe_ant_workspace(Nbl, Nt, Nf, geom.Pstride)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        two_pass = [blk for blk in blocks if cplx and blk['fwd_cpass'] == 0]
        tmp = torch.empty((Nbl, Nt, Nf, 2), dtype=torch.float32, device=dev) if two_pass else None

        def launch(blk, pp, c, cflag):
            mp = blk['mp']
            src = ctypes.c_void_p(inp.data_ptr() + 4 * (mp * st_mp + pp * st_pp + c))
            if cflag != 0 and blk['cross'] == 0:             # diagonal block, complex single pass: self block
                n = int(blk['self_pos'].shape[0])
                rc = lib.rime_fringe_ant_fwd_block(_ptr(blk['self_pos']), n, n, 0, *geo, src, _ptr(scale[mp, pp]),
                                                   _ptr(rowmin[c][mp, pp]), _ptr(blk['direct']), _ptr(blk['conj']),
                                                   *shape, cflag, _ptr(ws), ws.numel(), _stream())
                check(rc, 'rime_fringe_ant_fwd_block')
                return blk['mf_self']
            if blk.get('pair'):                              # conjugate-pair form: one real plane per call (c picks it)
                rc = lib.rime_fringe_pair_fwd_block(_ptr(blk['pos']), blk['nrows'], _ptr(blk['centre']), blk['flat'], *geo, src,
                                                    _ptr(scale[mp, pp]), _ptr(rowmin[c][mp, pp]),
                                                    _ptr(blk['direct']), _ptr(blk['conj']),
                                                    *shape, _ptr(ws), ws.numel(), _stream())
                check(rc, 'rime_fringe_pair_fwd_block')
                return blk['mf_fwd']
            rc = lib.rime_fringe_ant_fwd_block(_ptr(blk['pos']), blk['nrows'], blk['cross'], blk['mirror'], *geo, src,
                                               _ptr(scale[mp, pp]), _ptr(rowmin[c][mp, pp]),
                                               _ptr(blk['direct']), _ptr(blk['conj']),
                                               *shape, cflag, _ptr(ws), ws.numel(), _stream())
            check(rc, 'rime_fringe_ant_fwd_block')
            return blk['mf_fwd']

        for pp in range(Npp):
            for blk in blocks:
                single = cplx and blk['fwd_cpass'] != 0
                flops += launch(blk, pp, 0, blk['fwd_cpass'] if single else 0)
            check(lib.rime_fringe_ant_fwd_finish(_ptr(ws), ws.numel(), _ptr(out[pp]), Nbl, Nt, Nf, geom.Pstride,
                                                 _stream()), 'rime_fringe_ant_fwd_finish')
            if two_pass:                                     # V += i V[ai] on the baselines of those blocks
                for blk in two_pass:
                    flops += launch(blk, pp, 1, 0)
                check(lib.rime_fringe_ant_fwd_finish(_ptr(ws), ws.numel(), _ptr(tmp), Nbl, Nt, Nf, geom.Pstride,
                                                     _stream()), 'rime_fringe_ant_fwd_finish')
                if len(two_pass) < len(blocks):
                    tmp *= tp_mask                           # slots of single-pass blocks hold stale values
                out[pp][..., 0] -= tmp[..., 1]
                out[pp][..., 1] += tmp[..., 0]
    else:
        # inp: gvis as real (Npp, Nbl, Nt, Nf, 2); out: gpsky float32 view with psky's strides
        nbytes = lib.rime_fringe_ant_bwd_workspace(Nbl, Nt, Nf)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        gvt = ws[:Nt * Nf * 2 * Nbl * 4].view(torch.float32).view(Nt * Nf, 2 * Nbl)    # what bwd_prepare writes
        two_pass = [blk for blk in blocks if cplx and blk['cpass'] == 0]
        for pp in range(Npp):
            written = set()
            for c in range(2 if two_pass else 1):
                g = inp[pp] if c == 0 else torch.stack([inp[pp][..., 1], -inp[pp][..., 0]], dim=-1).contiguous()
                check(lib.rime_fringe_ant_bwd_prepare(_ptr(g), Nbl, Nt, Nf, _ptr(ws), ws.numel(), _stream()),
                      'rime_fringe_ant_bwd_prepare')
                if c == 0:
                    # max |gvis| per (t, f) from the transposed copy: a contiguous row reduction (the same
                    # reduction over the (Nbl, Nt, Nf, 2) layout is strided and 4x slower); -i g has the same maxima
                    scale_pp = torch.empty(Nt * Nf, dtype=torch.float32, device=dev)
                    check(lib.rime_fringe_row_scale(_ptr(gvt), 1, 1, 1, Nt * Nf, 0, 0, 0, 2 * Nbl, 2 * Nbl, _ptr(scale_pp),
                                                    None, _stream()), 'rime_fringe_row_scale')
                # single-pass blocks first: they initialise BOTH planes of their psky slice, the two-pass
                # blocks then accumulate plane by plane
                todo = sorted(blocks, key=lambda b: not (cplx and b['cpass'] != 0)) if c == 0 else two_pass
                for blk in todo:
                    single = cplx and blk['cpass'] != 0
                    mp = blk['mp']
                    planes = [(mp, 0), (mp, 1)] if single else [(mp, c)]
                    acc = int(planes[0] in written)
                    assert all((pl in written) == bool(acc) for pl in planes)
                    written.update(planes)
                    dst = ctypes.c_void_p(out.data_ptr() + 4 * (mp * st_mp + pp * st_pp + (0 if single else c)))
                    if blk.get('pair'):
                        rc = lib.rime_fringe_pair_bwd_block(_ptr(blk['pos']), blk['nrows'], _ptr(blk['centre']), blk['flat'], *geo,
                                                            _ptr(scale_pp), _ptr(blk['direct']), _ptr(blk['conj']),
                                                            *shape, acc, dst, _ptr(ws), ws.numel(), _stream())
                        check(rc, 'rime_fringe_pair_bwd_block')
                        flops += blk['mf_bwd_real']
                        continue
                    rc = lib.rime_fringe_ant_bwd_block(_ptr(blk['pos']), blk['nrows'], blk['cross'], blk['mirror'], *geo,
                                                       _ptr(scale_pp), _ptr(blk['direct']), _ptr(blk['conj']),
                                                       *shape, blk['cpass'] if single else 0, acc, dst,
                                                       _ptr(ws), ws.numel(), _stream())
                    check(rc, 'rime_fringe_ant_bwd_block')
                    flops += blk['mf_bwd'] if single else blk['mf_bwd_real']
            assert len(written) == Nmp * m, 'every psky plane must be written by a block'
    return flops * per16


def _setup_antenna_path(self, antpos, bl_ants, force=False, bl_mp=None, mp_pairs=None, group=MFMA_GROUP):
    """enable the antenna-factored matrix-core kernels when the baseline set suits them"""
    Nant = int(antpos.shape[0])
    bl_ants = [(int(a), int(b)) for a, b in bl_ants]
    if Nant > MFMA_MAX_ANTS or len(bl_ants) != self.Nbl or self.Nt > 65535:      # (reduce kernels: t on grid.z)
        return
    # worth it when the array needs at least two 32-antenna tiles and most pairs are requested
    # (measured against the baseline-formulation kernels: 128 antennas / 8128 baselines 4.8x (fwd)
    # and 5.8x (bwd) faster; 37 antennas / 666 baselines 1.1x and 1.6x faster; 19 antennas 1.7x and
    # 1.1x slower)
    if not force and (Nant < MFMA_MIN_ANTS or self.Nbl < Nant * Nant // 8):
        return
    # several beam models: every antenna must carry ONE model (beam_model.py:303-327 pairs models per baseline)
    ant_model = None
    if bl_mp is not None and mp_pairs is not None and len(mp_pairs) > 1:
        ant_model = [None] * Nant
        for (a1, a2), mp in zip(bl_ants, bl_mp):
            for ant, mdl in ((a1, mp_pairs[mp][0]), (a2, mp_pairs[mp][1])):
                if ant_model[ant] is None:
                    ant_model[ant] = mdl
                elif ant_model[ant] != mdl:
                    return
        ant_model = [0 if mdl is None else mdl for mdl in ant_model]
    elif bl_mp is not None and any(int(x) != 0 for x in bl_mp):          # (bl_mp is only passed when Nmp > 1)
        # without a model table the blocks read psky plane 0: several planes, or one plane that is not plane 0,
        # stay on the baseline-formulation kernels
        return
    raw = _antenna_blocks(bl_ants, Nant, bl_mp if ant_model is not None else None, ant_model, group)
    if raw is None:
        return
    # the factorisation must reproduce the baseline vectors it replaces
    pos = antpos.detach().to(torch.float64).to(self.blvecs.device).contiguous()
    i1 = torch.as_tensor([a for a, _ in bl_ants], device=pos.device)
    i2 = torch.as_tensor([b for _, b in bl_ants], device=pos.device)
    if not torch.allclose(pos[i2] - pos[i1], self.blvecs, rtol=0, atol=1e-9):
        return
    dev = self.blvecs.device
    blocks, mfma_fwd, mfma_bwd = [], 0, 0
    two_pass_mask = torch.zeros(self.Nbl, 1, 1, 1, dtype=torch.float32, device=dev)
    for blk in raw:
        pi = pos[torch.as_tensor(blk['ants_i'], device=dev)]
        if blk['ants_j'] is None:
            rows, TA = pi.contiguous(), (pi.shape[0] + 31) // 32
            mf_fwd = 12 * (TA * (TA - 1) // 2) + 7 * TA      # diagonal tiles: symmetric products folded
            if 32 < pi.shape[0] <= 48 and FWD_PACKED:
                mf_fwd = 16                                  # packed second row tile (round 4): 7 + 6 + 3 MFMAs per K step
            mf_bwd = 12 * (TA * (TA + 1) // 2)
            mf_bwd_real = 12 * (TA * (TA - 1) // 2) + 9 * TA   # real psky: symmetric form on the diagonal tiles (round 3)
            cross, fwd_cpass = 0, 0                          # forward diagonal blocks: one real plane per call ...
            self_pos, mf_self = None, 0
            if SELF_BLOCKS and blk['cpass'] != 0:
                # ... unless psky is complex and every baseline has one orientation: then the block runs as the
                # triangular cross block of the group with itself, ONE complex pass (rime_fringe_ant_fwd_block
                # with cross == Nrows)
                self_pos = torch.zeros(TA * 32, 3, dtype=torch.float64, device=dev)
                self_pos[:pi.shape[0]] = pi
                mf_self = 12 * (TA * (TA + 1) // 2)
                fwd_cpass = blk['cpass']
        else:
            self_pos, mf_self = None, 0
            pj = pos[torch.as_tensor(blk['ants_j'], device=dev)]
            rows = torch.zeros(blk['rows_i'] + blk['rows_j'], 3, dtype=torch.float64, device=dev)
            rows[:pi.shape[0]] = pi
            rows[blk['rows_i']:blk['rows_i'] + pj.shape[0]] = pj
            mf_fwd = mf_bwd = mf_bwd_real = 12 * (blk['rows_i'] // 32) * (blk['rows_j'] // 32)
            cross, fwd_cpass = blk['rows_i'], blk['cpass']
        if fwd_cpass == 0:
            slots = np.concatenate([blk['direct'][blk['direct'] >= 0], blk['conj'][blk['conj'] >= 0]])
            two_pass_mask[torch.as_tensor(slots, dtype=torch.int64, device=dev)] = 1.0
        blocks.append(dict(pos=rows, nrows=int(rows.shape[0]), cross=int(cross), mp=blk['mp'], mirror=0,
                           cpass=blk['cpass'], fwd_cpass=fwd_cpass, mf_fwd=mf_fwd, mf_bwd=mf_bwd, mf_bwd_real=mf_bwd_real,
                           self_pos=self_pos, mf_self=mf_self,
                           direct=torch.as_tensor(blk['direct'].reshape(-1), device=dev),
                           conj=torch.as_tensor(blk['conj'].reshape(-1), device=dev)))
        mfma_fwd += mf_fwd
        mfma_bwd += mf_bwd_real
    # executed matrix-core work per pass: per 16 pixels, 12 MFMAs of 2*32*32*16 flop on each 32x32
    # antenna tile of every block (3 hi/lo products x 4 real products); the forward runs 7 on the
    # diagonal tiles of a diagonal block
    per16 = self.Nt * self.Nf * (self.Pstride // 16) * 32768
    self.ant = dict(blocks=blocks, Nant=Nant, mfma_fwd=mfma_fwd, mfma_bwd=mfma_bwd, two_pass_mask=two_pass_mask,
                    mfma_flops_fwd=per16 * mfma_fwd, mfma_flops_bwd=per16 * mfma_bwd,
                    multi_model=ant_model is not None)
    # arrays with point symmetry: the REAL-psky passes run on mirrored forms of the diagonal blocks (conjugate phasors are
    # not evaluated twice); complex psky keeps the plain blocks (its single-pass forms depend on the pair orientation, which
    # a re-ordering of the rows changes)
    if MIRROR:
        posh = pos.cpu().numpy()
        mb = [(_mirror_block(b, posh[np.asarray(r['ants_i'])], dev) if r['ants_j'] is None else None) for b, r in zip(blocks, raw)]
        # ... and, where a block of more than 64 antennas fits into 64 rows of firsts (+ the hub), on the conjugate-pair form,
        # which does not contract the mirror rows either
        pb = [(_pair_block(b, posh[np.asarray(r['ants_i'])], dev) if (PAIR and r['ants_j'] is None) else None)
              for b, r in zip(blocks, raw)]
        if any(m is not None for m in mb) or any(q is not None for q in pb):
            self.ant['blocks_mirror'] = [m if m is not None else b for m, b in zip(mb, blocks)]
            self.ant['blocks_real'] = [q if q is not None else m for q, m in zip(pb, self.ant['blocks_mirror'])]
            self.ant['mirror_groups'] = [(bin(m['mirror']).count('1'), (m['nrows'] + 15) // 16)
                                         for m, q in zip(mb, pb) if m is not None and q is None]
            self.ant['pair_blocks'] = [(sum(1 for x in q['partner'] if x >= 0), q['nrows'], int(q['hub'] is not None))
                                       for q in pb if q is not None]
            if PAIR_CPLX and any(q is not None for q in pb):
                # complex psky: pair blocks (two real passes: fwd_cpass = cpass = 0) in place of their plain blocks
                self.ant['blocks_cplx'] = [q if q is not None else b for q, b in zip(pb, blocks)]
                mask = torch.zeros_like(two_pass_mask)
                for b in self.ant['blocks_cplx']:
                    if b['fwd_cpass'] == 0:
                        tabs = [b['direct'], b['conj']] + ([b['centre']] if b.get('centre') is not None else [])
                        slots = torch.cat([t[t >= 0] for t in tabs]).to(torch.int64)
                        mask[slots] = 1.0
                self.ant['two_pass_mask_cplx'] = mask
            self.ant['mfma_fwd'] = sum(b['mf_fwd'] for b in self.ant['blocks_real'])
            self.ant['mfma_bwd'] = sum(b['mf_bwd_real'] for b in self.ant['blocks_real'])
            self.ant['mfma_flops_fwd'], self.ant['mfma_flops_bwd'] = per16 * self.ant['mfma_fwd'], per16 * self.ant['mfma_bwd']


FringeGeometry._setup_antenna_path = _setup_antenna_path


def _fringe_algorithmic_bytes(geom, a, b):
    """HBM bytes a fused fringe sum cannot avoid: psky (or gpsky) once, the visibilities (or their
    gradient) once, the pointing vectors once -- both directions move the same tensors"""
    return int(a.numel() * a.element_size() + b.numel() * b.element_size() + geom.sdir.numel() * geom.sdir.element_size())


def _fringe_call(geom, backward, inp, out, Npp, cplx, strides=None):
    if geom.ant is not None and inp.dtype == torch.float32 and strides is not None:
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        flops = _fringe_ant_call(geom, backward, inp, out, strides, Npp, cplx)
        if prof is not None:
            e1.record()
            prof.append(('fringe_ant_bwd_kernel' if backward else 'fringe_ant_fwd_kernel', e0, e1, geom.elements, flops,
                         _fringe_algorithmic_bytes(geom, inp, out)))
        return
    code, rdt = _real_dtype(inp)
    fn = lib.rime_fringe_sum_bwd if backward else lib.rime_fringe_sum_fwd
    nbytes = lib.rime_fringe_sum_workspace(code, geom.Nbl, geom.Nt, geom.Nf, geom.Pstride,
                                           geom.Nmp, Npp, int(cplx), int(backward))
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=inp.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = fn(code, _ptr(geom.blvecs), _ptr(geom.sdir), _ptr(geom.freqs), _ptr(inp),
            geom.mp_offsets, _ptr(geom.bl_order), geom.Nbl, geom.Nt, geom.Nf, geom.Pstride,
            geom.Nmp, Npp, int(cplx), geom.sign, int(geom.uniform), geom.f0, geom.df,
            geom.max_blen, strides, _ptr(out), _ptr(ws), ws.numel(), _stream())
    check(rc, 'rime_fringe_sum_bwd' if backward else 'rime_fringe_sum_fwd')
    if prof is not None:
        e1.record()
        prof.append(('fringe_bwd_kernel' if backward else 'fringe_fwd_kernel', e0, e1, geom.elements, 0,
                     _fringe_algorithmic_bytes(geom, inp, out)))


def _npp_chunks(Npp, cplx):
    """polarisation-product planes in runs the baseline-formulation kernels take natively (real psky: 4, 2, 1
    planes per launch; complex: 4 or 1): imaging puts any number of maps on this axis"""
    sizes = (4, 1) if cplx else (4, 2, 1)
    out, q = [], 0
    while q < Npp:
        n = next(k for k in sizes if k <= Npp - q)
        out.append((q, n))
        q += n
    return out


def _fringe_call_planes(geom, backward, psky_like, vis_like, cplx):
    """_fringe_call over a psky-shaped tensor (Nt, Nmp, Npp, Nf, P) and a vis-shaped tensor (Npp, Nbl, Nt, Nf):
    one launch when the kernels take the plane count as it is, else one per run of planes"""
    Npp = psky_like.shape[2]
    native = (geom.ant is not None and psky_like.dtype in (torch.float32, torch.complex64)) or \
        (Npp in (1, 4)) or (Npp == 2 and not cplx)
    for q0, n in ([(0, Npp)] if native else _npp_chunks(Npp, cplx)):
        p = psky_like.narrow(2, q0, n)
        v = vis_like.narrow(0, q0, n)
        pr, vr = (torch.view_as_real(p) if cplx else p), torch.view_as_real(v)
        if backward:
            _fringe_call(geom, True, vr, pr, n, cplx, _dense_strides(p))
        else:
            _fringe_call(geom, False, pr, vr, n, cplx, _dense_strides(p))


def _blvec_grad(geom, p, g, cplx):
    """
    d loss / d blvecs [Nbl, 3] of the fringe sum: d vis / d b_k = sum_p psky (sign 2 pi i nu / c) s_k F, i.e. the
    FORWARD kernels on the three direction-cosine-weighted copies of psky (telescope_model.py:350-356 is
    differentiable w.r.t. blvecs through torch autograd; here it costs three more forward passes and is only
    computed when the baseline vectors require a gradient).  p: psky (Nt, Nmp, Npp, Nf, Ps); g: gvis (Npp, Nbl, Nt, Nf).
    """
    Nt, Nmp, Npp, Nf, Ps = p.shape
    rdt = torch.float32 if p.dtype in (torch.float32, torch.complex64) else torch.float64
    s3 = geom.sdir.to(rdt)                                                   # (Nt, 3, Ps)
    pw = (p[:, :, :, None] * s3[:, None, None, :, None, :]).reshape(Nt, Nmp, Npp * 3, Nf, Ps).contiguous()
    w = torch.empty((Npp * 3, geom.Nbl, Nt, Nf), dtype=g.dtype, device=g.device)
    _fringe_call_planes(geom, False, pw, w, cplx)
    w = w.reshape(Npp, 3, geom.Nbl, Nt, Nf)
    fac = (geom.sign * 2.0 * np.pi / 2.99792458e8) * geom.freqs                # (Nf,) float64
    # Re( conj(g) * i fac * w ) = -fac * Im( conj(g) w )
    t = (g.conj()[:, None] * w).imag.to(torch.float64) * (-fac)
    return t.sum(dim=(0, 3, 4)).t().contiguous()                             # (Nbl, 3)


class _FringeSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, psky, geom, blvecs=None):
        _require_cuda(psky)
        assert psky.dim() == 5, 'psky must be (Nt, Nmp, Npp, Nf, Pstride)'
        Nt, Nmp, Npp, Nf, Ps = psky.shape
        assert (Nt, Nmp, Nf, Ps) == (geom.Nt, geom.Nmp, geom.Nf, geom.Pstride), \
            'psky %s does not match geometry (Nt=%d, Nmp=%d, Nf=%d, Pstride=%d)' % (
                tuple(psky.shape), geom.Nt, geom.Nmp, geom.Nf, geom.Pstride)
        cplx = psky.is_complex()
        p = psky.detach()
        strides = _dense_strides(p)          # permuted (e.g. time-inner) layouts are read in place
        if strides is None:
            p = p.contiguous()
            strides = _dense_strides(p)
        _, rdt = _real_dtype(p)
        cdt = torch.complex64 if rdt == torch.float32 else torch.complex128
        vis = torch.empty((Npp, geom.Nbl, Nt, Nf), dtype=cdt, device=p.device)
        _fringe_call_planes(geom, False, p, vis, cplx)
        ctx.geom, ctx.cplx, ctx.dtype = geom, cplx, psky.dtype
        ctx.pshape, ctx.pstride = tuple(p.shape), tuple(p.stride())   # gradient buffer template only
        ctx.psky = p if (blvecs is not None and blvecs.requires_grad) else None     # kept only for the blvecs gradient
        return vis

    @staticmethod
    def backward(ctx, gvis):
        geom = ctx.geom
        g = gvis.contiguous()
        gp = None
        if ctx.needs_input_grad[0]:
            gp = torch.empty_strided(ctx.pshape, ctx.pstride, dtype=ctx.dtype, device=g.device)
            _fringe_call_planes(geom, True, gp, g, ctx.cplx)
        gb = None
        if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2] and ctx.psky is not None:
            gb = _blvec_grad(geom, ctx.psky, g, ctx.cplx)
        return gp, None, gb


def fringe_sum(psky, geom, blvecs=None):
    """
    psky (Nt, Nmp, Npp, Nf, Pstride) real or complex on the GPU; geom a FringeGeometry.
    Returns vis (Npp, Nbl, Nt, Nf) complex.  Differentiable w.r.t. psky and, when `blvecs` -- the (Nbl, 3)
    tensor the geometry was built from, still attached to its graph -- requires a gradient, w.r.t. the baseline
    vectors (hence antenna positions): three extra forward passes in the backward.
    """
    if blvecs is not None and blvecs.requires_grad:
        assert tuple(blvecs.shape) == (geom.Nbl, 3)
        return _FringeSum.apply(psky, geom, blvecs)
    return _FringeSum.apply(psky, geom)


def fringe_adjoint(gvis, geom, Npp=None, dtype=None):
    """
    The adjoint of fringe_sum for a REAL psky, without autograd:  out[t, 0, q, f, p] = Re sum_b
    conj(F[b, t, f, p]) gvis[q, b, t, f] -- the map-making direction (visibilities -> pixels,
    imaging.make_map).  gvis (Npp, Nbl, Nt, Nf) complex; returns (Nt, 1, Npp, Nf, Pstride) real.
    """
    _require_cuda(gvis)
    assert gvis.is_complex() and gvis.dim() == 4 and geom.Nmp == 1
    g = gvis.detach().contiguous()
    Npp = g.shape[0] if Npp is None else Npp
    assert tuple(g.shape) == (Npp, geom.Nbl, geom.Nt, geom.Nf)
    rdt = torch.float32 if g.dtype == torch.complex64 else torch.float64
    out = torch.empty((geom.Nt, 1, Npp, geom.Nf, geom.Pstride), dtype=rdt, device=g.device)
    _fringe_call_planes(geom, True, out, g, False)
    return out


def gen_fringe(blvecs, sdir, freqs, conj=False, dtype=torch.float32):
    """materialised fringe (Nbl, Nf, P) complex; blvecs (Nbl,3), sdir (3,P) float64 on the GPU"""
    _require_cuda(blvecs, sdir)
    b = blvecs.detach().to(torch.float64).contiguous()
    s = sdir.detach().to(torch.float64).contiguous()
    f = torch.as_tensor(freqs, dtype=torch.float64).to(b.device).contiguous()
    cdt = torch.complex64 if dtype == torch.float32 else torch.complex128
    out = torch.empty((b.shape[0], len(f), s.shape[1]), dtype=cdt, device=b.device)
    rc = lib.rime_gen_fringe(RIME_F32 if dtype == torch.float32 else RIME_F64, _ptr(b), _ptr(s), _ptr(f),
                             b.shape[0], len(f), s.shape[1], s.shape[1], -1 if conj else 1,
                             _ptr(torch.view_as_real(out)), _stream())
    check(rc, 'rime_gen_fringe')
    return out


def eq2top(ra, dec, M, vbary, vdiurnal):
    """
    ICRS (ra, dec) [deg, float64 on the GPU] -> (2, N) float64 tensor (zen, az) [deg]: annual
    aberration, rotation by the host-built matrix M (3, 3) (ICRS -> East, North, Up), diurnal
    aberration (astrometry.observation_frame gives M, vbary, vdiurnal per observation time).
    """
    _require_cuda(ra, dec)
    r = ra.detach().to(torch.float64).contiguous()
    d = dec.detach().to(torch.float64).contiguous()
    assert r.dim() == 1 and r.shape == d.shape
    out = torch.empty((2, r.numel()), dtype=torch.float64, device=r.device)
    Mh = (ctypes.c_double * 9)(*[float(x) for x in np.asarray(M, dtype=np.float64).reshape(-1)])
    vh = (ctypes.c_double * 3)(*[float(x) for x in np.asarray(vbary, dtype=np.float64).reshape(-1)])
    rc = lib.rime_eq2top(_ptr(r), _ptr(d), r.numel(), ctypes.cast(Mh, ctypes.c_void_p), ctypes.cast(vh, ctypes.c_void_p),
                         float(vdiurnal), _ptr(out[0]), _ptr(out[1]), _stream())
    check(rc, 'rime_eq2top')
    return out


# ---------------------------------------------------------------------------------------
class _JonesApply(torch.autograd.Function):
    """psky[a, d] = sum_bc J1[a, b] S[b, c] conj(J2[d, c]) in one pass (csrc/jones.hip); J2 is J1 when `same`"""
    @staticmethod
    def forward(ctx, J1, J2, S, same):
        _require_cuda(J1, J2, S)
        j1 = J1.detach().contiguous()
        j2 = j1 if same else J2.detach().contiguous()
        sk = S.detach().contiguous()
        assert sk.is_complex() and tuple(j1.shape[:2]) == (2, 2) and tuple(sk.shape[:2]) == (2, 2) and j1.shape == j2.shape
        assert j2.dtype == j1.dtype, 'both Jones operands must have one dtype (%s vs %s)' % (j1.dtype, j2.dtype)
        code, rdt = _real_dtype(sk)
        bc = j1.is_complex()
        assert (j1.dtype == sk.dtype) if bc else (j1.dtype == rdt), 'beam %s vs sky %s' % (j1.dtype, sk.dtype)
        N, Ns = j1[0, 0].numel(), sk[0, 0].numel()
        assert N % Ns == 0 and tuple(j1.shape[-2:]) == tuple(sk.shape[-2:])
        out = torch.empty(j1.shape, dtype=sk.dtype, device=sk.device)
        vr = torch.view_as_real
        rc = lib.rime_jones_apply_fwd(code, int(bc), _ptr(vr(j1) if bc else j1), _ptr(vr(j2) if bc else j2), _ptr(vr(sk)),
                                      N, Ns, _ptr(vr(out)), _stream())
        check(rc, 'rime_jones_apply_fwd')
        ctx.save_for_backward(j1, j2, sk)
        ctx.same, ctx.sky_shape = same, tuple(S.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        j1, j2, sk = ctx.saved_tensors
        g = g.contiguous()
        code, _ = _real_dtype(sk)
        bc = j1.is_complex()
        N, Ns = j1[0, 0].numel(), sk[0, 0].numel()
        g1, g2 = torch.empty_like(j1), torch.empty_like(j1)
        gs = torch.empty(j1.shape, dtype=sk.dtype, device=sk.device)
        vr = torch.view_as_real
        rc = lib.rime_jones_apply_bwd(code, int(bc), _ptr(vr(j1) if bc else j1), _ptr(vr(j2) if bc else j2), _ptr(vr(sk)),
                                      _ptr(vr(g)), N, Ns, _ptr(vr(g1) if bc else g1), _ptr(vr(g2) if bc else g2),
                                      _ptr(vr(gs)), _stream())
        check(rc, 'rime_jones_apply_bwd')
        if N != Ns:                                       # one sky for several model pairs: sum their contributions
            gs = gs.reshape(2, 2, N // Ns, Ns).sum(2)
        gs = gs.reshape(ctx.sky_shape)
        if ctx.same:
            return g1 + g2, None, gs, None
        return g1, g2, gs, None


def jones_apply(J1, J2, S):
    """
    Full-polarisation beam x sky product J1 S J2^dagger per pixel, channel and beam-model pair (the 4-pol branch of
    PixelBeam.apply_beam, beam_model.py:345-363).  J1, J2 (2, 2, Nmp, Nf, P) real or complex (J2 may be J1 itself),
    S (2, 2, 1 | Nmp, Nf, P) complex -> (2, 2, Nmp, Nf, P) complex.
    """
    return _JonesApply.apply(J1, J2, S, J2 is J1)


class _Stokes2Coh(torch.autograd.Function):
    """C = I [[1 + fQ, fU - i fV], [fU + i fV, 1 - fQ]] in one pass each way (csrc/jones.hip); fractions without gradient"""
    @staticmethod
    def forward(ctx, I, frb):
        _require_cuda(I, frb)
        x = I.detach().contiguous()
        code, rdt = _real_dtype(x)
        cdt = torch.complex64 if rdt == torch.float32 else torch.complex128
        P = x.shape[-1]
        R = x.numel() // P
        out = torch.empty((2, 2) + tuple(x.shape), dtype=cdt, device=x.device)
        # frb: (3,) + I.shape expanded view of the fractions: element strides, 0 on broadcast axes; rows must share ONE stride
        fs = _frac_strides(frb, x.shape)
        check(lib.rime_stokes2coh_fwd(code, _ptr(x), _ptr(frb), fs[0], fs[1], fs[2], R, P, _ptr(torch.view_as_real(out)), _stream()),
              'rime_stokes2coh_fwd')
        ctx.frb, ctx.fs, ctx.shape, ctx.code = frb, fs, tuple(I.shape), code
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        P = ctx.shape[-1]
        R = int(np.prod(ctx.shape[:-1])) if len(ctx.shape) > 1 else 1
        gI = torch.empty(ctx.shape, dtype=ctx.frb.dtype, device=g.device)
        check(lib.rime_stokes2coh_bwd(ctx.code, _ptr(torch.view_as_real(g)), _ptr(ctx.frb), ctx.fs[0], ctx.fs[1], ctx.fs[2], R, P,
                                      _ptr(gI), _stream()), 'rime_stokes2coh_bwd')
        return gI, None


def _frac_strides(frb, shape):
    """(fs_k, fs_r, fs_p) element strides of an expanded (3,) + shape view, or None when the leading axes of `shape` cannot
    be addressed by one row stride (then the caller keeps the torch composition)"""
    st = frb.stride()
    fs_p = st[-1] if shape[-1] > 1 else 0
    lead, lst = list(shape[:-1]), list(st[1:-1])
    # rows r = flattened leading axes: one stride only if the axes are jointly contiguous or all broadcast
    live = [(n, s_) for n, s_ in zip(lead, lst) if n > 1]
    if not live or all(s_ == 0 for _, s_ in live):
        fs_r = 0
    else:
        fs_r = live[-1][1]
        acc = fs_r
        for n, s_ in reversed(live):
            if s_ != acc:
                return None
            acc *= n
    return (st[0], fs_r, fs_p)


def stokes2coherency(I, frac):
    """
    Stokes I map (..., Npix) real on the GPU + fractional polarisation `frac` (3, 1, ...) broadcastable to (3,) + I.shape
    (fQ, fU, fV) -> coherency (2, 2, ..., Npix) complex (sky_model.Stokes2Coherency's Stokes-I branch, sky_model.py:1160-1300)
    in one fused pass; differentiable w.r.t. I.  Returns None when the layout is not served (fractions that require a
    gradient, another dtype, leading axes without a common stride): the caller keeps the torch composition.
    """
    if not (I.is_cuda and I.dtype in (torch.float32, torch.float64) and not I.is_complex()):
        return None
    if frac.requires_grad or frac.dtype != I.dtype or frac.shape[0] != 3 or frac.device != I.device:
        return None
    # axis 1 of `frac` is the singleton the reference keeps there; the axes behind it must line up ONE TO ONE with I's (a frac
    # with fewer trailing axes would be left-padded by expand and could land its Q/U/V axis on I's channel axis: ADVICE r04)
    if frac.dim() - 2 != I.dim() or frac.shape[1] != 1:
        return None
    try:
        frb = frac.detach().reshape((3,) + tuple(frac.shape[2:])).expand((3,) + tuple(I.shape))
    except RuntimeError:
        return None
    if _frac_strides(frb, tuple(I.shape)) is None or any(s_ < 0 for s_ in frb.stride()):
        return None
    return _Stokes2Coh.apply(I, frb)


# ---------------------------------------------------------------------------------------
class InterpStencil:
    """(inds, wgts) of PixInterp plus the CSR inverse index the deterministic adjoint uses."""
    def __init__(self, inds, wgts, Npb):
        _require_cuda(inds, wgts)
        self.P, self.Nnn = inds.shape
        self.Npb = int(Npb)
        self.inds = inds.to(torch.int32).contiguous()
        self.wgts = wgts.contiguous()
        flat = self.inds.reshape(-1).to(torch.int64)
        # entries of weight 0 (the padding slots of a one-node FoV-cut stencil all point at the last pixel; samples on a
        # grid node) stay out of the inverse index: the gather skips them too (w != 0), so 0 * inf / NaN of a padded
        # gradient slot cannot reach the map gradient, and no node collects thousands of dead entries for one wave to walk
        keep = torch.nonzero(self.wgts.reshape(-1) != 0).reshape(-1)
        order = keep[torch.sort(flat[keep], stable=True).indices]
        counts = torch.bincount(flat[keep], minlength=self.Npb)
        ptr = torch.zeros(self.Npb + 1, dtype=torch.int64, device=inds.device)
        ptr[1:] = torch.cumsum(counts, 0)
        self.csr_ptr = ptr.to(torch.int32).contiguous()
        self.csr_src = order.to(torch.int32).contiguous()
        self._wcache = {self.wgts.dtype: self.wgts}

    def weights(self, dtype):
        if dtype not in self._wcache:
            self._wcache[dtype] = self.wgts.to(dtype).contiguous()
        return self._wcache[dtype]


class _InterpGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, st, out_stride):
        _require_cuda(m)
        lead = m.shape[:-1]
        Npb = m.shape[-1]
        assert Npb == st.Npb, 'map has %d pixels, stencil was built for %d' % (Npb, st.Npb)
        cplx = m.is_complex()
        mm = m.detach().contiguous()
        code, rdt = _real_dtype(mm)
        R = int(np.prod(lead)) if len(lead) else 1
        out = torch.zeros(lead + (out_stride,), dtype=m.dtype, device=m.device)
        rc = lib.rime_interp_gather_fwd(code, int(cplx), _ptr(torch.view_as_real(mm) if cplx else mm),
                                        _ptr(st.inds), _ptr(st.weights(rdt)), R, Npb, st.P, st.Nnn,
                                        _ptr(torch.view_as_real(out) if cplx else out), out_stride,
                                        _stream())
        check(rc, 'rime_interp_gather_fwd')
        ctx.st, ctx.shape, ctx.cplx, ctx.out_stride = st, tuple(m.shape), cplx, out_stride
        return out

    @staticmethod
    def backward(ctx, gout):
        st = ctx.st
        code, rdt = _real_dtype(gout)
        R = int(np.prod(ctx.shape[:-1])) if len(ctx.shape) > 1 else 1
        if st.Nnn == 1:
            # one-node stencil (FoV cut, redundant inflation): row-major adjoint, no transposed copies, and the gradient that
            # goes upstream is contiguous
            g = gout.contiguous()
            gm = torch.empty(ctx.shape, dtype=gout.dtype, device=gout.device)
            rc = lib.rime_interp_scatter_rows_bwd(code, int(ctx.cplx), _ptr(torch.view_as_real(g) if ctx.cplx else g),
                                                  ctx.out_stride, _ptr(st.csr_ptr), _ptr(st.csr_src), _ptr(st.weights(rdt)),
                                                  R, st.Npb, _ptr(torch.view_as_real(gm) if ctx.cplx else gm), _stream())
            check(rc, 'rime_interp_scatter_rows_bwd')
            return gm, None, None
        if st.csr_src.numel() == 0:                      # every weight 0: nothing reaches the map
            return torch.zeros(ctx.shape, dtype=gout.dtype, device=gout.device), None, None
        # transposed working layout [pixel][row]: every read / write of the kernel is coalesced
        gT = gout.reshape(R, ctx.out_stride)[:, :st.P].t().contiguous()
        gmT = torch.empty((st.Npb, R), dtype=gout.dtype, device=gout.device)
        rc = lib.rime_interp_scatter_bwd(code, int(ctx.cplx),
                                         _ptr(torch.view_as_real(gT) if ctx.cplx else gT),
                                         _ptr(st.csr_ptr), _ptr(st.csr_src), _ptr(st.weights(rdt)),
                                         R, st.Npb, st.P, st.Nnn,
                                         _ptr(torch.view_as_real(gmT) if ctx.cplx else gmT), _stream())
        check(rc, 'rime_interp_scatter_bwd')
        return gmT.t().contiguous().reshape(ctx.shape), None, None


def interp_gather(m, stencil, out_stride=None):
    """out[..., p] = sum_k w[p,k] m[..., inds[p,k]]; columns P..out_stride are zero padding."""
    return _InterpGather.apply(m, stencil, stencil.P if out_stride is None else int(out_stride))


class _NodeMajor(torch.autograd.Function):
    """(R, Npix_beam) beam map -> node-major (Npix_beam, R) copy, the layout the fused psky builder gathers from (a node's value
    for 4 channels is one vector load), as an autograd node of its own: a forward with several sky components (C4: diffuse +
    point sources) transposes the map ONCE, and the components' node-major gradients are summed before the ONE transposition
    back (round 5: one 34-MB transposing copy less each way per C4 step)"""
    @staticmethod
    def forward(ctx, bmap):
        return bmap.detach().t().contiguous()

    @staticmethod
    def backward(ctx, g):
        # one explicit transposition: everything upstream (the |.| of a power beam, the accumulation into .grad) then runs on a
        # contiguous tensor instead of a transposed view
        return g.t().contiguous()


def node_major(bmap):
    return _NodeMajor.apply(bmap)


class _BeamSkyProduct(torch.autograd.Function):
    """psky[r, q] = interp(bmap)[r, q] * sky[r, cut[q]] in one pass (see rime_beam_sky_fwd); bmapT = the NODE-MAJOR map"""
    @staticmethod
    def forward(ctx, bmapT, sky, st, cut, pos, Nt, Ps):
        _require_cuda(bmapT, sky)
        Npb, R = bmapT.shape
        Npix = sky.shape[1]
        assert sky.shape[0] == R and Npb == st.Npb and st.P == Nt * Ps and bmapT.dtype == sky.dtype
        b, k = bmapT.detach().contiguous(), sky.detach().contiguous()
        code, rdt = _real_dtype(b)
        out = torch.empty((R, Nt * Ps), dtype=b.dtype, device=b.device)
        rc = lib.rime_beam_sky_fwd(code, _ptr(b), _ptr(k), _ptr(st.inds), _ptr(st.weights(rdt)), _ptr(cut),
                                   R, Npb, Npix, Nt * Ps, st.Nnn, _ptr(out), _stream())
        check(rc, 'rime_beam_sky_fwd')
        ctx.save_for_backward(b, k)
        ctx.aux = (st, cut, pos, Nt, Ps)
        return out

    @staticmethod
    def backward(ctx, g):
        b, k = ctx.saved_tensors
        st, cut, pos, Nt, Ps = ctx.aux
        Npb, R = b.shape
        Npix = k.shape[1]
        Q = Nt * Ps
        code, rdt = _real_dtype(b)
        g = g.contiguous()
        T1 = torch.empty((Q, R), dtype=b.dtype, device=b.device)
        gsky = torch.empty((R, Npix), dtype=b.dtype, device=b.device)
        nws = int(lib.rime_beam_sky_bwd_workspace(code, R, Npix, Nt))
        ws = torch.empty(nws, dtype=torch.uint8, device=b.device) if nws else None
        rc = lib.rime_beam_sky_bwd(code, _ptr(g), _ptr(b), _ptr(k), _ptr(st.inds), _ptr(st.weights(rdt)), _ptr(cut),
                                   _ptr(pos), R, Npb, Npix, Nt, Ps, st.Nnn, _ptr(T1), _ptr(gsky),
                                   _ptr(ws) if ws is not None else None, nws, _stream())
        check(rc, 'rime_beam_sky_bwd')
        gmT = torch.empty((Npb, R), dtype=b.dtype, device=b.device)
        rc = lib.rime_interp_scatter_bwd(code, 0, _ptr(T1), _ptr(st.csr_ptr), _ptr(st.csr_src), _ptr(st.weights(rdt)),
                                         R, Npb, Q, st.Nnn, _ptr(gmT), _stream())
        check(rc, 'rime_interp_scatter_bwd')
        return gmT, gsky, None, None, None, None, None


def beam_sky_product(bmap, sky, stencil, cut, pos, Nt, Ps, node_major_map=None):
    """
    Fused interpolate-cut-multiply of the 1-pol power-beam case: bmap (R, Npix_beam) and sky (R, Npix)
    real, same dtype; stencil an InterpStencil over the Nt*Ps (padded) pointing angles; cut int32
    (Nt*Ps) sky-pixel index per point (Npix = padding); pos int32 (Nt, Npix) its inverse per time step
    (-1 = not visible).  Returns psky (R, Nt*Ps).  Differentiable w.r.t. bmap and sky.
    `node_major_map`: ops.node_major(bmap) made by the caller (shared by several calls on the same map); bmap is then ignored.
    """
    bT = node_major(bmap) if node_major_map is None else node_major_map
    return _BeamSkyProduct.apply(bT, sky, stencil, cut, pos, int(Nt), int(Ps))


class _Chisq(torch.autograd.Function):
    """chi^2 = sum icov |pred - data|^2 and its backward in one pass each (see rime_chisq_fwd)"""
    @staticmethod
    def forward(ctx, pred, data, icov):
        _require_cuda(pred)
        assert pred.is_complex(), 'prediction must be complex'
        p = pred.detach().contiguous()
        code, rdt = _real_dtype(p)
        d = None if data is None else data.detach().to(p.dtype).expand_as(p).contiguous()
        w = None if icov is None else icov.detach().to(rdt).expand_as(p).contiguous()
        out = torch.empty((), dtype=rdt, device=p.device)
        ws = torch.empty(int(lib.rime_chisq_workspace()), dtype=torch.uint8, device=p.device)
        rc = lib.rime_chisq_fwd(code, _ptr(torch.view_as_real(p)), _ptr(torch.view_as_real(d)) if d is not None else None,
                                _ptr(w) if w is not None else None, p.numel(), _ptr(out), _ptr(ws), ws.numel(), _stream())
        check(rc, 'rime_chisq_fwd')
        ctx.saved = (p, d, w)
        return out

    @staticmethod
    def backward(ctx, g):
        p, d, w = ctx.saved
        code, rdt = _real_dtype(p)
        gp = torch.empty_like(p)
        gg = g.detach().to(rdt).reshape(1).contiguous()
        rc = lib.rime_chisq_bwd(code, _ptr(torch.view_as_real(p)), _ptr(torch.view_as_real(d)) if d is not None else None,
                                _ptr(w) if w is not None else None, _ptr(gg), p.numel(),
                                _ptr(torch.view_as_real(gp)), _stream())
        check(rc, 'rime_chisq_bwd')
        return gp, None, None


def chisq(pred, data=None, icov=None):
    """
    sum_i icov_i |pred_i - data_i|^2 (real scalar) for a complex prediction; data complex or None (0),
    icov real (inverse variance, diagonal) or None (1), both broadcastable to pred.  Differentiable
    w.r.t. pred.  The fused form of LogProb.forward_chisq's residual + apply_icov(cov_axis=None) + sum.
    """
    return _Chisq.apply(pred, data, icov)


# ---------------------------------------------------------------------------------------
def _antenna_incidence(idx, Nant, dtype):
    """dense 0/1 [Nant, Nbl] matrix of which baselines an antenna slot takes part in (cached on idx)"""
    tag = getattr(idx, '_rime_incidence', None)
    if tag is not None and tag[0] == (idx._version, Nant, dtype):
        return tag[1]
    M = torch.zeros(Nant, idx.numel(), dtype=dtype, device=idx.device)
    M[idx.long(), torch.arange(idx.numel(), device=idx.device)] = 1
    try:
        idx._rime_incidence = ((idx._version, Nant, dtype), M)
    except Exception:
        pass
    return M


class _ApplyCal(torch.autograd.Function):
    """V' = G1 V G2^dagger in one pass over the visibilities; backward in one pass + two incidence products"""
    @staticmethod
    def forward(ctx, vis, gains, a1, a2, diag):
        _require_cuda(vis, gains, a1, a2)
        assert vis.is_complex() and gains.is_complex(), 'complex visibilities and gains'
        NP = vis.shape[0]
        assert vis.ndim == 5 and gains.ndim == 5 and vis.shape[:2] == (NP, NP) and gains.shape[:2] == (NP, NP)
        assert NP in (1, 2), 'Npol must be 1 or 2'
        _, _, Nbl, Nt, Nf = vis.shape
        Nant, Ntg, Nfg = gains.shape[2:]
        assert Ntg in (1, Nt) and Nfg in (1, Nf), 'gains must match or broadcast over time / channel'
        assert a1.dtype == torch.int32 and a2.dtype == torch.int32 and a1.numel() == Nbl and a2.numel() == Nbl
        v = vis.detach().contiguous()
        g = gains.detach().to(v.dtype).contiguous()
        code, _ = _real_dtype(v)
        st = (Nant * Ntg * Nfg, Ntg * Nfg, Nfg if Ntg > 1 else 0, 1 if Nfg > 1 else 0)
        out = torch.empty_like(v)
        rc = lib.rime_apply_cal_fwd(code, NP, int(bool(diag)), _ptr(torch.view_as_real(v)), _ptr(torch.view_as_real(g)),
                                    _ptr(a1), _ptr(a2), Nbl, Nt, Nf, Nant, *st, _ptr(torch.view_as_real(out)), _stream())
        check(rc, 'rime_apply_cal_fwd')
        ctx.saved = (v, g, a1, a2, st, bool(diag), gains.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        v, g, a1, a2, st, diag, gdt = ctx.saved
        NP, _, Nbl, Nt, Nf = v.shape
        Nant, Ntg, Nfg = g.shape[2:]
        code, rdt = _real_dtype(v)
        go = gout.detach().to(v.dtype).contiguous()
        gvis = torch.empty_like(v)
        d = torch.empty((2,) + tuple(v.shape), dtype=v.dtype, device=v.device)
        rc = lib.rime_apply_cal_bwd(code, NP, int(diag), _ptr(torch.view_as_real(v)), _ptr(torch.view_as_real(g)),
                                    _ptr(torch.view_as_real(go)), _ptr(a1), _ptr(a2), Nbl, Nt, Nf, Nant, *st,
                                    _ptr(torch.view_as_real(gvis)), _ptr(torch.view_as_real(d[0])),
                                    _ptr(torch.view_as_real(d[1])), _stream())
        check(rc, 'rime_apply_cal_bwd')
        ggains = None
        if ctx.needs_input_grad[1]:
            dr = torch.view_as_real(d).reshape(2, NP * NP, Nbl, Nt * Nf * 2)
            gg = torch.matmul(_antenna_incidence(a1, Nant, rdt), dr[0])
            gg += torch.matmul(_antenna_incidence(a2, Nant, rdt), dr[1])
            gg = torch.view_as_complex(gg.reshape(NP, NP, Nant, Nt, Nf, 2))
            if Ntg == 1 and Nt > 1:
                gg = gg.sum(3, keepdim=True)
            if Nfg == 1 and Nf > 1:
                gg = gg.sum(4, keepdim=True)
            ggains = gg.to(gdt)
        return (gvis if ctx.needs_input_grad[0] else None), ggains, None, None, None


def apply_cal(vis, gains, a1, a2, diag=False):
    """
    V'[p, q, b, t, f] = sum G1[p, r] V[r, s] conj(G2[q, s]) with G1 = gains[:, :, a1[b]], G2 = gains[:, :, a2[b]]
    (calibration._apply_cal, calibration.py:2412-2487, complex visibilities).  vis (Np, Np, Nbl, Nt, Nf)
    complex; gains (Np, Np, Nant, Nt | 1, Nf | 1) complex; a1, a2 int32 [Nbl] on the GPU.  diag: only the
    diagonal products, off-diagonal results zero (the reference's '2pol' mode).  Differentiable w.r.t.
    vis and gains.
    """
    return _ApplyCal.apply(vis, gains, a1, a2, diag)


# ---------------------------------------------------------------------------------------
ALM_SPLIT_F16 = True        # float32: f16 hi/lo split operands on the f16 matrix cores (22 bits) instead
                            # of the exact-f32 MFMA kernels (set False to force those)


def _ylm_scale(Y):
    """
    power of two bringing max|Ylm| to [2^10, 2^11].  Ylm matrices are reused every forward and a max
    over C3's 3 GB matrix costs as much as the transform itself, so the value is remembered ON the
    tensor object (with its version counter): a different tensor, or an in-place update, recomputes.
    """
    tag = getattr(Y, '_rime_yscale', None)
    if tag is not None and tag[0] == Y._version:
        return tag[1]
    amax = float(Y.detach().abs().amax().item())
    s = float(2.0 ** np.floor(np.log2(2048.0 / amax))) if amax > 0 and np.isfinite(amax) else 1.0
    try:
        Y._rime_yscale = (Y._version, s)
    except Exception:
        pass
    return s


ALM_PACKED = os.environ.get('RIME_ALM_PACKED', '1') != '0'      # cached pre-split f16 copies of Ylm in fragment order
ALM_PACKED_MIN_BYTES = 1 << 24                                   # below 16 MB of Ylm the transform is launch-bound anyway


# registry of the packed copies: id(Ylm tensor object) -> (weak reference to it, tag).  The copies are derived data that must
# not travel with the tensor (a torch.cuda.Event cannot be pickled or deep-copied, and a tensor's __dict__ is): a pickled /
# deep-copied model packs again on first use (round 5).  The weak reference's callback drops the entry -- and with it the two
# buffers -- when the tensor object dies; a recycled id() is caught by comparing the referent.
_YLM_PACKED = {}


def ylm_packed_state(Ylm):
    """the cache entry of this Ylm tensor OBJECT: (version, y_scale, data_ptr, {direction: (buffer, event, stream) | False}) or None"""
    ent = _YLM_PACKED.get(id(Ylm))
    return ent[1] if ent is not None and ent[0]() is Ylm else None


def _ylm_packed(Ylm, Y, ys, direction):
    """
    The packed copy of Ylm for one direction (0 forward, 1 backward; include/rime_hip.h: rime_alm2pix_pack), built on
    first use and remembered FOR the Ylm tensor object (module registry keyed on the object, see above) together with its
    version counter and y_scale -- a different tensor (AlmModel.setup_Ylm replaces it) or an in-place update packs again.
    Returns None when packing is switched
    off, the matrix is small, or the GPU has no room for another copy (the unpacked kernels then run).
    `Ylm` is the caller's tensor object (the cache belongs to it), `Y` its contiguous detached alias.

    MEMORY: each direction keeps a buffer of Ylm's own size on the device (8 bytes per (coefficient, pixel)): a model that
    runs forward AND backward holds THREE times the matrix (C3: 3.3 GB -> 9.9 GB) until the Ylm tensor is released or
    `release_ylm_packed(Ylm)` is called; RIME_ALM_PACKED=0 switches the copies off (kernels that split on the fly, ~25 %
    slower at the C3 shape).  A copy is only made while the device has that much free memory + 1 GB, counting the blocks the
    caching allocator holds but does not use; a refusal is re-examined on later calls (not remembered for the life of the
    tensor).  The pack kernel runs on the stream that is current at first use; an event recorded behind it is kept with
    the buffer, every later use on ANOTHER stream waits for it first and is recorded on the buffer (`record_stream`), so
    that releasing the copy while that stream still reads it cannot hand the block to a new owner.
    """
    if not ALM_PACKED or Y.dtype != torch.complex64 or ys <= 0 or Y.numel() * 8 < ALM_PACKED_MIN_BYTES:
        return None
    tag = ylm_packed_state(Ylm)
    if tag is None or tag[0] != Ylm._version or tag[1] != ys or tag[2] != Y.data_ptr():
        tag = (Ylm._version, ys, Y.data_ptr(), {})
        key = id(Ylm)
        try:
            import weakref
            _YLM_PACKED[key] = (weakref.ref(Ylm, lambda r, key=key: _YLM_PACKED.pop(key, None) if key in _YLM_PACKED and _YLM_PACKED[key][0] is r else None), tag)
        except TypeError:
            return None
    entry = tag[3].get(direction)
    stream = torch.cuda.current_stream(Y.device)
    if entry is None or entry is False:
        Nc, Npix = Y.shape
        nbytes = int(lib.rime_alm2pix_packed_bytes(Nc, Npix, direction))
        free, _ = torch.cuda.mem_get_info(Y.device)
        # blocks the caching allocator has reserved but not handed out are usable too
        free += torch.cuda.memory_reserved(Y.device) - torch.cuda.memory_allocated(Y.device)
        if nbytes == 0 or free < nbytes + (1 << 30):
            tag[3][direction] = False                    # no room NOW: asked again on the next call (two cheap queries)
            return None
        buf = torch.empty(nbytes, dtype=torch.uint8, device=Y.device)
        check(lib.rime_alm2pix_pack(_ptr(torch.view_as_real(Y)), ys, Nc, Npix, direction, _ptr(buf), _stream()), 'rime_alm2pix_pack')
        ready = torch.cuda.Event()
        ready.record(stream)
        entry = tag[3][direction] = (buf, ready, stream.cuda_stream)
    buf, ready, made_on = entry
    if made_on != stream.cuda_stream:
        stream.wait_event(ready)                         # packed on another stream: order this use behind the pack kernel
        buf.record_stream(stream)                        # ... and keep the block from being re-used under this stream's kernels
    return buf


def release_ylm_packed(Ylm):
    """drop the cached packed copies of `Ylm` (two buffers of its size; see _ylm_packed); they are rebuilt on the next use"""
    if ylm_packed_state(Ylm) is not None:
        _YLM_PACKED.pop(id(Ylm), None)


class _Alm2Pix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alm, Ylm):
        _require_cuda(alm, Ylm)
        assert alm.is_complex() and Ylm.is_complex()
        lead = alm.shape[:-1]
        Nc, Npix = Ylm.shape
        assert alm.shape[-1] == Nc
        a = alm.detach().contiguous()
        Y = Ylm.detach().contiguous()
        code, rdt = _real_dtype(a)
        assert Y.dtype == a.dtype, 'alm %s vs Ylm %s' % (a.dtype, Y.dtype)
        R = int(np.prod(lead)) if len(lead) else 1
        out = torch.empty(lead + (Npix,), dtype=rdt, device=a.device)
        ys = _ylm_scale(Ylm) if (rdt == torch.float32 and ALM_SPLIT_F16) else 0.0
        nbytes = lib.rime_alm2pix_fwd_workspace(code, R, Nc, Npix) if ys > 0 else 0
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=a.device)
        packed = _ylm_packed(Ylm, Y, ys, 0) if ys > 0 else None
        if packed is not None:
            rc = lib.rime_alm2pix_fwd_packed(_ptr(torch.view_as_real(a)), _ptr(packed), ys, R, Nc, Npix, _ptr(out),
                                             _ptr(ws), ws.numel(), _stream())
            check(rc, 'rime_alm2pix_fwd_packed')
        else:
            rc = lib.rime_alm2pix_fwd(code, _ptr(torch.view_as_real(a)), _ptr(torch.view_as_real(Y)), ys,
                                      R, Nc, Npix, _ptr(out), _ptr(ws), ws.numel(), _stream())
            check(rc, 'rime_alm2pix_fwd')
        # the packed-copy cache lives on the CALLER's tensor object: a weak reference finds it again in the backward without
        # keeping that object (and its two packed buffers) alive through the graph
        import weakref
        ctx.Y, ctx.shape, ctx.dtype, ctx.ys, ctx.Ylm = Y, tuple(alm.shape), alm.dtype, ys, weakref.ref(Ylm)
        return out

    @staticmethod
    def backward(ctx, gout):
        Y = ctx.Y
        g = gout.contiguous()
        code, _ = _real_dtype(g)
        Nc, Npix = Y.shape
        ga = torch.empty(ctx.shape, dtype=ctx.dtype, device=g.device)
        R = int(np.prod(ctx.shape[:-1])) if len(ctx.shape) > 1 else 1
        nbytes = lib.rime_alm2pix_bwd_workspace(code, R, Nc, Npix)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=g.device)
        owner = ctx.Ylm()
        packed = _ylm_packed(owner, Y, ctx.ys, 1) if (ctx.ys > 0 and owner is not None) else None
        if packed is not None:
            rc = lib.rime_alm2pix_bwd_packed(_ptr(g), _ptr(packed), ctx.ys, R, Nc, Npix,
                                             _ptr(torch.view_as_real(ga)), _ptr(ws), ws.numel(), _stream())
            check(rc, 'rime_alm2pix_bwd_packed')
        else:
            rc = lib.rime_alm2pix_bwd(code, _ptr(g), _ptr(torch.view_as_real(Y)), ctx.ys, R, Nc, Npix,
                                      _ptr(torch.view_as_real(ga)), _ptr(ws), ws.numel(), _stream())
            check(rc, 'rime_alm2pix_bwd')
        return ga, None


def alm2pix(alm, Ylm):
    """Re(alm @ Ylm): alm (..., Ncoeff) complex, Ylm (Ncoeff, Npix) complex -> (..., Npix) real."""
    return _Alm2Pix.apply(alm, Ylm)
