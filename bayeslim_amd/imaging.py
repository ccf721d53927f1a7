"""
Map-making arithmetic of the reference's imaging.py on the fused fringe
kernels: the imaging matrix A = conj(fringe) * beam of VisMapper.build_A (imaging.py:251-296) is never
materialised -- (Nbl, Nf, Npix) complex, C4: 1.6 TB per time step.  `make_map` is the adjoint of the
RIME fringe sum (the backward kernels, including the antenna-factored matrix-core path), `compute_Am`
its forward, `compute_Pm` their composition.

The VisMapper container (time / baseline / channel selection, normalisation bookkeeping, PSF
contraction modes) is out of scope; these are the functions it calls per time step
(imaging.py:717-736, 755-774, 777-815), taking an ops.FringeGeometry instead of A.
"""
import torch

from . import ops, telescope_model


def geometry(blvecs, zen, az, freqs, antpos=None, bl_ants=None):
    """FringeGeometry of ONE time step at pointing angles (zen, az) [deg] (all pixels kept: apply the
    FoV cut before, as build_A does); antpos / bl_ants enable the antenna-factored kernels"""
    zen = torch.as_tensor(zen, dtype=torch.float64, device=blvecs.device)
    az = torch.as_tensor(az, dtype=torch.float64, device=blvecs.device)
    P = zen.numel()
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(1, 3, Ps, dtype=torch.float64, device=blvecs.device)
    sdir[0, :, :P] = telescope_model.pointing_vectors(zen, az)
    geom = ops.FringeGeometry(blvecs, sdir, freqs, npix=[P], antpos=antpos, bl_ants=bl_ants)
    geom.P = P
    return geom


def _pad(x, Ps):
    return x if x.shape[-1] == Ps else torch.nn.functional.pad(x, (0, Ps - x.shape[-1]))


def make_map(v, w, geom, beam=None):
    """
    dirty map  m[..., f, p] = beam[f, p] * Re sum_b conj(F[b, f, p]) (v w)[..., b, f]   (imaging.py:717-736
    with A = conj(fringe) * beam).  v (..., Nbl, Nf) complex, w (Nbl, Nf) real -> (..., Nf, P) real.
    """
    lead = v.shape[:-2]
    g = (v * w).reshape((-1,) + tuple(v.shape[-2:]))               # (Nmaps, Nbl, Nf)
    out = ops.fringe_adjoint(g[:, :, None, :], geom)[0, 0]         # (Nmaps, Nf, Ps)
    out = out[..., :geom.P]
    if beam is not None:
        out = out * beam
    return out.reshape(lead + tuple(out.shape[-2:]))


def compute_Am(geom, m, beam=None):
    """
    conj(A) @ m = sum_p F[b, f, p] beam[f, p] m[..., f, p]: the RIME forward of a map (imaging.py:755-774).
    m (..., Nf, P) real -> (..., Nbl, Nf) complex.
    """
    lead = m.shape[:-2]
    x = m if beam is None else m * beam
    x = _pad(x.reshape((-1,) + tuple(m.shape[-2:])), geom.Pstride)  # (Nmaps, Nf, Ps)
    vis = ops.fringe_sum(x[None, None].contiguous(), geom)          # psky (1, 1, Nmaps, Nf, Ps) -> (Nmaps, Nbl, 1, Nf)
    return vis[:, :, 0].reshape(lead + (geom.Nbl, geom.Nf))


def compute_Pm(geom, w, m, beam=None, D=None):
    """P m = D A^T w (conj(A) m) (imaging.py:777-815): (..., Nf, P) real"""
    Pm = make_map(compute_Am(geom, m, beam), w, geom, beam)
    return Pm if D is None else Pm * D
