"""
The likelihood epilogue of the RIME path -- the part of the reference's
optim.py that touches the visibility tensor right after RIME.forward: the residual, the inverse
covariance weighting and the chi-square sum of LogProb.forward_chisq (optim.py:959-1030) with
apply_icov (optim.py:1836-1915).  Optimisers, samplers and the LogProb container itself are out of
scope.

Only the diagonal inverse covariance (cov_axis=None) runs on the fused HIP kernel; the reference's
'bl' / 'time' / 'freq' / 'pix' branches reference an undefined name (`d`, optim.py:1899-1913) and
cannot run there either, and 'full' is a dense matrix product left to torch.
"""
import torch

from . import ops


def apply_icov(data, icov, cov_axis=None, mode='vis'):
    """data^dagger Sigma^-1 data, elementwise for cov_axis=None (optim.py:1889-1894)"""
    if cov_axis is None:
        out = data.conj() * data
        return out if icov is None else out * icov
    if cov_axis == 'full':
        return data.ravel().conj() @ icov @ data.ravel()
    raise NotImplementedError("cov_axis=%r: not runnable in the reference either (optim.py:1899-1913)" % (cov_axis,))


def forward_chisq(prediction, data=None, icov=None, cov_axis=None, sum_chisq=True):
    """
    chi-square of a model prediction against target data (optim.py:1019-1027): returns (chisq, res).
    With sum_chisq and a diagonal (or absent) icov on the GPU the residual, weighting and sum are one
    fused pass (ops.chisq); `res` is then None -- the reference returns it only for diagnostics.
    """
    pred = prediction.data if hasattr(prediction, 'data') and not isinstance(prediction, torch.Tensor) else prediction
    if sum_chisq and cov_axis is None and pred.is_cuda and pred.is_complex():
        return ops.chisq(pred, data, icov), None
    res = pred if data is None else pred - data
    chisq = apply_icov(res, icov, cov_axis)
    if sum_chisq:
        chisq = torch.sum(chisq)
    if torch.is_complex(chisq):
        chisq = chisq.real
    return chisq, res
