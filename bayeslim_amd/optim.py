"""
The likelihood epilogue of the RIME path -- the part of the reference's
optim.py that touches the visibility tensor right after RIME.forward: the residual, the inverse
covariance weighting and the chi-square sum of LogProb.forward_chisq (optim.py:959-1030) with
apply_icov (optim.py:1836-1915), and the LogProb container that drives RIME from an optimiser (optim.py:385-1389:
minibatch iteration, main-parameter tensor, priors, closure) with the likelihood on the fused chi-square kernel, and the
Trainer loop around it (optim.py:1631-1833).  Optimisers, samplers, Hessians and the single-process DistributedLogProb (replaced by dist.py) are out of scope
(SURVEY.md section 2).

Only the diagonal inverse covariance (cov_axis=None) runs on the fused HIP kernel; the reference's
'bl' / 'time' / 'freq' / 'pix' branches reference an undefined name (`d`, optim.py:1899-1913) and
cannot run there either, and 'full' is a dense matrix product left to torch.
"""
import time

import numpy as np
import torch

from . import ops, utils
from .dataset import TensorData


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


# ---------------------------------------------------------------------------------------
# log-priors (optim.py:17-312)
# ---------------------------------------------------------------------------------------
class BaseLogPrior:
    """indexing / pre-function / device handling shared by the priors (optim.py:17-74)"""
    def __init__(self, index=None, func=None, fkwargs=None, attrs=None):
        self.index, self.func = index, func
        self.fkwargs = fkwargs if fkwargs is not None else {}
        self.attrs = attrs if attrs is not None else []

    def _index_func(self, params):
        if self.index is not None:
            params = params[self.index]
        if self.func is not None:
            params = self.func(params, **self.fkwargs)
        return params

    def forward(self, params):
        raise NotImplementedError

    def __call__(self, params):
        return self.forward(params)

    def push(self, device):
        if not isinstance(device, torch.dtype) and self.index is not None:
            self.index = tuple(torch.as_tensor(i, device=device) if isinstance(i, (torch.Tensor, np.ndarray)) else i
                               for i in self.index)
        for a in self.attrs:
            if hasattr(self, a):
                setattr(self, a, utils.push(getattr(self, a), device))


class LogUniformPrior(BaseLogPrior):
    """log of a uniform density between the bounds, -inf outside; the value stays attached to the graph of params
    (optim.py:77-131)"""
    def __init__(self, lower_bound, upper_bound, index=None, func=None, fkwargs=None):
        super().__init__(index, func, fkwargs, attrs=['lower_bound', 'upper_bound'])
        self.lower_bound, self.upper_bound = torch.as_tensor(lower_bound), torch.as_tensor(upper_bound)
        self.norm = torch.sum(torch.log(1 / (self.upper_bound - self.lower_bound)))

    def forward(self, params):
        params = self._index_func(params)
        outside = torch.abs(torch.sign(self.lower_bound - params) + torch.sign(self.upper_bound - params)) == 2
        if outside.any():
            return torch.sum(params) * -np.inf
        lp = torch.sum(params)
        return lp / lp * self.norm


class LogGaussPrior(BaseLogPrior):
    """log of a Gaussian density (diagonal or full covariance; one-sided variants) (optim.py:217-312)"""
    def __init__(self, mean, cov, diag_cov=True, side='both', density=True, index=None, func=None, fkwargs=None):
        super().__init__(index, func, fkwargs, attrs=['mean', 'icov'])
        self.mean = torch.atleast_1d(torch.as_tensor(mean))
        self.cov = torch.atleast_1d(torch.as_tensor(cov))
        self.diag_cov, self.side, self.density = diag_cov, side, density
        self.compute_icov()

    def compute_icov(self, **kwargs):
        if self.diag_cov:
            self.icov = 1.0 / self.cov
            self.logdet = torch.sum(torch.log(self.cov))
            self.ndim = self.cov.numel()
        else:
            self.icov = torch.linalg.pinv(self.cov, hermitian=True)
            self.logdet = torch.slogdet(self.cov).logabsdet
            self.ndim = len(self.cov)
        self.norm = 0.5 * (self.ndim * torch.log(torch.tensor(2 * np.pi)) + self.logdet)
        self.icov = self.icov.to(self.mean.device)

    def forward(self, params):
        res = self._index_func(params) - self.mean
        if self.side == 'upper':
            res = torch.where(res < 0, torch.zeros_like(res), res)
        elif self.side == 'lower':
            res = torch.where(res > 0, torch.zeros_like(res), res)
        if self.diag_cov:
            sq = (res * res.conj()).real if torch.is_complex(res) else res ** 2
            chisq = torch.sum(sq * self.icov)
        else:
            res = res.ravel()
            chisq = torch.sum(res @ self.icov.to(res.dtype) @ res.conj())
        out = -0.5 * chisq.real
        return out - self.norm if self.density else out


# ---------------------------------------------------------------------------------------
# LogProb (optim.py:385-1389)
# ---------------------------------------------------------------------------------------
_RESOL = {torch.float32: 1, torch.complex64: 1, torch.float64: 2, torch.complex128: 2}


class LogProb(utils.Module):
    """
    (Negative) log posterior of a forward model against minibatched target data: Gaussian likelihood
    -log L = (d - mu)^H Sigma^-1 (d - mu) + log-normalisation (complex circular; half of it otherwise) plus the
    log-priors attached to the model's modules (or given in prior_dict), evaluated once per pass at batch 0.
    Same constructor, batching protocol, main-parameter tensor, closure and gradient modifiers as the reference
    (optim.py:385-1389); a linear preconditioner `LM` may be any callable R^N -> R^N.  With a diagonal (or absent)
    inverse covariance the residual, weighting and sum run as ONE fused kernel pass over the prediction (ops.chisq).
    """
    def __init__(self, model, target, start_inp=None, cov_parameter=False, prior_dict=None, device=None,
                 compute='post', negate=True, grad_type='accumulate', complex_circular=True):
        super().__init__()
        self.model = model
        assert isinstance(target, torch.utils.data.Dataset)
        self.target, self.start_inp = target, start_inp
        if cov_parameter:
            raise NotImplementedError
        self.cov_parameter = cov_parameter
        self.device = device
        self.prior_dict = prior_dict
        self.compute, self.negate = compute, negate
        self.closure_eval = 0
        self.grad_type = grad_type
        self.complex_circular = complex_circular
        self.set_grad_mod()
        self.clear_prior_cache()
        self.set_main_params()
        names = [m.name for m in self.model.modules() if hasattr(m, 'name')]
        if len(names) != len(set(names)):
            print("Warning: overlapping module names in model could lead to conflicts in prior evaluation")

    # ---- one flat parameter tensor scattered to the modules before every forward (optim.py:485-910)
    def set_main_params(self, model_params=None, LM=None, set_p0=False):
        """
        model_params: list of 'dotted.name', ('dotted.name', index) or ('dotted.name', index, short_name), the
        index a tuple or a LIST of tuples (several pieces of one tensor).  The selected values are collected
        into self.main_params (a Parameter); afterwards the modules' tensors are non-leaf views rebuilt from it
        before each evaluation.  None removes the main tensor and makes the modules' tensors Parameters again.
        """
        if getattr(self, '_main_names', None) is not None:
            for pname in self._main_names.values():
                self.model.set_param(pname)
        self.main_params = None
        self.main_p0 = None
        self._main_indices = self._main_shapes = self._main_devices = self._main_index = None
        self._main_names = self._main_dtypes = self._main_dtype = None
        self._main_LM, self._main_set_p0, self._main_N = LM, set_p0, None
        if model_params is None:
            return
        N = 0
        self._main_indices, self._main_shapes, self._main_devices = {}, {}, {}
        self._main_index, self._main_names, self._main_dtypes = {}, {}, {}
        for param in model_params:
            if isinstance(param, str):
                idx, name = None, param
            elif len(param) == 2:
                (param, idx), name = param, param[0]
            else:
                param, idx, name = param
            ten = self.model[param]
            as_index = lambda ix: tuple(utils._idx2ten(i, device=ten.device) for i in ix)
            if idx is None or not isinstance(idx, list):
                idx = None if idx is None else as_index(idx)
                p = ten.detach() if idx is None else ten[idx].detach()
                shape, indices = p.shape, slice(N, N + p.numel())
                N += p.numel()
            else:
                idx = [as_index(ix) for ix in idx]
                shape, indices = [], []
                for ix in idx:
                    p = ten[ix].detach()
                    shape.append(p.shape)
                    indices.append(slice(N, N + p.numel()))
                    N += p.numel()
            self._main_indices[name], self._main_shapes[name], self._main_devices[name] = indices, shape, p.device
            self._main_index[name], self._main_names[name], self._main_dtypes[name] = idx, param, p.dtype
        self._main_N = N
        for i, dt in enumerate(self._main_dtypes.values()):               # the widest dtype holds them all
            if i == 0 or _RESOL[dt] > _RESOL[self._main_dtype] or (dt.is_complex and not self._main_dtype.is_complex
                                                                    and _RESOL[dt] >= _RESOL[self._main_dtype]):
                self._main_dtype = dt
        self.collect_main_params()
        self.send_main_params()

    def _pieces(self, k):
        inds, idx, shape = self._main_indices[k], self._main_index[k], self._main_shapes[k]
        return zip(inds, idx, shape) if isinstance(inds, list) else [(inds, idx, shape)]

    def collect_main_params(self, inplace=True):
        """gather the modules' current values into the flat tensor (optim.py:760-801)"""
        if not self._main_indices:
            return None
        params = torch.zeros(self._main_N, dtype=self._main_dtype, device=self.device)
        for k, name in self._main_names.items():
            for inds, ix, _ in self._pieces(k):
                v = self.model[name] if ix is None else self.model[name][ix]
                params[inds] = v.detach().to(self.device).to(self._main_dtype).flatten()
        if not inplace:
            return params
        if self._main_set_p0:
            self.main_p0, self.main_params = params, torch.nn.Parameter(torch.zeros_like(params))
        else:
            self.main_p0, self.main_params = None, torch.nn.Parameter(params)
        self.send_main_params()

    def send_main_params(self, main_params=None, fill=None, main_p0=None):
        """scatter (LM(main_params) + main_p0) to the modules as graph tensors (optim.py:803-910)"""
        main_params = main_params if main_params is not None else self.main_params
        main_p0 = main_p0 if main_p0 is not None else self.main_p0
        if main_params is None:
            return
        if self._main_LM is not None:
            main_params = self._main_LM(main_params)
        if main_p0 is not None:
            main_params = main_params + main_p0
        for k, pname in self._main_names.items():
            dev, dt = self._main_devices[k], self._main_dtypes[k]
            for i, (inds, ix, shape) in enumerate(self._pieces(k)):
                value = main_params[inds].reshape(shape)
                if not utils.check_devices(value.device, dev):
                    value = value.to(dev)
                if value.dtype != dt:
                    value = value.real.to(dt) if (value.is_complex() and not dt.is_complex) else value.to(dt)
                # as the reference (optim.py:900-906): the first piece of a tensor replaces its entries, every later
                # piece is ADDED to what the tensor holds -- meant for pieces layered over a `fill`; with fill=None the
                # later pieces grow by their value on every send.  Kept identical on purpose (drop-in), noted in DESIGN.md.
                utils.set_model_attr(self.model, pname, value, idx=ix, clobber_param=(i == 0), no_grad=False,
                                     fill=fill if i == 0 else None, add=(i != 0))

    def clear_graph_tensors(self):
        if self._main_names is not None:
            for pname in self._main_names.values():
                self.model[pname] = self.model[pname].detach()

    # ---- minibatches (optim.py:912-957)
    @property
    def Nbatch(self):
        return self.model.Nbatch if hasattr(self.model, 'Nbatch') else 1

    @property
    def batch_idx(self):
        return self.model.batch_idx if hasattr(self.model, 'batch_idx') else 0

    @batch_idx.setter
    def batch_idx(self, val):
        if hasattr(self.model, 'batch_idx'):
            self.model.batch_idx = val
        elif val > 0:
            raise ValueError("No attr batch_idx and requested idx > 0")

    def get_batch_data(self, idx=None):
        if idx is not None:
            self.batch_idx = idx
        return self.target[self.batch_idx], (None if self.start_inp is None else self.start_inp[self.batch_idx])

    # ---- likelihood, prior, posterior (optim.py:959-1189)
    def forward_chisq(self, idx=None, main_params=None, sum_chisq=True, **kwargs):
        """(chi-square, residual) of the forward model against this minibatch's target; the residual is None when
        the fused kernel produced the sum directly"""
        target, inp = self.get_batch_data(idx)
        data = target.get_data()
        icov, cov_axis = (target.get_icov(), target.cov_axis) if hasattr(target, 'icov') else (None, None)
        if self.batch_idx == 0:
            self.clear_prior_cache()
        main_params = main_params if main_params is not None else self.main_params
        if main_params is not None:
            self.send_main_params(main_params=main_params)
        prediction = self.model(inp, prior_cache=self.prior_cache)
        if isinstance(prediction, TensorData):
            prediction = prediction.data
        if not utils.check_devices(prediction.device, self.device):
            prediction = prediction.to(self.device)
        if sum_chisq and cov_axis is None and prediction.is_cuda and prediction.is_complex():
            d = data.to(prediction.device).expand(prediction.shape)
            ic = None if icov is None else icov.to(prediction.device).expand(prediction.shape)
            return ops.chisq(prediction, d, ic), None
        res = prediction - data
        chisq = apply_icov(res, icov, cov_axis)
        if sum_chisq:
            chisq = torch.sum(chisq)
        return (chisq.real if torch.is_complex(chisq) else chisq), res

    def forward_like(self, idx=None, main_params=None, **kwargs):
        chisq, _ = self.forward_chisq(idx, main_params=main_params)
        target, _ = self.get_batch_data()
        norm = 0
        if getattr(target, 'icov', None) is not None:
            logdet = target.cov_logdet.to(chisq.device) if isinstance(target.cov_logdet, torch.Tensor) else target.cov_logdet
            if self.complex_circular:            # L(z) = exp(-z^H Cz^-1 z) / (pi^n det Cz)
                norm = target.cov_ndim * np.log(np.pi) + logdet
            else:                                # L(x) = exp(-x^T Cx^-1 x / 2) / ((2 pi)^n det Cx)^(1/2)
                norm = 0.5 * (target.cov_ndim * np.log(2 * np.pi) + logdet)
        loglike = (-chisq if self.complex_circular else -0.5 * chisq) - norm
        return -loglike if self.negate else loglike

    def forward_prior(self, idx=None, main_params=None, **kwargs):
        if idx is not None:
            self.batch_idx = idx
        main_params = main_params if main_params is not None else self.main_params
        if self.compute == 'prior' and main_params is not None:
            self.send_main_params(main_params=main_params)
        if self.compute == 'prior' and self.batch_idx == 0:
            self.clear_prior_cache()
        logprior = torch.zeros(1, device=self.device)
        if self.prior_dict is not None:
            for key, pr in self.prior_dict.items():
                for p in (pr if isinstance(pr, (tuple, list)) else [pr]):
                    logprior = logprior + p(self.model[key])
        else:
            if len(self.prior_cache) == 0:
                for _, mod in self.model.named_modules():
                    if hasattr(mod, 'params') and hasattr(mod, 'eval_prior'):
                        mod.eval_prior(self.prior_cache)
            for k in self.prior_cache:
                logprior = logprior + self.prior_cache[k].to(logprior.device)
        return -logprior if self.negate else logprior

    def forward(self, idx=None, **kwargs):
        assert self.compute in ['post', 'like', 'prior']
        if idx is not None:
            self.batch_idx = idx
        prob = torch.zeros(1, device=self.device)
        if self.compute in ['post', 'like']:
            prob = self.forward_like(**kwargs)
        if self.compute in ['post', 'prior'] and self.batch_idx == 0:        # once per pass over the minibatches
            if self.compute == 'prior':
                for mod in self.modules():
                    if hasattr(mod, 'clear_graph_tensors'):
                        mod.clear_graph_tensors()
            prob = prob + self.forward_prior(**kwargs)
        return prob

    def __call__(self, idx=None, **kwargs):
        return self.forward(idx=idx, **kwargs)

    def closure(self, **kwargs):
        """evaluate, back-propagate, return the loss: every minibatch accumulated into .grad ('accumulate', loss
        averaged over batches) or the current one only ('stochastic') (optim.py:1191-1226)"""
        self.closure_eval += 1
        if torch.is_grad_enabled():
            self.zero_grad()
        if self.grad_type == 'accumulate':
            loss = 0
            for i in range(self.Nbatch):
                self.batch_idx = i
                out = self()
                if out.requires_grad:
                    out.backward(**kwargs)
                loss = loss + out.detach()
            loss = loss / self.Nbatch
            self.batch_idx = 0
        elif self.grad_type == 'stochastic':
            out = self()
            if out.requires_grad:
                out.backward(**kwargs)
            loss = out.detach()
        else:
            raise ValueError(self.grad_type)
        self.grad_modify()
        self.clear_prior_cache()
        return loss

    # ---- gradient modifiers (optim.py:1228-1309)
    def set_grad_mod(self, grad_mods=None, alpha=1.0):
        """grad_mods: [(dotted parameter name under self, {'mod_type': 'clamp' | 'replace' | 'isolate' | 'clip' | 'mult',
        'value': ..., 'index': ..., 'dim': ...}), ...]"""
        self.grad_mods, self.alpha = grad_mods, alpha

    def grad_modify(self):
        if self.grad_mods is None:
            return
        for param, mod in self.grad_mods:
            grad = self[param].grad
            if grad is None:
                continue
            idx, value, kind = mod.get('index', slice(None)), mod.get('value') * self.alpha, mod.get('mod_type')
            if kind == 'clamp':
                g = grad[idx]
                g[(g < -value) | (g > value)] = 0.0
                grad[idx] = g
            elif kind == 'replace':
                grad[idx] = value
            elif kind == 'isolate':
                dim, a = mod.get('dim', None), torch.abs(grad[idx])
                gmax = torch.max(a) if dim is None else torch.max(a, dim=dim, keepdims=True).values
                grad[idx] *= (a / gmax) ** value
            elif kind == 'clip':
                grad[idx] *= (torch.argsort(torch.abs(grad[idx]), dim=mod.get('dim'), descending=True) <= value)
            elif kind == 'mult':
                grad[idx] *= value

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
        for d in self.target.data:
            d.push(device)
        if self.start_inp is not None:
            for d in self.start_inp.data:
                d.push(device)

    def clear_prior_cache(self):
        self.prior_cache = {}


class Trainer:
    """
    The training loop around a LogProb (optim.py:1631-1833): `opt` a torch optimiser class (instantiated on
    prob.parameters()) or instance; every epoch runs prob.Nbatch optimiser steps in 'stochastic' mode, one in
    'accumulate' mode, each step calling prob.closure; losses, cumulative times and (optionally) the parameter
    history are kept.
    """
    def __init__(self, prob, opt=None, track=False, track_params=None):
        self.prob = prob
        self._epoch_loss, self._epoch_times = [], []
        self.track = track
        self.set_opt(opt)
        self.Nbatch = prob.Nbatch if prob.grad_type == 'stochastic' else 1
        self.chain = {}
        if track:
            self.init_chain(track_params)

    def init_chain(self, track_params=None):
        """names (attributes of prob) whose values are recorded before every step: given, else the main-parameter
        pieces, else every model parameter"""
        if track_params is not None:
            names = list(track_params)
        elif self.prob.main_params is not None:
            names = ['model.' + k for k in self.prob._main_names.values()]
        else:
            names = ['model.' + k for k, _ in self.prob.model.named_parameters()]
        self.chain = {k: [] for k in names}

    def set_opt(self, opt, *args, **kwargs):
        if opt is not None:
            self.opt = opt(self.prob.parameters(), *args, **kwargs) if isinstance(opt, type) else opt

    def train(self, Nepochs=1, Nreport=None):
        start = time.time()
        for epoch in range(Nepochs):
            t0 = time.time()
            if Nreport is not None and epoch > 0 and epoch % Nreport == 0:
                print("epoch {}, {:.1f} sec".format(epoch, time.time() - start))
            self.opt.zero_grad()
            L = 0
            for _ in range(self.Nbatch):
                if self.track:
                    for k in self.chain:
                        self.chain[k].append(self.prob[k].detach().clone())
                L = L + self.opt.step(self.prob.closure)
            self._epoch_loss.append(L / self.Nbatch)
            self._epoch_times.append((self._epoch_times[-1] if self._epoch_times else 0.0) + (time.time() - t0))
        return dict(duration=time.time() - start)

    def get_chain(self, name=None, idx=None):
        assert self.track
        pick = (lambda c: torch.stack(c)) if idx is None else (lambda c: c[idx])
        return pick(self.chain[name]) if name is not None else {k: pick(c) for k, c in self.chain.items()}

    def revert_chain(self, Nepochs):
        """step the tracked parameters back by Nepochs recorded states (the current state is not in the chain)"""
        if self.track and Nepochs > 0:
            for k in self.chain:
                with torch.no_grad():
                    self.prob[k] = self.chain[k][-Nepochs]
                    self.chain[k] = self.chain[k][:-Nepochs]
            self._epoch_loss = self._epoch_loss[:-Nepochs]
            self._epoch_times = self._epoch_times[:-Nepochs]
            self.prob.collect_main_params()

    @property
    def loss(self):
        return torch.as_tensor(self._epoch_loss)

    @property
    def times(self):
        return torch.as_tensor(self._epoch_times)
