"""
Gain application at the headline workload's visibility shape (C4: 8128 baselines x 8 times x 256 channels,
complex64; and a 2x2 full-pol case): fused HIP kernels (ops.apply_cal) against the torch composition the
reference runs (index_select x2, conj, products / einsum), forward + backward, HIP-event timed.
Prints algorithmic HBM bytes / time as a fraction of 8 TB/s.   usage: python tools/bench_cal.py
"""
import itertools
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayeslim_amd import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def torch_apply(vis, gains, i1, i2, Np, diag):
    g1, g2 = gains.index_select(2, i1), gains.index_select(2, i2)
    if Np == 1:
        return g1 * g2.conj() * vis
    if diag:
        G = g1 * g2.conj()
        out = torch.zeros_like(vis)
        out[0, 0] = G[0, 0] * vis[0, 0]
        out[1, 1] = G[1, 1] * vis[1, 1]
        return out
    return torch.einsum('ab...,bc...,dc...->ad...', g1, vis, g2.conj())


def main():
    dev = torch.device('cuda:0')
    Nant, Nt, Nf = 128, 8, 256
    pairs = list(itertools.combinations(range(Nant), 2))
    i1 = torch.tensor([p[0] for p in pairs], device=dev)
    i2 = torch.tensor([p[1] for p in pairs], device=dev)
    a1, a2 = i1.int(), i2.int()
    for Np, diag in ((1, False), (2, True), (2, False)):
        vis = torch.randn(Np, Np, len(pairs), Nt, Nf, dtype=torch.complex64, device=dev, requires_grad=True)
        gains = torch.randn(Np, Np, Nant, Nt, Nf, dtype=torch.complex64, device=dev, requires_grad=True)
        cot = torch.randn_like(vis.detach())

        def step(f):
            vis.grad = gains.grad = None
            out = f()
            out.backward(cot)

        t_f = timed(lambda: ops.apply_cal(vis.detach(), gains.detach(), a1, a2, diag))
        t_fb = timed(lambda: step(lambda: ops.apply_cal(vis, gains, a1, a2, diag)))
        r_f = timed(lambda: torch_apply(vis.detach(), gains.detach(), i1, i2, Np, diag))
        r_fb = timed(lambda: step(lambda: torch_apply(vis, gains, i1, i2, Np, diag)))
        nbytes = vis.numel() * 8
        print('Np=%d diag=%d vis %.0f MB | fused fwd %.3f ms (%.2f of 8 TB/s on 2x vis bytes)  fwd+bwd %.3f ms | '
              'torch fwd %.3f ms  fwd+bwd %.3f ms | speedup fwd %.1fx  fwd+bwd %.1fx'
              % (Np, diag, nbytes / 1e6, t_f, 2 * nbytes / (t_f * 1e-3) / 8e12, t_fb, r_f, r_fb, r_f / t_f, r_fb / t_fb), flush=True)


if __name__ == '__main__':
    main()
