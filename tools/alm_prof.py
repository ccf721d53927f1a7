import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayeslim_amd import ops
R, Nc, Npix = 128, 8385, 49152
a = torch.randn(R, Nc, dtype=torch.complex64, device='cuda', requires_grad=True)
Y = torch.randn(Nc, Npix, dtype=torch.complex64, device='cuda')
for _ in range(6):
    out = ops.alm2pix(a, Y)
    out.backward(torch.ones_like(out)); a.grad = None
torch.cuda.synchronize()
