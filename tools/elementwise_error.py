"""
Error of the float32 visibilities RELATIVE TO EACH VISIBILITY ITSELF (not to the largest one) against the reference's float64
outputs, on the reference-generated fixtures the matrix-core kernels serve.   python tools/elementwise_error.py  (one GPU)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import test_rime_gpu as tr              # noqa: E402  (the fixtures' model builders)

torch.set_default_dtype(torch.float32)
for tag in ['hex37', 'rand70', 'rand128', 'rand150', 'hex128']:     # hex37 and hex128: the conjugate-pair kernels
    g = tr.load_golden('rime_%s_mini' % tag)
    rime, sky, beam = tr._c2_setup(None, g)
    v = rime().data.detach().cpu().numpy().astype(np.complex128)
    r = g['vis']
    e = np.abs(v - r) / np.abs(r)
    big = np.abs(r) > 0.05 * np.abs(r).max()
    print('%-8s max-norm %.2e | elementwise: median %.2e  p90 %.2e  p99 %.2e  max %.2e  (max over |V| > 0.05 max|V|: %.2e)  '
          'min|V| / max|V| %.1e' % (tag, np.abs(v - r).max() / np.abs(r).max(), np.median(e), np.quantile(e, .9),
                                    np.quantile(e, .99), e.max(), e[big].max(), np.abs(r).min() / np.abs(r).max()))
