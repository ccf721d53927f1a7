"""
bayeslim_amd -- MI355X-native drop-in for the RIME visibility-synthesis hot path of BayesLIM
(rime_model.RIME.forward + backward).  The arithmetic runs in hand-written HIP kernels for
gfx950 behind the C ABI of include/rime_hip.h; importing this package loads that library and
fails loudly if it has not been built.
"""
from . import _lib, ops            # noqa: F401  (loads librime_hip.so)

__version__ = '0.1.0'
