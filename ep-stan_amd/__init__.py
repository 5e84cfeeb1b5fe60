"""epstan_amd -- MI355X-native drop-in for ep-stan's data-parallel EP inner loop.

Mirrors the hot-path surface of /root/reference/epstan: `method.Master`,
`method.Worker`, `util.invert_normal_params`, `util.olse`.  All numerical work
runs in hand-written HIP kernels (libepx.so, include/epx.h) reached through
ctypes; there is no CPU fallback.
"""

__all__ = ['method', 'util', 'engine', 'models', 'seeds']
__version__ = '0.1.0'
