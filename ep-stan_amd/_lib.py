"""ctypes binding of libepx.so (include/epx.h)."""

import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EPX_LIB', os.path.join(HERE, 'libepx.so'))   # EPX_LIB: diagnostic builds only

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)
c_int_p = ctypes.POINTER(ctypes.c_int)
COMM_ID_BYTES = 128          # EPX_COMM_ID_BYTES
OP_SUM, OP_MIN, OP_MAX = 0, 1, 2
# epx_host_allreduce_fn (include/epx.h): int fn(double *buf, long long n, int op, void *user)
HOST_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_double_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p)


class SamplerOpts(ctypes.Structure):
    """epx_sampler_opts (include/epx.h)."""
    _fields_ = [('chains', ctypes.c_int32), ('iter', ctypes.c_int32),
                ('warmup', ctypes.c_int32), ('thin', ctypes.c_int32),
                ('init', ctypes.c_int32), ('max_depth', ctypes.c_int32),
                ('layout', ctypes.c_int32), ('reserved', ctypes.c_int32)]


# every symbol include/epx.h declares, with its argument types
SIGNATURES = {
    'epx_last_error': (ctypes.c_char_p, []),
    'epx_device_count': (ctypes.c_int, [c_int_p]),
    'epx_device_synchronize': (ctypes.c_int, [ctypes.c_int]),
    'epx_model_dims': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_int_p, c_int_p]),
    'epx_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      c_int64_p, c_double_p, c_int32_p,
                                      ctypes.POINTER(ctypes.c_void_p)]),
    'epx_ctx_create_real': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           c_int64_p, c_double_p, c_double_p,
                                           ctypes.POINTER(ctypes.c_void_p)]),
    'epx_ctx_create_real_groups': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  c_int64_p, c_int32_p, c_int64_p, c_double_p, c_double_p,
                                                  ctypes.POINTER(ctypes.c_void_p)]),
    'epx_ctx_create_groups': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             c_int64_p, c_int32_p, c_int64_p, c_double_p, c_int32_p,
                                             ctypes.POINTER(ctypes.c_void_p)]),
    'epx_ctx_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'epx_set_prior': (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    'epx_set_sites': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p]),
    'epx_get_sites': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p]),
    'epx_set_site': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p]),
    'epx_get_site': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p]),
    'epx_set_global': (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    'epx_get_global': (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    'epx_cavity_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_uint8_p]),
    'epx_cavity_site': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p,
                                       c_double_p, c_double_p, c_uint8_p]),
    'epx_get_cavity': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p]),
    'epx_tilted_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_int64_p,
                                        ctypes.POINTER(SamplerOpts), ctypes.c_int, c_uint8_p,
                                        c_double_p, c_double_p]),
    'epx_moments_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p,
                                         ctypes.c_int, ctypes.c_int, c_uint8_p]),
    'epx_get_tilted': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, c_int_p]),
    'epx_get_draws': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p]),
    'epx_num_draws': (ctypes.c_int, [ctypes.c_void_p, c_int_p]),
    'epx_site_sums': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_void_p]),
    'epx_packed_len': (ctypes.c_int, [ctypes.c_void_p, c_int_p]),
    'epx_damped_trial': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, c_double_p, ctypes.c_void_p,
                                        c_int_p, c_int_p, c_int_p]),
    'epx_accept': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    'epx_global_moments': (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    'epx_force_pd': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, c_uint8_p]),
    'epx_get_adapt': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p]),
    'epx_logdensity_grad': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, c_double_p]),
    'epx_logdensity_grad_layout': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_int,
                                                  ctypes.POINTER(ctypes.c_double), c_double_p]),
    'epx_sample_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_int64_p,
                                        ctypes.POINTER(SamplerOpts), c_double_p, c_double_p]),
    'epx_nuts_transitions': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_int64_p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    'epx_last_layout': (ctypes.c_int, [ctypes.c_void_p]),
    'epx_mix_sums': (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    'epx_damp_sweep': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, ctypes.c_void_p,
                                      c_double_p, c_double_p, ctypes.c_double, c_double_p, c_double_p, ctypes.c_int,
                                      c_double_p]),
    'epx_set_piece_queue': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p]),
    'epx_set_trace': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'epx_get_trace': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_longlong]),
    'epx_last_segments': (ctypes.c_int, [ctypes.c_void_p]),
    'epx_comm_unique_id': (ctypes.c_int, [ctypes.c_void_p]),
    'epx_comm_init': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'epx_comm_init_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, HOST_ALLREDUCE_FN, ctypes.c_void_p]),
    'epx_comm_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'epx_comm_size': (ctypes.c_int, [ctypes.c_void_p, c_int_p, c_int_p]),
    'epx_comm_allreduce': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int]),
    'epx_comm_allgather': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p]),
    'epx_update_trial': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                        c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int,
                                        c_int_p, c_int_p, c_int64_p, c_double_p, c_double_p]),
    'epx_set_site_order': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    'epx_set_site_split': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'epx_last_split': (ctypes.c_int, [ctypes.c_void_p]),
    'epx_cu_count': (ctypes.c_int, [ctypes.c_void_p]),
    'epx_get_chain_stats': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p]),
    'epx_rng_probe': (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32,
                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_double_p]),
    'epx_invert_normal_params': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                                c_double_p, ctypes.c_int, c_int32_p]),
    'epx_olse': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int,
                                c_double_p, c_int32_p]),
}

_lib = None


class EpxError(RuntimeError):
    """An entry point of libepx.so returned an error."""


def load():
    """Load libepx.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EpxError(
                'libepx.so not found at {}: build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` '
                '(there is no CPU fallback)'.format(LIB_PATH))
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise EpxError(load().epx_last_error().decode('utf-8', 'replace'))


def dptr(a):
    """double* of a float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64
    return a.ctypes.data_as(c_double_p)


def device_synchronize(device=0):
    check(load().epx_device_synchronize(int(device)))


def device_count():
    n = ctypes.c_int(0)
    rc = load().epx_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0
