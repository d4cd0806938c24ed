"""ctypes binding of libspr_hip.so (include/spr_hip.h).

The library is the product: there is no Python/NumPy fallback behind it.  ``load()``
raises ``RuntimeError`` when the shared object is missing or cannot be loaded, and
every call goes through ``check()`` which turns a negative status into a Python
exception carrying ``spr_last_error()``.

``import torch`` happens before the ``CDLL`` so that the HIP runtime already mapped
by PyTorch (SONAME libamdhip64.so.7) is the one the kernels' launch stubs bind to --
device pointers and streams handed over from torch tensors are only meaningful
inside that runtime instance.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPR_HIP_LIBRARY: load another build of the same library (the ASan host build of `make asan`, tests only)
LIB_PATH = os.environ.get('SPR_HIP_LIBRARY') or os.path.join(_HERE, 'libspr_hip.so')

SPR_ABI_VERSION = 5          # include/spr_hip.h: the value these prototypes were written for
SPR_MAX_M = 256
SPR_MAX_M_WIDE = 512
SPR_MAX_R = 128
SPR_MAX_R_STREAM = 256
SPR_MAX_R_WIDE = 1024

_i32, _i64, _u64, _sz = C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
_p = C.c_void_p
_dbl = C.c_double

# name -> (restype, argtypes); mirrors include/spr_hip.h one to one
PROTOTYPES = {
    'spr_abi_version': (C.c_int, []),
    'spr_last_error': (C.c_char_p, []),
    'spr_device_cus': (C.c_int, [C.POINTER(C.c_int)]),
    'spr_upload_bytes': (C.c_int, [_p, _p, _i64, _p]),
    'spr_download_bytes': (C.c_int, [_p, _p, _i64, _p, _u64, _p]),
    'spr_host_tridiag_vectors': (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i32]),
    'spr_host_eig_top': (C.c_int, [_p, _i32, _i32, _p, _p, _p, _p, _p]),
    'spr_host_svd_top': (C.c_int, [_p, _i32, _i32, _p, _p, _p, _p, _p]),
    'spr_stats_gram_workspace': (_sz, [_i32, _i32]),
    'spr_stats_gram_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _sz, _p]),
    'spr_stats_gram_finalize_f64': (C.c_int, [_i64, _i32, _i64, _i64, _i32, _p, _sz, _p, _p, _i32, _i32, _p]),
    'spr_stats_gram_shifted_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _p, _p, _sz, _p]),
    'spr_gram_shift_finish_f64': (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _i32, _p]),
    'spr_rowstats_workspace': (_sz, [_i32]),
    'spr_rowstats_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _p, _p, _sz, _p]),
    'spr_gram_cross_workspace': (_sz, [_i32, _i32]),
    'spr_gram_cross_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    'spr_gram_cross_pair_f64': (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    'spr_rowmean_stats_f64': (C.c_int, [_p, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    'spr_spectrum_max_m': (_i32, []),
    'spr_spectrum_f64': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'spr_gram_combine_f64': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    'spr_project_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _i32, _p]),
    'spr_project_norms_supported': (_i32, [_i32, _i32, _i64, _i64, _p, _i32]),
    'spr_project_norms_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _p, _p]),
    'spr_project_stream_norms_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _p, _p, _sz, _p]),
    'spr_project_stream_workspace': (_sz, [_i32, _i32, _i32]),
    'spr_project_stream_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _p, _sz, _p]),
    'spr_scale_rows_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _p, _p, _i64, _p]),
    'spr_unscale_f64': (C.c_int, [_p, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _p]),
    'spr_feature_minmax_workspace': (_sz, [_i32]),
    'spr_feature_minmax_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    'spr_feature_digit_hist_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _i32, _i32, _i32, _p, _p]),
    'spr_colsums_workspace': (_sz, [_i32, _i32]),
    'spr_colsums_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _p, _p, _sz, _p]),
    'spr_fill_feature_f64': (C.c_int, [_p, _i64, _i64, _i64, _i32, _p, _p]),
    'spr_reconstruct_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _i32, _p, _i64, _p]),
    'spr_field_unstage_f64': (C.c_int, [_p, _i32, _i32, _i64, _p, _i64, _p]),
    'spr_field_unstage_blocks_f64': (C.c_int, [_p, _i32, _i32, _i64, _p, _p, _i64, _p]),
    'spr_p2p_handle_bytes': (_sz, []),
    'spr_p2p_alloc': (C.c_int, [_sz, _i32, _p, _p]),
    'spr_p2p_free': (C.c_int, [_p]),
    'spr_p2p_open': (C.c_int, [_p, _p]),
    'spr_p2p_close': (C.c_int, [_p]),
    'spr_p2p_device_id': (C.c_int, [_p, _i32]),
    'spr_p2p_peer_access': (C.c_int, [_p, _p]),
    'spr_p2p_signal': (C.c_int, [_p, _u64, _p]),
    'spr_p2p_wait': (C.c_int, [_p, _u64, _p]),
    'spr_p2p_flags_set': (C.c_int, [_p, _i32, _u64, _p]),
    'spr_p2p_flags_wait': (C.c_int, [_p, _i32, _u64, _dbl, _p, _p]),
    'spr_p2p_copy': (C.c_int, [_p, _p, _i64, _p]),
    'spr_p2p_poison_bit': (_u64, []),
    'spr_field_gather_p2p': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i32, _p, _p, _u64, _dbl, _p, _u64, _p, _p, _p, _p, _u64]),
    'spr_field_gather_p2p_join': (C.c_int, [_p, _i32, _u64, _dbl, _p, _p]),
    'spr_field_gather_p2p_release': (C.c_int, [_p, _i32, _u64, _p]),
    'spr_qr_workspace': (_sz, [_i64]),
    'spr_qr_workspace_r': (_sz, [_i64, _i32]),
    'spr_qr_batch': (_i32, []),
    'spr_mask_rows_f64': (C.c_int, [_p, _i64, _i32, _i64, _p, _p]),
    'spr_qr_init_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    'spr_qr_init_norms_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    'spr_qr_step_f64': (C.c_int, [_i64, _i32, _i32, _p, _i32, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _i64, _dbl, _p, _sz, _p]),
    'spr_qr_steps_f64': (C.c_int, [_i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _dbl, _p, _sz, _p]),
    'spr_qr_epoch_supported': (_i32, [_i32, _i64, _p, _i32, _i64]),
    'spr_qr_epoch_max_directions': (_i32, [_i32]),
    'spr_qr_pool_workspace': (_sz, []),
    'spr_qr_pool_build': (C.c_int, [_p, _i64, _dbl, _p, _i64, _p, _p, _sz, _p]),
    'spr_qr_epoch_sweep_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _i64, _dbl, _p, _p, _p, _sz, _p]),
    'spr_qr_exclude_f64': (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _i32, _p, _i32, _dbl, _p]),
    'spr_qr_refresh_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _p, _p, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    'spr_measure_csr_f64': (C.c_int, [_p, _p, _p, _i32, _p, _i64, _i32, _i64, _i64, _p, _p, _i64, _i32, _p, _p, _p, _p]),
    'spr_solve_ols_f64': (C.c_int, [_p, _i32, _i32, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _p]),
    'spr_solve_ols_workspace': (_sz, [_i32, _i32, _i32]),
    'spr_solve_ols_wide_f64': (C.c_int, [_p, _i32, _i32, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    'spr_solve_pinv_f64': (C.c_int, [_p, _i32, _i32, _p, _i32, _p, _i32, _p, _i32, _dbl, _p, _p, _p, _p, _p]),
    'spr_solve_pinv_workspace': (_sz, [_i32, _i32]),
    'spr_solve_pinv_wide_f64': (C.c_int, [_p, _i32, _i32, _p, _i32, _p, _i32, _p, _i32, _dbl, _p, _p, _p, _p, _p, _sz, _p]),
    'spr_comm_unique_id_bytes': (_sz, []),
    'spr_comm_unique_id': (C.c_int, [_p]),
    'spr_comm_init': (C.c_int, [_p, _i32, _i32, _p]),
    'spr_comm_destroy': (C.c_int, [_p]),
    'spr_comm_info': (C.c_int, [_p, _p, _p]),
    'spr_comm_library': (C.c_char_p, []),
    'spr_allreduce_f64': (C.c_int, [_p, _p, _i64, _p]),
    'spr_allreduce_i64': (C.c_int, [_p, _p, _i64, _p]),
    'spr_allgather': (C.c_int, [_p, _p, _p, _i64, _p]),
    'spr_fit_gram_pass_buffer': (_sz, [_i32, _i32, _i32]),
    'spr_fit_gram_pass_workspace': (_sz, [_i32, _i32, _i64]),
    'spr_fit_gram_pass': (C.c_int, [_p, _p, _i32, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _sz, _p, _p, _p, _p, _p, _sz, _p]),
    'spr_synth_f64': (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _p, _i32, _i32, _dbl, _u64, _p]),
    'spr_synth_gather_f64': (C.c_int, [_p, _i32, _i64, _i32, _p, _i32, _i32, _dbl, _u64, _p, _p]),
}

_lib = None


class SprError(RuntimeError):
    """A libspr_hip.so call returned a negative status."""


PROTOTYPES['spr_project_x32_acc'] = (C.c_int, [_p, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _p, _i64, _p])

# f32-storage twins: identical argument lists (the typed pointer is a void* here)
for _f64, _x32 in (('spr_stats_gram_f64', 'spr_stats_gram_x32'), ('spr_stats_gram_shifted_f64', 'spr_stats_gram_shifted_x32'), ('spr_rowstats_f64', 'spr_rowstats_x32'),
                   ('spr_gram_cross_f64', 'spr_gram_cross_x32'), ('spr_gram_cross_pair_f64', 'spr_gram_cross_pair_x32'), ('spr_project_f64', 'spr_project_x32'),
                   ('spr_project_f64', 'spr_project_x32_f64out'),
                   ('spr_project_stream_f64', 'spr_project_stream_x32'),
                   ('spr_project_stream_f64', 'spr_project_stream_x32_f64out'),
                   ('spr_project_norms_f64', 'spr_project_norms_x32'),
                   ('spr_project_norms_f64', 'spr_project_norms_x32_f64out'),
                   ('spr_project_stream_norms_f64', 'spr_project_stream_norms_x32'),
                   ('spr_project_stream_norms_f64', 'spr_project_stream_norms_x32_f64out'),
                   ('spr_qr_init_norms_f64', 'spr_qr_init_norms_u32'),
                   ('spr_qr_epoch_sweep_f64', 'spr_qr_epoch_sweep_u32'),
                   ('spr_scale_rows_f64', 'spr_scale_rows_x32'), ('spr_feature_minmax_f64', 'spr_feature_minmax_x32'),
                   ('spr_colsums_f64', 'spr_colsums_x32'), ('spr_feature_digit_hist_f64', 'spr_feature_digit_hist_x32'),
                   ('spr_synth_f64', 'spr_synth_f32'), ('spr_reconstruct_f64', 'spr_reconstruct_u32'),
                   ('spr_mask_rows_f64', 'spr_mask_rows_u32'), ('spr_qr_init_f64', 'spr_qr_init_u32'),
                   ('spr_qr_refresh_f64', 'spr_qr_refresh_u32'), ('spr_measure_csr_f64', 'spr_measure_csr_u32')):
    PROTOTYPES[_x32] = PROTOTYPES[_f64]


def load():
    """Load (once) and return the ctypes handle; raise loudly if it cannot be done."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `make -C openmeasure_amd/csrc` '
            '(or __graft_entry__.build()). There is no CPU fallback for the SPR kernels.')
    import torch  # noqa: F401  -- maps PyTorch's HIP runtime first (see module docstring)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    have = lib.spr_abi_version()
    if have != SPR_ABI_VERSION:
        # a changed argument list under an unchanged name would pass shifted arguments -- wild pointers on the GPU
        raise RuntimeError(f'{LIB_PATH} reports ABI version {have}, this binding was written for {SPR_ABI_VERSION}: '
                           'rebuild the library (make -C openmeasure_amd/csrc) or use the matching package')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc == 0:
        return
    msg = load().spr_last_error().decode('utf-8', 'replace')
    kinds = {-1: ValueError, -2: NotImplementedError, -3: SprError, -4: MemoryError}
    raise kinds.get(rc, SprError)(f'{what or "libspr_hip"} failed (status {rc}): {msg}')
