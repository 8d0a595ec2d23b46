"""ctypes binding of libag2v.so (C ABI declared in include/ag2v.h).

The product path has no fallback: if the library is missing, or a kernel returns an error code, a RuntimeError is raised
(the reference's convention is "raise, the training loop logs and continues", scripts/train.py:466-468).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AG2V_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libag2v.so')   # AG2V_LIB_PATH: another build, for A/B runs

_P = c_void_p
_SIGS = {
    # name: (restype, [argtypes])
    'ag2v_set_conv_precision': (c_int, [c_int]),
    'ag2v_get_conv_precision': (c_int, []),
    'ag2v_last_conv_kernel': (c_char_p, []),
    'ag2v_set_bf16x3_min_rows': (c_int, [c_int]),
    'ag2v_get_bf16x3_min_rows': (c_int, []),
    'ag2v_halo_reload_env': (c_int, []),
    'ag2v_halo_min_tiles': (c_int, []),
    'ag2v_conv2d_workspace_floats': (c_size_t, [c_int] * 4),
    'ag2v_conv2d_fwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 9 + [_P, _P]),
    'ag2v_conv2d_dgrad': (c_int, [_P, _P, _P] + [c_int] * 9 + [_P, _P]),
    'ag2v_conv2d_fwd_act': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 10 + [_P, _P]),
    'ag2v_conv2d_dgrad_masked': (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P, _P]),
    'ag2v_conv2d_dgrad_acc': (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [_P, _P]),
    'ag2v_conv2d_wgrad': (c_int, [_P, _P, _P] + [c_int] * 9 + [_P]),
    'ag2v_conv2d_wgrad_acc': (c_int, [_P, _P, _P] + [c_int] * 10 + [_P]),
    'ag2v_conv2d_fwd_ex': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 10 + [_P, _P, _P]),
    'ag2v_conv2d_dgrad_workspace_floats': (c_size_t, [c_int] * 9),
    'ag2v_conv2d_dgrad_ex': (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [_P, c_size_t, _P, _P]),
    'ag2v_conv2d_wgrad_ex': (c_int, [_P, _P, _P] + [c_int] * 11 + [_P, _P]),
    'ag2v_conv2d_wgrad_det_workspace_floats': (c_size_t, [c_int] * 10 + [_P]),
    'ag2v_conv2d_split_input_ok': (c_int, [c_int] * 12),
    'ag2v_conv2d_fwd_mtile_ok': (c_int, [c_int] * 14),
    'ag2v_conv2d_split_dy_ok': (c_int, [c_int] * 11),
    'ag2v_channel_sum': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ag2v_channel_sum_acc': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    'ag2v_channel_sum_acc_s': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'ag2v_bn_workspace_floats': (c_size_t, [c_int, c_int, c_int]),
    'ag2v_bn_stats': (c_int, [_P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    'ag2v_norm_apply': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    'ag2v_norm_bwd': (c_int, [c_int] + [_P] * 11 + [c_int, c_int, c_int, c_float, _P, _P]),
    'ag2v_bn_stats_g': (c_int, [_P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    'ag2v_norm_apply_g': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    'ag2v_norm_apply_gs': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'ag2v_norm_bwd_g': (c_int, [c_int] + [_P] * 11 + [c_int, c_int, c_int, c_int, c_float, _P, _P]),
    'ag2v_norm_bwd_pyr': (c_int, [_P, _P, _P, _P, c_int] + [_P] * 7 + [c_int] * 5 + [c_float, _P, _P]),
    'ag2v_norm_bwd_pyr_p': (c_int, [_P, _P, _P, _P, c_int] + [_P] * 7 + [c_int] * 5 + [c_float, _P, c_int, c_float, _P]),
    'ag2v_norm_bwd_gs': (c_int, [c_int] + [_P] * 12 + [c_int, c_int, c_int, c_int, c_float, _P, c_int, _P]),
    'ag2v_norm_bwd_gs_p': (c_int, [c_int] + [_P] * 12 + [c_int, c_int, c_int, c_int, c_float, _P, c_int, c_int, c_float, _P]),
    'ag2v_norm_bwd_reduce_g': (c_int, [c_int] + [_P] * 7 + [c_int, c_int, c_int, c_int, c_float, _P, _P]),
    'ag2v_norm_bwd_apply_g': (c_int, [c_int] + [_P] * 8 + [c_float, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    'ag2v_spade_pair_supported': (c_int, [c_int, c_int, c_int]),
    'ag2v_spade_pair_workspace_floats': (c_size_t, [c_int, c_int, c_int]),
    'ag2v_spade_pair_stats': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    'ag2v_spade_pair_apply': (c_int, [_P] * 7 + [c_int] * 6 + [c_float, c_float, _P]),
    'ag2v_spade_pair_apply_s': (c_int, [_P] * 7 + [c_int] * 6 + [c_float, c_float, c_int, c_int, _P]),
    'ag2v_spade_pair_bwd': (c_int, [_P] * 13 + [c_int] * 6 + [c_float, c_float, _P, _P]),
    'ag2v_spade_pair_bwd_s': (c_int, [_P] * 13 + [c_int] * 6 + [c_float, c_float, _P, c_int, _P]),
    'ag2v_spade_pair_bwd_sp': (c_int, [_P] * 13 + [c_int] * 6 + [c_float, c_float, _P, c_int, c_int, c_float, _P]),
    'ag2v_instnorm_lrelu_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    'ag2v_instnorm_lrelu_bwd': (c_int, [_P] * 6 + [c_int, c_int, c_float, _P]),
    'ag2v_layout_render_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P]),
    'ag2v_layout_render_bwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P]),
    'ag2v_layout_conv_fwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 9 + [_P]),
    'ag2v_layout_conv_img_fwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 9 + [_P, _P, _P, c_int, _P]),
    'ag2v_layout_conv_img_fwd_stats': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 9 + [_P, _P, _P, c_int, _P, c_int, _P]),
    'ag2v_layout_conv_stats_supported': (c_int, [c_int] * 11),
    'ag2v_layout_conv_stats_floats': (c_size_t, [c_int] * 3),
    'ag2v_bn_finalize_partials': (c_int, [_P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P]),
    'ag2v_layout_conv_bwd': (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P]),
    'ag2v_flow_warp_fwd': (c_int, [_P, _P, _P] + [c_int] * 4 + [_P]),
    'ag2v_flow_warp_bwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 4 + [_P]),
    'ag2v_avgpool3s2_fwd': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ag2v_avgpool3s2_bwd': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ag2v_maxpool2_fwd': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ag2v_maxpool2_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    'ag2v_nearest_resize_fwd': (c_int, [_P, _P] + [c_int] * 5 + [_P]),
    'ag2v_nearest_resize_bwd': (c_int, [_P, _P] + [c_int] * 5 + [_P]),
    'ag2v_norm_apply_half': (c_int, [_P] * 7 + [c_int] * 5 + [c_float, c_int, _P]),
    'ag2v_nearest_pyramid_bwd': (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P]),
    'ag2v_triple_gather_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P]),
    'ag2v_triple_gather_bwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P]),
    'ag2v_triple_pool_fwd': (c_int, [_P] * 5 + [c_int] * 5 + [_P]),
    'ag2v_triple_pool_bwd': (c_int, [_P] * 6 + [c_int] * 5 + [_P]),
    'ag2v_gemm_f32': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_longlong,
                              c_int, c_int, c_int, _P]),
    'ag2v_relu_bwd': (c_int, [_P, _P, _P, c_longlong, _P]),
    'ag2v_col_sum': (c_int, [_P, _P, c_int, c_int, _P]),
    'ag2v_embedding_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'ag2v_embedding_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'ag2v_act_fwd': (c_int, [c_int, _P, _P, c_longlong, c_float, _P]),
    'ag2v_act_bwd': (c_int, [c_int, _P, _P, _P, c_longlong, c_float, _P]),
    'ag2v_reduce_workspace_floats': (c_size_t, []),
    'ag2v_reduce_fwd': (c_int, [c_int, _P, _P, c_longlong, c_float, _P, _P, _P]),
    'ag2v_reduce_bwd': (c_int, [c_int, _P, _P, c_longlong, _P, c_float, _P, _P]),
    'ag2v_adam_step': (c_int, [_P, _P, _P, _P, c_longlong, c_float, c_float, c_float, c_float, c_int, _P]),
    'ag2v_spectral_workspace_floats': (c_size_t, [c_int]),
    'ag2v_spectral_norm_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_float, c_int, _P, _P, _P, _P]),
    'ag2v_spectral_norm_fwd_save': (c_int, [_P, _P, _P, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P, _P]),
    'ag2v_spectral_norm_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    'ag2v_spectral_norm_bwd_acc': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, c_int, _P, _P]),
    'ag2v_spectral_sigmas': (c_int, [_P, _P, _P, c_int, c_int, c_float, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'ag2v_sn_desc_bytes': (c_size_t, []),
    'ag2v_spectral_sigmas_multi': (c_int, [_P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'ag2v_graph_op_bytes': (c_size_t, []),
    'ag2v_graph_program_grid': (c_int, []),
    'ag2v_graph_program_max_edges': (c_int, []),
    'ag2v_graph_program_run': (c_int, [_P, c_int, _P, c_int, _P]),
    'ag2v_sn_group_combine': (c_int, [_P, _P, _P, c_int, c_longlong, _P, c_int, _P, _P]),
    'ag2v_spectral_sigmas_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
}
EXPORTS = tuple(sorted(_SIGS))


class ConvOpts(ctypes.Structure):
    """ag2v_conv_opts (include/ag2v.h): per-call options of the *_ex convolution entry points."""
    _fields_ = [('precision', c_int), ('min_rows', c_int), ('gscale', c_void_p), ('group_n', c_int), ('kernel', c_char_p),
                ('in_split', c_int), ('dy_split', c_int), ('out_split', c_int), ('deterministic', c_int), ('det_workspace', c_void_p),
                ('det_workspace_floats', ctypes.c_size_t), ('out_nstride', ctypes.c_longlong), ('out_mstride', ctypes.c_longlong),
                ('out_chan_sums', c_void_p), ('chan_sums_done', c_int)]

_lib = None


def load():
    """dlopen libag2v.so once; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "ag2video_amd: %s not found -- build the HIP kernels first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C ag2video_amd/csrc). "
            "There is no CPU fallback." % LIB_PATH)
    # torch first: its bundled libamdhip64.so.7 must be the HIP runtime of the process. Loaded in the other order the
    # /opt/rocm runtime this library's RUNPATH names wins the soname, torch runs on a runtime it was not built
    # against, and the first kernel launch fails with a HIP launch error.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_ERR = {-1: 'AG2V_EINVAL (rejected argument)', -2: 'AG2V_EHIP (HIP launch error)'}


def call(name, *args):
    """Invoke an int-returning entry point; non-zero => RuntimeError."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (name, _ERR.get(rc, rc)))


def query(name, *args):
    return getattr(load(), name)(*args)
