"""ctypes binding of libhitadv_hip.so (C ABI declared in include/hitadv.h).

This is the only place the shared library is touched.  There is NO fallback: if the
library is missing or a launch fails, the caller gets a RuntimeError.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HITADV_LIBRARY: another build of the same library (diagnostic builds of tools/: stamped kernels, -DHITADV_PORTABLE_HANDOFF)
LIB_PATH = os.environ.get("HITADV_LIBRARY") or os.path.join(_HERE, "libhitadv_hip.so")

_c = ctypes
_P, _I, _F, _L = _c.c_void_p, _c.c_int, _c.c_float, _c.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/hitadv.h
PROTOTYPES = {
    "hitadv_version": [],
    "hitadv_pairwise_sqdist": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "hitadv_nn_min": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "hitadv_nn_min_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "hitadv_knn_points": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "hitadv_knn_points_bwd": [_P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P],
    "hitadv_topk_rows": [_P, _L, _I, _I, _I, _P, _P, _P],
    "hitadv_deform_fwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "hitadv_deform_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_deform_bwd_scratch_floats": [_I, _I, _I],
    "hitadv_best_update": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "hitadv_adam_step": [_P, _P, _P, _P, _L, _F, _P, _P, _P, _P, _L, _F, _P, _P],
    "hitadv_copy": [_P, _P, _L, _P],
    "hitadv_transpose_small": [_P, _P, _I, _I, _I, _I, _P],
    "hitadv_fps_from_start": [_P, _P, _I, _I, _I, _P, _P],
    "hitadv_fps_pct": [_P, _P, _I, _I, _I, _P, _P],
    "hitadv_furthest_point_sampling": [_I, _I, _I, _P, _P, _P, _P],
    "hitadv_gather_points": [_I, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_gather_points_grad": [_I, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_query_ball_point": [_I, _I, _I, _F, _I, _P, _P, _P, _P],
    "hitadv_query_ball_point_victim": [_I, _I, _I, _F, _I, _I, _P, _P, _P, _P],
    "hitadv_group_points": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_group_points_grad": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_three_nn": [_I, _I, _I, _P, _P, _P, _P, _P],
    "hitadv_three_interpolate": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "hitadv_three_interpolate_grad": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "hitadv_regulariser_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "hitadv_regulariser_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "hitadv_regulariser_scratch_floats": [_I],
    "hitadv_regulariser_bwd_add": [_P] * 8 + [_I, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "hitadv_adam_step_sum": [_P, _P, _P, _P, _P, _c.c_int64, _F, _F, _F, _P, _P, _P, _P, _P, _c.c_int64, _F, _F, _F, _P, _P],
    "hitadv_adv_loss": [_I, _P, _P, _I, _I, _F, _P, _P, _P],
    "hitadv_iteration_head": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P],
    "hitadv_iteration_head_reg": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P,
                                  _P, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _I, _P],
    "hitadv_group_linear_max_supported": [_I, _I, _I],
    "hitadv_rows_linear_supported": [_I, _I],
    "hitadv_rows_linear": [_P, _P, _P, _L, _I, _I, _I, _P, _P, _P],
    "hitadv_group_add_relu_linear_supported": [_I, _I, _I, _I],
    "hitadv_group_add_relu_linear": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P],
    "hitadv_group_linear_max_fwd": [_P, _P, _P, _c.c_int64, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_group_linear_max_bwd": [_P, _P, _P, _P, _c.c_int64, _I, _I, _I, _P, _P, _P],
    "hitadv_group_linear_max_bwd_masked": [_P, _P, _P, _P, _c.c_int64, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_gemm_f16x2_supported": [_I, _I],
    "hitadv_bmm_f32_supported": [_I, _I, _I],
    "hitadv_bmm_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hitadv_offset_attention_supported": [_I],
    "hitadv_offset_attention_fwd": [_P, _I, _I, _P, _P, _P],
    "hitadv_offset_attention_bwd": [_P, _P, _P, _I, _I, _P, _P, _P],
    "hitadv_group_linear_max_g16_supported": [_I, _I, _I],
    "hitadv_group_linear_max_g16_fwd": [_P, _P, _P, _c.c_int64, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_group_linear_max_g16_bwd": [_P, _P, _P, _c.c_int64, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_split_rows_f16x2": [_P, _I, _I, _P, _P, _P],
    "hitadv_gemm_f16x2": [_P, _P, _P, _P, _c.c_int64, _I, _I, _I, _P, _P, _P],
    "hitadv_linear_lrelu_pool_scratch": [_I, _I, _I],
    "hitadv_linear_lrelu_pool_fwd": [_P, _P, _P, _I, _I, _I, _I, _c.c_float, _P, _P, _P, _P, _P, _P, _P, _P],
    "hitadv_linear_lrelu_pool_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _c.c_float, _P, _P, _P],
    "hitadv_iteration_head_reg_stack": [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P,
                                        _P, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _I, _P],
    "hitadv_deform_bwd_partials_reg_stack": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P],
    "hitadv_adam_step_partials_reg_stack": [_I, _P, _P, _P, _I, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F,
                                            _F, _F, _P, _P],
    "hitadv_iteration_head_scratch_floats": [_I],
    "hitadv_regulariser_fwd_fused": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "hitadv_deform_bwd_partials": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "hitadv_deform_bwd_slabs": [_I],
    "hitadv_adam_step_partials": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _F, _P, _P],
    "hitadv_deform_bwd_partials_reg": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P],
    "hitadv_deform_bwd_adam_reg": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P, _I, _I, _I, _F, _F,
                                   _F, _F, _F, _F, _P, _P, _P, _P],
    "hitadv_adam_step_partials_reg": [_P, _P, _P, _I, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F,
                                      _F, _F, _P, _P],
    "hitadv_linear_max_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "hitadv_max_over_points": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P],
    "hitadv_max_over_points_scratch": [_I, _I],
    "hitadv_linear_max_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "hitadv_linear_max_fwd_scratch": [_I, _I, _I],
    "hitadv_split_weights_bf16x3": [_P, _I, _I, _P, _P],
    "hitadv_linear_max_fwd_bf16x3": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "hitadv_linear_max_fwd_bf16x3_scratch": [_I, _I, _I, _I],
    "hitadv_split_weights_f16x2": [_P, _I, _I, _P, _P, _P],
    "hitadv_linear_max_fwd_f16x2_packed": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "hitadv_linear_max_fwd_f16x2": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "hitadv_pointnet_rowmlp_fwd": [_I] + [_P] * 13 + [_I, _I, _I, _P, _P],
    "hitadv_pointnet_rowmlp_fwd_stn": [_P] * 15 + [_I, _I, _I, _P, _P],
    "hitadv_pointnet_rowmlp_fwd_deform": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "hitadv_pointnet_rowmlp_tiles": [_I],
    "hitadv_pointnet_rowmlp_form": [_I],
    "hitadv_pointnet_rowmlp_bwd_tiles": [_I, _I],
    "hitadv_pointnet_rowmlp_bwd_words": [_I, _I, _I],
    "hitadv_pointnet_rowmlp_bwd": [_I, _P, _P, _P, _P, _I] + [_P] * 15 + [_I, _I, _I, _I, _P],
    "hitadv_pointnet_rowmlp_bwd_fix": [_I, _P, _P, _P, _P, _I] + [_P] * 15 + [_I, _I, _I, _I, _P, _P],
    "hitadv_sum_partials": [_P, _P, _I, _I, _I, _P, _P],
    "hitadv_knn_features": [_P, _P, _I, _I, _I, _I, _P, _P],
    "hitadv_row_sqnorm": [_P, _L, _I, _P, _P],
    "hitadv_edge_max_fwd": [_P, _P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "hitadv_edge_max_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P],
    "hitadv_edge_max_bwd_scratch_ints": [_I, _I, _I],
    "hitadv_group_add_relu_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "hitadv_group_add_relu_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "hitadv_group_add_relu_bwd_scratch_ints": [_I, _I, _I, _I],
    "hitadv_lrelu_pool_fwd": [_P, _I, _I, _I, _F, _P, _P, _P],
    "hitadv_lrelu_pool_bwd": [_P, _P, _P, _I, _I, _I, _F, _P, _P],
    "hitadv_fc_layer": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "hitadv_fc_layer_scratch_floats": [_I, _I, _I],
    "hitadv_fc_layer_pre": [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
}
_RESTYPE = {"hitadv_version": _c.c_char_p, "hitadv_deform_bwd_scratch_floats": _c.c_int64,
            "hitadv_max_over_points_scratch": _c.c_int64, "hitadv_linear_max_fwd_scratch": _c.c_int64, "hitadv_linear_max_fwd_bf16x3_scratch": _c.c_int64, "hitadv_pointnet_rowmlp_tiles": _c.c_int64, "hitadv_pointnet_rowmlp_bwd_tiles": _c.c_int64, "hitadv_fc_layer_scratch_floats": _c.c_int64, "hitadv_regulariser_scratch_floats": _c.c_int64,
            "hitadv_edge_max_bwd_scratch_ints": _c.c_int64, "hitadv_iteration_head_scratch_floats": _c.c_int64, "hitadv_deform_bwd_slabs": _c.c_int64, "hitadv_group_add_relu_bwd_scratch_ints": _c.c_int64,
            "hitadv_linear_lrelu_pool_scratch": _c.c_int64}

_lib = None


class HitAdvLibraryError(RuntimeError):
    pass


def load():
    """Load the library once.  torch must already be imported so that the HIP runtime the
    library binds to (SONAME libamdhip64.so.7) is the one torch itself uses."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (load order matters, see docstring)
    if not os.path.exists(LIB_PATH):
        raise HitAdvLibraryError(
            "libhitadv_hip.so is not built (%s). Build it with `make -C hit_adv_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, _c.c_int)
    _lib = lib
    form = os.environ.get("HITADV_ROWMLP_FORM")  # A/B knob (include/hitadv.h: hitadv_pointnet_rowmlp_form)
    if form is not None:
        lib.hitadv_pointnet_rowmlp_form(int(form))
    return lib


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        what = "invalid argument" if rc == -1 else "hipError %d" % rc
        raise HitAdvLibraryError("%s failed: %s" % (name, what))
