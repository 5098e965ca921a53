"""ctypes binding of libsvolsdf_hip.so (the C-ABI declared in include/svolsdf_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVS_LIB_PATH: another build of the same library (A/B timing of kernel variants on one box, tools/ab_variant.sh)
LIB_PATH = os.environ.get("SVS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libsvolsdf_hip.so")

_lib = None


class SvsError(RuntimeError):
    pass


_P = c_void_p   # every device pointer crosses the boundary as a plain address
_PP = POINTER(c_void_p)

# name -> (restype, argtypes); mirrors include/svolsdf_hip.h one to one
class WGradJob(ctypes.Structure):
    """svs_wgrad_job of include/svolsdf_hip.h"""
    _fields_ = [("a0", ctypes.c_void_p), ("b0", ctypes.c_void_p), ("sa0", ctypes.c_longlong), ("sb0", ctypes.c_longlong),
                ("a1", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("sa1", ctypes.c_longlong), ("sb1", ctypes.c_longlong),
                ("b_extra", ctypes.c_void_p), ("s_extra", ctypes.c_longlong),
                ("n_points", ctypes.c_int), ("ldw", ctypes.c_int),
                ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("absmax", ctypes.c_void_p),
                ("rec0", ctypes.c_void_p), ("rec1", ctypes.c_void_p)]


class UnpackJob(ctypes.Structure):
    """svs_unpack_job of include/svolsdf_hip.h"""
    _fields_ = [("dWk", ctypes.c_void_p), ("dbk", ctypes.c_void_p), ("ldw", ctypes.c_int), ("map", ctypes.c_int),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("row_off", ctypes.c_int),
                ("weight_v", ctypes.c_void_p), ("weight_g", ctypes.c_void_p), ("row0", ctypes.c_void_p),
                ("grad_v", ctypes.c_void_p), ("grad_g", ctypes.c_void_p), ("grad_b", ctypes.c_void_p)]


SIGNATURES = {
    "svs_version": (c_int, []),
    "svs_set_deterministic": (c_int, [c_int]),
    "svs_get_deterministic": (c_int, []),
    "svs_last_error_string": (c_char_p, []),
    "svs_rays_from_uv": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P]),
    "svs_stream_bytes": (c_size_t, [c_int]),
    "svs_pack_stream": (c_int, [c_int, c_int, _PP, _PP, _PP, _P, _P, _P]),
    "svs_pack_workspace_bytes": (c_size_t, []),
    "svs_sdf_vals": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, c_int, c_float, c_float, c_int, _P, _P,
                             c_int, c_int, _P]),
    "svs_sdf_hbuf_bytes": (c_size_t, [c_int]),
    "svs_sdf_gbuf_bytes": (c_size_t, [c_int]),
    "svs_feat_tiles_bytes": (c_size_t, [c_int]),
    "svs_sdf_outputs": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, c_int, c_float, c_float, c_int, _P, _P,
                                _P, _P, _P, _P, _P]),
    "svs_tiles_to_rows": (c_int, [_P, c_int, c_int, _P, _P]),
    "svs_rgb_eval": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, _P, c_int, _P, _P, c_int, _P, _P, _P]),
    "svs_rgb_rbuf_bytes": (c_size_t, [c_int]),
    "svs_block_bytes": (c_size_t, [c_int, c_int]),
    "svs_rgb_zbuf_bytes": (c_size_t, [c_int]),
    "svs_sdf_ubuf_bytes": (c_size_t, [c_int]),
    "svs_rgb_bwd": (c_int, [c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "svs_sdf_bwd_a": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P,
                              _P, _P]),
    "svs_sdf_bwd_b": (c_int, [c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "svs_lin8_row0_grad": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "svs_unpack_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "svs_unpack_wgrad_multi": (c_int, [_P, c_int, _P]),
    "svs_sampler_ctl_bytes": (c_size_t, [c_int, c_int]),
    "svs_sampler_ctl_stride": (c_int, []),
    "svs_sampler_cap": (c_int, []),
    "svs_sampler_max_new": (c_int, []),
    "svs_sampler_init": (c_int, [_P, c_int, _P, c_int, c_int, c_float, c_float, c_int, c_float, _P, c_float, c_int,
                                 _P, _P, _P, _P, c_int, _P, _P]),
    "svs_sampler_round": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_float, c_float, c_int, c_float,
                                  c_float, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "svs_composite": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P]),
    "svs_composite_bwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "svs_wgrad": (c_int, [_P, _P, ctypes.c_longlong, ctypes.c_longlong, _P, _P, ctypes.c_longlong, ctypes.c_longlong,
                          _P, ctypes.c_longlong, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "svs_wgrad_multi": (c_int, [_P, c_int, c_int, _P]),
    "svs_bg_points": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_float, _P, _P, _P, _P]),
    "svs_bg_sdf_eval": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "svs_bg_rgb_bwd": (c_int, [c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "svs_bg_sdf_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "svs_bg_rbuf_bytes": (c_size_t, [c_int]),
    "svs_bg_rgb_eval": (c_int, [c_int, _P, c_int, _P, _P, c_int, _P, _P, _P]),
    "svs_composite_bg_bwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, _P, _P, _P, _P, _P]),
    "svs_composite_bg": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P,
                                 _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "svs_adam_workspace_bytes": (c_size_t, []),
    "svs_clip_guard_adam": (c_int, [_P, _P, _P, _P, ctypes.c_longlong, c_int, _P, c_double, c_double, c_double, c_double,
                                    c_double, _P, _P, _P]),
    "svs_cost_lookup": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, POINTER(c_float),
                                _PP, _PP, _PP, POINTER(c_int), _P, _P, _P, _P, _P]),
    "svs_loss": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float,
                         c_float, c_int, c_float, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "svs_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "svs_conv2d": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_featurenet_fpn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "svs_featurenet_fpn": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "svs_featurenet_fpn2": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "svs_conv2d_mfma_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "svs_conv2d_mfma_wfrag_bytes": (c_size_t, [c_int, c_int, c_int]),
    "svs_conv2d_mfma_pack": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "svs_conv2d_mfma": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_conv2d_mfma_lateral": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "svs_chw_to_hwc": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "svs_fuse_mats_per_src": (c_int, []),
    "svs_cloud_grid_bytes": (c_size_t, [c_int]),
    "svs_cloud_nn": (c_int, [_P, c_int, _P, c_int, POINTER(c_double), c_double, c_double, _P, _P, _P, _P]),
    "svs_cloud_downsample_begin": (c_int, [_P, c_int, POINTER(c_double), c_double, _P, _P, _P]),
    "svs_cloud_downsample_round": (c_int, [POINTER(c_double), c_int, c_double, _P, _P, _P, _P]),
    "svs_cloud_obs_filter": (c_int, [_P, c_int, POINTER(c_float), c_double, c_double, _P, c_int, c_int, c_int, _P, _P, _P]),
    "svs_cloud_plane_side": (c_int, [_P, c_int, POINTER(c_double), _P, _P]),
    "svs_cloud_compact": (c_int, [_P, _P, c_int, _P, _P, _P, _P]),
    "svs_cloud_mean_workspace_bytes": (c_size_t, []),
    "svs_cloud_mean_below": (c_int, [_P, c_int, c_double, _P, _P, _P]),
    "svs_cloud_bounds_workspace_bytes": (c_size_t, []),
    "svs_cloud_bounds": (c_int, [_P, c_int, _P, _P, _P]),
    "svs_mesh_sample_count": (c_int, [_P, c_int, _P, _P]),
    "svs_mesh_sample_points": (c_int, [_P, c_int, _P, _P, _P]),
    "svs_fuse_view": (c_int, [_P, _P, _PP, _P, c_int, c_int, c_int, c_float, c_double, c_float, c_int, _P, _P, _P, _P, _P,
                              _P, _P, _P, _P, _P]),
    "svs_fuse_points": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "svs_warp_variance": (c_int, [_P, _PP, POINTER(c_float), c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "svs_conv3d": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_conv3d_mfma_wfrag_bytes": (c_size_t, [c_int]),
    "svs_conv3d_gemm_supported": (c_int, [c_int, c_int]),
    "svs_conv3d_gemm_wfrag_bytes": (c_size_t, [c_int, c_int, c_int]),
    "svs_conv3d_gemm_pack": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "svs_conv3d_gemm": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_conv3d_mfma": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_conv3d_s2c8_wfrag_bytes": (c_size_t, []),
    "svs_conv3d_s2c8": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_conv3d_rows_wfrag_bytes": (c_size_t, [c_int]),
    "svs_conv3d_rows": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_conv3d_c1": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_split_volume_dims": (c_size_t, [c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "svs_split_volume_pack": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "svs_warp_variance_split": (c_int, [_P, _PP, POINTER(c_float), c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "svs_conv3d_pair_wfrag_bytes": (c_size_t, [c_int]),
    "svs_conv3d_pair": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "svs_prob_depth_conf": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "svs_depth_hypotheses": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int,
                                     _P, _P]),
    "svs_selftest_exp": (c_int, [_P, _P, _P, c_int, _P]),
    "svs_selftest_arith": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "svs_selftest_cumsum": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "svs_selftest_rowsum": (c_int, [_P, _P, c_int, c_int, _P]),
}

SIGNATURES.update({
    "svs_randperm_prefix": (c_int, [_P, c_size_t, c_longlong, c_longlong, _P]),
    "svs_eikonal_points": (c_int, [_P, _P, _P, _P, c_int, _P, _P]),
    "svs_split_last": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "svs_stage_in": (c_int, [_P, _P, c_size_t, _P]),
    "svs_plan_build": (c_int, [_P, _PP, c_int, _PP]),
    "svs_plan_run": (c_int, [_P, _P]),
    "svs_plan_info": (c_int, [_P, _P]),
    "svs_plan_describe": (c_int, [_P, c_char_p, c_size_t]),
    "svs_plan_destroy": (c_int, [_P]),
})

# entry points of the experimental kernels: present only in a library built with SVS_BUILD_EXPERIMENTS=1
EXPERIMENTAL_SIGNATURES = {
    "svs_sdf_vals16": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, c_float, c_float, c_int, _P, _P,
                               c_int, c_int, _P]),
    "svs_sdf_vals_pair": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, c_float, c_float, c_int, _P, _P,
                                  c_int, c_int, _P]),
}


ABI_VERSION = 101          # svs_version() of the library this binding was written against (include/svolsdf_hip.h)


def load():
    """Load the shared library once; raises SvsError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- first, so that the HIP runtime torch ships is the one this library binds to
    if not os.path.exists(LIB_PATH):
        raise SvsError(f"{LIB_PATH} not found: run `python s-volsdf_amd/build.py` (there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENTAL_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    have = lib.svs_version()
    if have != ABI_VERSION:
        raise SvsError(f"{LIB_PATH} has ABI version {have}, this binding was written against {ABI_VERSION}: "
                       "rebuild with `python s-volsdf_amd/build.py --force`")
    if os.environ.get("SVS_DETERMINISTIC") == "1":
        lib.svs_set_deterministic(1)
    _lib = lib
    return lib


def deterministic():
    """SVS_DETERMINISTIC=1 (or svs_set_deterministic(1)): weight gradients summed in one fixed order, steps on one stream."""
    return bool(load().svs_get_deterministic())


def check(rc, what=""):
    if rc != 0:
        msg = load().svs_last_error_string()
        raise SvsError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
