"""CPU-side checks of the C-ABI: the library builds for gfx950, loads, and exports exactly what
include/svolsdf_hip.h declares with the argument counts the ctypes table uses (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("svs_build", os.path.join(ROOT, "s-volsdf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)
    from svs_hip import lib as L
    return L


def _header_decls():
    src = open(os.path.join(ROOT, "include", "svolsdf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    # (the experimental kernels' prototypes: only in a library built with SVS_BUILD_EXPERIMENTS=1)
    src = re.sub(r"#ifdef SVS_EXPERIMENTAL_KERNELS.*?#endif", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(svs_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
    return decls


def test_header_matches_library_and_ctypes_table(lib):
    decls = _header_decls()
    L = lib.load()
    assert set(decls) == set(lib.SIGNATURES), set(decls) ^ set(lib.SIGNATURES)
    for name, nargs in decls.items():
        assert hasattr(L, name), f"{name} declared in the header but not exported"
        assert len(lib.SIGNATURES[name][1]) == nargs, name
    assert L.svs_version() == 101


def test_size_queries(lib):
    L = lib.load()
    assert L.svs_stream_bytes(1) > L.svs_stream_bytes(0) > 1 << 20
    assert L.svs_stream_bytes(3) > 1 << 20
    assert L.svs_sampler_cap() >= 640 and L.svs_sampler_max_new() == 128
    assert L.svs_sdf_hbuf_bytes(128) == 4 * 8 * 128 * 64 * 4
    assert L.svs_feat_tiles_bytes(129) == 8 * 128 * 64 * 4


def test_argument_errors_are_reported(lib):
    L = lib.load()
    rc = L.svs_rays_from_uv(None, None, None, 0, None, None, None, None)
    assert rc < 0 and b"svs_rays_from_uv" in L.svs_last_error_string()
    rc = L.svs_composite(4, 1000, None, None, None, None, None, None, 0.0, None, None, None, None, None, None)
    assert rc < 0


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libsvolsdf_hip.so")
    with pytest.raises(lib.SvsError):
        lib.load()


def test_precision_setting_is_validated(monkeypatch):
    """SVS_MLP_PRECISION names one of the three arithmetic settings (f32 covers the background networks too since round 5:
    csrc/svs_bg_f32.hip); anything else is an error before any device work."""
    import pytest
    from svs_hip import ops
    monkeypatch.setenv("SVS_MLP_PRECISION", "f32")
    assert ops.default_precision() == ops.F32
    monkeypatch.setenv("SVS_MLP_PRECISION", "bf16")
    with pytest.raises(ValueError):
        ops.default_precision()
    # the background entry points refuse a precision they do not know (argument check, nothing is launched)
    import ctypes
    from svs_hip import lib
    L = lib.load()
    d = ctypes.c_void_p(64)
    assert L.svs_bg_sdf_eval(d, 32, d, 7, d, d, None, None, None, None) < 0 and b"precision" in L.svs_last_error_string()
    assert L.svs_bg_rgb_eval(32, d, 0, d, d, 7, d, None, None) < 0 and b"precision" in L.svs_last_error_string()


def test_argument_errors_come_back_as_codes():
    """Entry points validate their arguments before touching the device: null pointers and bad shapes return the
    documented negative codes (and set the error string) -- no GPU needed, nothing is launched."""
    import ctypes
    from svs_hip import lib
    L = lib.load()
    dummy = ctypes.c_void_p(64)                      # never dereferenced: the shape checks come first
    rc = L.svs_featurenet_fpn(None, 512, 640, 8, None, None, None, None, None, None, None)
    assert rc < 0 and b"null" in L.svs_last_error_string()
    rc = L.svs_featurenet_fpn(dummy, 510, 640, 8, dummy, dummy, dummy, dummy, dummy, dummy, None)
    assert rc < 0 and b"multiples of 4" in L.svs_last_error_string()
    assert L.svs_featurenet_fpn_workspace_bytes(8, 512, 640) > 0 and L.svs_featurenet_fpn_workspace_bytes(8, 2, 2) == 0
    rc = L.svs_conv2d(dummy, dummy, None, None, 0, dummy, 8, 8, 16, 16, 4, 1, 0, None)        # k = 4
    assert rc < 0 and b"k in {1,3,5}" in L.svs_last_error_string()
    rc = L.svs_wgrad_multi(None, 0, 1, None)
    assert rc < 0
    rc = L.svs_mesh_sample_count(None, 5, None, None)
    assert rc < 0 and b"svs_mesh_sample_count" in L.svs_last_error_string()
    rc = L.svs_mesh_sample_points(dummy, 5, None, dummy, None)
    assert rc < 0 and b"svs_mesh_sample_points" in L.svs_last_error_string()
    assert L.svs_mesh_sample_count(None, 0, None, None) == 0          # an empty mesh is not an error
    # launch plans and the eikonal points: argument checks come before any HIP call
    plan = ctypes.c_void_p()
    assert L.svs_plan_build(None, None, 0, ctypes.byref(plan)) < 0 and not plan.value
    assert L.svs_plan_build(dummy, None, 2, ctypes.byref(plan)) < 0          # two side streams announced, none given
    assert L.svs_plan_run(None, None) < 0 and L.svs_plan_destroy(None) < 0
    counts = (ctypes.c_int * 8)()
    assert L.svs_plan_info(None, counts) < 0 and L.svs_plan_describe(None, None, 0) < 0
    rc = L.svs_eikonal_points(None, dummy, dummy, dummy, 4, dummy, None)
    assert rc < 0 and b"svs_eikonal_points" in L.svs_last_error_string()
    assert L.svs_eikonal_points(dummy, dummy, dummy, dummy, 0, dummy, None) < 0
    assert L.svs_split_last(dummy, 4, 1, dummy, dummy, None) < 0 and L.svs_split_last(None, 4, 3, dummy, dummy, None) < 0
    assert L.svs_stage_in(None, dummy, 64, None) < 0 and L.svs_stage_in(dummy, dummy, 6, None) < 0
    assert L.svs_stage_in(ctypes.c_void_p(68), dummy, 64, None) < 0            # not 16-byte aligned
