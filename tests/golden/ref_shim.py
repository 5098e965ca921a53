"""Import shim for the reference (cvlab-stonybrook/s-volsdf) on a CPU-only box.

Used ONLY by tests/golden/make_fixtures.py, in the build container, to emit golden
arrays.  Nothing from /root/reference is copied: this file registers empty stub
modules for third-party packages the image lacks and makes `.cuda()` an identity.
It is never imported by the product, the tests or the bench (they take the model
configuration from volsdf.utils.conf; the copies below feed the REFERENCE's classes,
whose `volsdf` package shadows this repository's while the fixtures are generated).
"""
import ctypes
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


class _NoLog:
    def __getattr__(self, k):
        return lambda *a, **kw: None


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    _stub("loguru", logger=_NoLog())
    for n in ("imageio", "cv2", "GPUtil", "hydra", "trimesh"):
        _stub(n)
    _stub("skimage", measure=_stub("skimage.measure"))
    _stub("omegaconf", OmegaConf=object, DictConfig=object)
    _stub("pyhocon", ConfigFactory=object)
    _stub("torchvision", ops=_stub("torchvision.ops", DeformConv2d=None, deform_conv2d=None))
    _stub("torch.utils.tensorboard", SummaryWriter=object)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


# ------------------------------------------------------------------------------------------
# torch.exp pinned to torch's OPEN implementation
# ------------------------------------------------------------------------------------------
_SLEEF = {}


def torch_vec8(name):
    """A numpy float32 -> float32 function that evaluates the 8-lane routine `name` exported by torch's own
    libtorch_cpu.so (Sleef_expf8_u10, Sleef_expm1f8_u10, ...) through the trampoline sleef_call.c."""
    if "lib" not in _SLEEF:
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(tempfile.mkdtemp(prefix="svs_sleef_"), "sleef_call.so")
        subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-shared", "-fPIC",
                               os.path.join(here, "sleef_call.c"), "-o", so])
        _SLEEF["tramp"] = ctypes.CDLL(so)
        _SLEEF["lib"] = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))
    fn = ctypes.cast(getattr(_SLEEF["lib"], name), ctypes.c_void_p)

    def call(x):
        x = np.ascontiguousarray(x, np.float32)
        flat = np.concatenate([x.ravel(), np.zeros((-x.size) % 8, np.float32)])
        out = np.empty_like(flat)
        _SLEEF["tramp"].svs_apply8(fn, flat.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_long(flat.size))
        return out[:x.size].reshape(x.shape)

    return call


def pin_open_exp():
    """Bind `torch.exp` and `torch.sqrt` of float32 CPU tensors to torch's OPEN implementations: Sleef_expf8_u10 (what
    ATen's Vectorized<float>::exp() calls, and what `torch.exp` IS in a build without MKL) and the IEEE square root
    (Vectorized<float>::sqrt() = vsqrtps).  In this MKL build both dispatch to Intel MKL VML (vmsExp / vmsSqrt), closed
    source, whose AVX-512 / AVX2 / SSE kernels disagree with each other: exp on 1-4 % of all float32 inputs, sqrt on
    0.7 % (the AVX-512 kernel is not correctly rounded; check_primitives.py) -- i.e. the reference's own results depend
    on the host.  The pin makes the fixtures host-independent and restatable; `expm1` (already Sleef) and `sum`
    (ATen) are not touched.  Returns a function that restores the original bindings."""
    sleef_exp = torch_vec8("Sleef_expf8_u10avx2")        # the AVX2 + FMA build of the routine, named explicitly
    saved = (torch.exp, torch.sqrt)
    o_exp, o_sqrt = torch.exp, torch.sqrt

    class _Exp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            y = torch.from_numpy(sleef_exp(x.detach().numpy()).reshape(tuple(x.shape)))
            ctx.save_for_backward(y)
            return y

        @staticmethod
        def backward(ctx, g):
            (y,) = ctx.saved_tensors
            return g * y                       # d exp = exp, the formula of torch's own derivative

    class _Sqrt(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            with np.errstate(invalid="ignore"):
                y = torch.from_numpy(np.asarray(np.sqrt(x.detach().contiguous().numpy())).reshape(tuple(x.shape)))
            ctx.save_for_backward(y)
            return y

        @staticmethod
        def backward(ctx, g):
            (y,) = ctx.saved_tensors
            return g / (2 * y)                 # torch's own formula

    def _plain(x, a, kw):
        return not a and not kw and isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.device.type == "cpu"

    def exp(x, *a, **kw):
        return _Exp.apply(x) if _plain(x, a, kw) else o_exp(x, *a, **kw)

    def sqrt(x, *a, **kw):
        return _Sqrt.apply(x) if _plain(x, a, kw) else o_sqrt(x, *a, **kw)

    # the function forms only: every use in the reference's hot path is `torch.exp(...)` / `torch.sqrt(...)`
    # (ray_sampler.py:78,110,130-146,225-227, network_bg.py:158-177,191, rend_util.py:213); the tensor METHODS stay
    # torch's, so torch.optim.Adam's `.sqrt()` is not touched
    torch.exp, torch.sqrt = exp, sqrt

    def restore():
        torch.exp, torch.sqrt = saved
    return restore


class DictConf(dict):
    """dict-backed stand-in for pyhocon.ConfigTree (get_int/get_float/... with dotted keys)."""

    def _get(self, key, default=None):
        cur = self
        for part in key.split("."):
            if not isinstance(cur, dict) or part not in cur:
                if default is None:
                    raise KeyError(key)
                return default
            cur = cur[part]
        return cur

    def get_int(self, k, default=None):
        return int(self._get(k, default))

    def get_float(self, k, default=None):
        return float(self._get(k, default))

    def get_bool(self, k, default=None):
        v = self._get(k, default)
        return bool(v)

    def get_list(self, k, default=None):
        return list(self._get(k, default))

    def get_string(self, k, default=None):
        return str(self._get(k, default))

    def get_config(self, k, default=None):
        v = self._get(k, default)
        return DictConf(v)


def dtu_model_conf(near=1e-4, beta=0.1):
    """Values of config/vol/dtu.yaml:27-56 overlaid with config/ours.yaml:22-24."""
    return DictConf(
        feature_vector_size=256,
        scene_bounding_sphere=3.0,
        implicit_network=dict(d_in=3, d_out=1, dims=[256] * 8, geometric_init=True, bias=0.6,
                              skip_in=[4], weight_norm=True, multires=6, sphere_scale=20.0),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, dims=[256] * 4, weight_norm=True,
                               multires_view=1),
        density=dict(params_init=dict(beta=beta), beta_min=0.0001),
        ray_sampler=dict(near=near, N_samples=64, N_samples_eval=128, N_samples_extra=32,
                         eps=0.1, beta_iters=10, max_total_iters=5),
    )


def bmvs_model_conf(beta=0.1):
    """Values of config/vol/bmvs.yaml:26-77 (the fg + inverted-sphere bg model, VolSDFNetworkBG)."""
    return DictConf(
        feature_vector_size=256,
        scene_bounding_sphere=3.0,
        implicit_network=dict(d_in=3, d_out=1, dims=[256] * 8, geometric_init=True, bias=0.6,
                              skip_in=[4], weight_norm=True, multires=6),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, dims=[256] * 4, weight_norm=True, multires_view=1),
        density=dict(params_init=dict(beta=beta), beta_min=0.0001),
        ray_sampler=dict(near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                         max_total_iters=5, N_samples_inverse_sphere=32, add_tiny=1.0e-6),
        bg_network=dict(
            feature_vector_size=256,
            implicit_network=dict(d_in=4, d_out=1, dims=[256] * 8, geometric_init=False, bias=0.0, skip_in=[4],
                                  weight_norm=False, multires=10),
            rendering_network=dict(mode="nerf", d_in=3, d_out=3, dims=[128], weight_norm=False, multires_view=4),
        ),
    )
