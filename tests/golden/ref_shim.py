"""Import shim for the reference (cvlab-stonybrook/s-volsdf) on a CPU-only box.

Used ONLY by tests/golden/make_fixtures.py, in the build container, to emit golden
arrays.  Nothing from /root/reference is copied: this file registers empty stub
modules for third-party packages the image lacks and makes `.cuda()` an identity.
It is never imported by the product, the tests or the bench (they take the model
configuration from volsdf.utils.conf; the copies below feed the REFERENCE's classes,
whose `volsdf` package shadows this repository's while the fixtures are generated).
"""
import sys
import types

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
