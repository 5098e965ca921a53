"""Configuration plumbing without third-party packages: the pyhocon `ConfigTree` accessors the reference's modules call
(`get_int / get_float / get_string / get_bool / get_list / get_config`, dotted keys, `default=`) over plain nested dicts,
and an attribute / item view for the hydra `args` object (volsdf/vsdf.py:24-27).  omegaconf objects are converted when
omegaconf is installed; plain dicts and namespaces work as they are."""


def to_plain(obj):
    """Nested dict / list copy of a config object (omegaconf DictConfig, dict, namespace, attr_view)."""
    if isinstance(obj, AttrView):
        obj = obj._d
    try:
        from omegaconf import DictConfig, ListConfig, OmegaConf
        if isinstance(obj, (DictConfig, ListConfig)):
            return OmegaConf.to_container(obj, resolve=True, throw_on_missing=True)
    except ImportError:
        pass
    if isinstance(obj, dict):
        return {k: to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_plain(v) for v in obj]
    if hasattr(obj, "__dict__") and not isinstance(obj, type):
        return {k: to_plain(v) for k, v in vars(obj).items()}
    return obj


class AttrView:
    """`args.exps_folder` and `args['vol']` on the same object."""

    def __init__(self, d):
        object.__setattr__(self, "_d", d)

    def __getattr__(self, k):
        try:
            v = self._d[k]
        except (KeyError, TypeError):
            try:
                v = getattr(self._d, k)
            except AttributeError:
                raise AttributeError(k) from None
        return AttrView(v) if isinstance(v, dict) else v

    def __setattr__(self, k, v):
        self._d[k] = v

    __getitem__ = __getattr__

    def __contains__(self, k):
        return k in self._d


def attr_view(obj):
    return obj if not isinstance(obj, dict) else AttrView(obj)


_MISSING = object()


class Conf(dict):
    def _get(self, key, default=_MISSING):
        cur = self
        for part in key.split("."):
            if not isinstance(cur, dict) or part not in cur:
                if default is _MISSING:
                    raise KeyError(key)
                return default
            cur = cur[part]
        return cur

    def get_int(self, key, default=_MISSING):
        return int(self._get(key, default))

    def get_float(self, key, default=_MISSING):
        return float(self._get(key, default))

    def get_string(self, key, default=_MISSING):
        return str(self._get(key, default))

    def get_bool(self, key, default=_MISSING):
        return bool(self._get(key, default))

    def get_list(self, key, default=_MISSING):
        return list(self._get(key, default))

    def get_config(self, key, default=_MISSING):
        v = self._get(key, default)
        return Conf(v) if isinstance(v, dict) else v

    def get(self, key, default=None):
        return self._get(key, default)


def dtu_model_conf(near=1e-4, beta=0.1):
    """The `model` section the reference ships for DTU (config/vol/dtu.yaml:27-56 overlaid with config/ours.yaml:22-24):
    what BASELINE configs[1] runs.  Used by bench.py, smoke() and the tests; a real run takes it from `args.vol.model`."""
    return Conf(
        feature_vector_size=256,
        scene_bounding_sphere=3.0,
        implicit_network=dict(d_in=3, d_out=1, dims=[256] * 8, geometric_init=True, bias=0.6, skip_in=[4], weight_norm=True,
                              multires=6, sphere_scale=20.0),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, dims=[256] * 4, weight_norm=True, multires_view=1),
        density=dict(params_init=dict(beta=beta), beta_min=0.0001),
        ray_sampler=dict(near=near, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                         max_total_iters=5),
    )


def bmvs_model_conf(beta=0.1):
    """The `model` section for BlendedMVS (config/vol/bmvs.yaml:26-77): foreground + inverted-sphere background,
    VolSDFNetworkBG (BASELINE configs[3])."""
    conf = dtu_model_conf(near=0.0, beta=beta)
    del conf["implicit_network"]["sphere_scale"]
    conf["ray_sampler"].update(N_samples_inverse_sphere=32, add_tiny=1.0e-6)
    conf["bg_network"] = dict(
        feature_vector_size=256,
        implicit_network=dict(d_in=4, d_out=1, dims=[256] * 8, geometric_init=False, bias=0.0, skip_in=[4], weight_norm=False,
                              multires=10),
        rendering_network=dict(mode="nerf", d_in=3, d_out=3, dims=[128], weight_norm=False, multires_view=4),
    )
    return conf
