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
