"""How the drop-in packages of this repository (`models`, `helpers`, `datasets`, `evals`, `volsdf`, `volsdf.utils`,
`volsdf.model`) coexist with a checkout of the reference on one `sys.path`.

The reference imports its own code by absolute names (`runner.py:32-40`: `models.TransMVSNet`, `helpers.utils`,
`datasets.general_eval`, `volsdf.datasets.scene_dataset`, `volsdf.vsdf` ...).  This repository provides only the
hot-path modules under those names; everything else has to keep coming from the reference.  Two mechanisms:

* **package fall-through** -- every mirror package's `__init__` calls `extend_package_path(__name__, __path__)`, which
  appends the same-named directory of the reference checkout to the package's search path.  A submodule that exists
  here wins, one that does not (`models/TransMVSNet.py`, `volsdf/datasets/`, `volsdf/utils/plots.py`,
  `datasets/general_eval.py`, `helpers/help.py` ...) is found in the reference.
* **module overlay** -- a module that exists on both sides but is only partly on the hot path (`helpers/utils.py`,
  `volsdf/utils/rend_util.py`, `volsdf/utils/general.py`) starts with `overlay(globals(), __name__)`: the reference's
  module of that name is loaded under a private name and its public names are copied in, then the definitions that
  follow in the file replace the accelerated ones.

The checkout is found through `SVOLSDF_REFERENCE_ROOT` or, failing that, the first `sys.path` entry other than this
tree that holds `volsdf/vsdf.py` and `models/CasMVSNet.py`.  Without a checkout both mechanisms are no-ops and the
packages hold just the modules of this repository (the situation on a box that only runs the hot path).

`python -m svs_hip.launch /path/to/s-volsdf/runner.py ...` (svs_hip/launch.py) arranges the path and runs an unmodified
script of the reference.
"""
import importlib.util
import os
import sys

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # .../s-volsdf_amd
_PRIVATE = "_svs_reference."
_MARKERS = (os.path.join("volsdf", "vsdf.py"), os.path.join("models", "CasMVSNet.py"))


def _is_checkout(d):
    return bool(d) and os.path.realpath(d) != os.path.realpath(PKG_ROOT) and all(
        os.path.isfile(os.path.join(d, m)) for m in _MARKERS)


def reference_root():
    """Directory of the reference checkout, or None."""
    env = os.environ.get("SVOLSDF_REFERENCE_ROOT")
    if env:
        if not _is_checkout(env):
            raise ImportError(f"SVOLSDF_REFERENCE_ROOT={env!r} is not a checkout of the reference "
                              f"(expected {_MARKERS[0]} and {_MARKERS[1]} below it)")
        return os.path.abspath(env)
    for entry in sys.path:
        d = os.path.abspath(entry or os.getcwd())
        if _is_checkout(d):
            return d
    return None


def extend_package_path(name, path):
    """`__path__ = extend_package_path(__name__, __path__)` in a mirror package's `__init__`."""
    root = reference_root()
    if root is not None:
        d = os.path.join(root, *name.split("."))
        if os.path.isdir(d) and d not in path:
            path.append(d)
    return path


def reference_module(name):
    """The reference's module `name` (dotted), loaded from its file under a private `sys.modules` key so that it does
    not collide with the module of this repository that carries the public name.  None without a checkout / file."""
    key = _PRIVATE + name
    if key in sys.modules:
        return sys.modules[key]
    root = reference_root()
    if root is None:
        return None
    path = os.path.join(root, *name.split(".")) + ".py"
    if not os.path.isfile(path):
        return None
    spec = importlib.util.spec_from_file_location(key, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[key] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[key]
        raise
    return mod


def overlay(namespace, name):
    """Copy the public names of the reference's module `name` into `namespace` (call FIRST in the shadowing module, so
    that its own definitions override).  Returns the reference module or None."""
    ref = reference_module(name)
    if ref is not None:
        for k, v in vars(ref).items():
            if not k.startswith("__"):
                namespace.setdefault(k, v)
    return ref


def in_this_tree(obj_or_module):
    """True when the object's defining file lives under this repository's package root."""
    mod = sys.modules.get(getattr(obj_or_module, "__module__", None) or "", obj_or_module)
    f = getattr(mod, "__file__", None)
    return bool(f) and os.path.realpath(f).startswith(os.path.realpath(PKG_ROOT) + os.sep)
