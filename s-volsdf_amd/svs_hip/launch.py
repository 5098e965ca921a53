"""Run an UNMODIFIED script of the reference (runner.py, eval_vsdf.py) on the MI355X path:

    python -m svs_hip.launch /path/to/s-volsdf/runner.py testlist=scan106 ...        (with s-volsdf_amd on PYTHONPATH)
    python /path/to/s-volsdf_amd/svs_hip/launch.py /path/to/s-volsdf/runner.py ...   (no PYTHONPATH needed)

`python runner.py` itself cannot pick the drop-in modules up from PYTHONPATH: the interpreter puts the script's own
directory in front of every PYTHONPATH entry, so the reference's `volsdf/` would win.  This launcher orders `sys.path`
as [s-volsdf_amd, <reference checkout>, ...], checks that the hot-path modules resolve to this tree, and executes the
script as `__main__` with its own `sys.argv` (hydra finds `config/` next to the script as usual).
"""
import importlib.util
import os
import runpy
import sys

HOT_PATH_MODULES = ("volsdf.vsdf", "volsdf.model.network", "volsdf.model.network_bg", "volsdf.model.ray_sampler",
                    "volsdf.model.loss", "models.CasMVSNet", "helpers.utils", "datasets.data_io")


def arrange_path(script):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script_dir = os.path.dirname(os.path.abspath(script))
    os.environ.setdefault("SVOLSDF_REFERENCE_ROOT", script_dir)
    for d in (script_dir, here):
        while d in sys.path:
            sys.path.remove(d)
    sys.path[:0] = [here, script_dir]
    return here, script_dir


def check_resolution(here):
    wrong = []
    for name in HOT_PATH_MODULES:
        spec = importlib.util.find_spec(name)
        origin = os.path.realpath(spec.origin) if spec and spec.origin else ""
        if not origin.startswith(os.path.realpath(here) + os.sep):
            wrong.append(f"{name} -> {origin or 'not found'}")
    if wrong:
        raise ImportError("hot-path modules do not resolve to the MI355X tree: " + "; ".join(wrong))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    here, _ = arrange_path(script)
    check_resolution(here)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
