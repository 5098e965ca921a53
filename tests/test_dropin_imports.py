"""The drop-in boundary: with the path ordered as INTEGRATION.md §2 / svs_hip.launch arrange it, the
reference's own import block (runner.py:18-40) and its dataset module import cleanly, the hot-path names resolve to
this repository and everything else to the reference checkout.

Build-container only: the reference never travels to the GPU box, so the test skips without /root/reference.  It runs
in a child interpreter (the package fall-through is decided when the mirror packages are first imported) with empty
stub modules for the third-party packages this image lacks (tests/golden/ref_shim.py).
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "s-volsdf_amd")
REFERENCE = os.environ.get("SVOLSDF_REFERENCE_ROOT", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "runner.py")),
                                reason="needs a checkout of the reference (build container only)")

CHILD = textwrap.dedent("""
    import os, sys, types
    sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
    import ref_shim
    ref_shim.REFERENCE_ROOT = {ref!r}
    ref_shim.install()
    sys.path.remove({ref!r})                       # the launcher below decides the order, not the shim
    ref_shim._stub("plyfile", PlyData=object, PlyElement=object)
    sk = sys.modules["skimage"]
    sk.morphology = ref_shim._stub("skimage.morphology", binary_dilation=None, disk=None)
    sys.modules["omegaconf"].OmegaConf = type("OmegaConf", (), dict(set_struct=staticmethod(lambda *a: None)))
    sys.modules["torchvision"].utils = ref_shim._stub("torchvision.utils", make_grid=None)

    sys.path.insert(0, {pkg!r})
    from svs_hip import launch, refpath
    here, ref = launch.arrange_path(os.path.join({ref!r}, "runner.py"))
    assert sys.path[:2] == [here, ref], sys.path[:3]
    launch.check_resolution(here)

    # runner.py:18-40, the reference's own text read at run time (never stored in this repository)
    lines = open(os.path.join(ref, "runner.py")).read().splitlines()[17:40]
    block = "\\n".join(lines)
    assert "from volsdf.vsdf import VolOpt" in block and "from helpers.utils import *" in block, block
    ns = dict(__name__="runner_imports")
    exec(compile(block, "runner.py[18:40]", "exec"), ns)
    exec("from volsdf.datasets.scene_dataset import SceneDataset\\n"
         "import volsdf.utils.plots as plots\\nimport volsdf.utils.general as general\\n"
         "from volsdf.utils import rend_util\\nfrom helpers.help import run_help\\n"
         "from volsdf.model.network import VolSDFNetwork\\nfrom volsdf.model.network_bg import VolSDFNetworkBG\\n"
         "from volsdf.model.loss import VolSDFLoss\\nimport evals.eval_dtu as eval_dtu", ns)

    def where(obj):
        mod = sys.modules[obj.__module__] if hasattr(obj, "__module__") and not isinstance(obj, types.ModuleType) else obj
        return os.path.realpath(mod.__file__)

    ours = ["VolOpt", "CascadeMVSNet", "check_geometric_consistency", "read_pfm", "save_pfm", "VolSDFNetwork",
            "VolSDFNetworkBG", "VolSDFLoss", "read_camera_parameters", "eval_dtu"]
    theirs = ["TransMVSNet", "UCSNet", "tocuda", "tensor2numpy", "MVSDataset", "get_trains_ids", "get_eval_ids",
              "SceneDataset", "plots", "run_help", "load_K_Rt_from_P", "glob_imgs"]
    for n in ours:
        assert where(ns[n]).startswith(os.path.realpath(here) + os.sep), (n, where(ns[n]))
        assert refpath.in_this_tree(ns[n]), n
    for n in theirs:
        assert where(ns[n]).startswith(os.path.realpath(ref) + os.sep), (n, where(ns[n]))
    # shadowed modules: the accelerated names are ours, the rest of the reference module is still there
    ru, ge = ns["rend_util"], ns["general"]
    assert where(ru).startswith(os.path.realpath(here)) and where(ge).startswith(os.path.realpath(here))
    for n in ("load_K_Rt_from_P", "load_rgb", "get_uv", "quat_to_rot", "lift"):
        assert where(getattr(ru, n)).startswith(os.path.realpath(ref)), n
    for n in ("get_camera_params", "get_sphere_intersections"):
        assert where(getattr(ru, n)).startswith(os.path.realpath(here)), n
    for n in ("split_input", "merge_output", "glob_imgs"):
        assert where(getattr(ge, n)).startswith(os.path.realpath(ref)), n
    assert where(ge.get_class).startswith(os.path.realpath(here))
    # the config plug points resolve through the (shadowed) class lookup to this tree
    for dotted in ("volsdf.model.network.VolSDFNetwork", "volsdf.model.network_bg.VolSDFNetworkBG",
                   "volsdf.model.loss.VolSDFLoss"):
        assert refpath.in_this_tree(ge.get_class(dotted)), dotted
    assert not refpath.in_this_tree(ge.get_class("volsdf.datasets.scene_dataset.SceneDataset"))
    print("DROPIN-OK")
""")


def _run(code, env_extra=None):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    env.pop("SVOLSDF_REFERENCE_ROOT", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_runner_import_block_resolves():
    r = _run(CHILD.format(root=ROOT, ref=REFERENCE, pkg=PKG))
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_packages_stand_alone_without_checkout():
    """Without a checkout the mirror packages hold exactly this repository's modules (the GPU box's situation)."""
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {PKG!r})
        from svs_hip import refpath
        assert refpath.reference_root() is None
        import models, helpers.utils, datasets.data_io, volsdf.utils.general as g, volsdf.utils.rend_util as ru
        assert len(models.__path__) == 1 and len(helpers.__path__) == 1
        assert g.get_class("volsdf.model.loss.VolSDFLoss").__name__ == "VolSDFLoss"
        assert not hasattr(ru, "load_K_Rt_from_P") and hasattr(ru, "get_camera_params")
        try:
            import models.TransMVSNet
        except ModuleNotFoundError:
            print("ALONE-OK")
    """)
    r = _run(code)
    assert r.returncode == 0 and "ALONE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_env_root_must_be_a_checkout(tmp_path):
    code = f"import sys; sys.path.insert(0, {PKG!r}); import models"
    r = _run(code, {"SVOLSDF_REFERENCE_ROOT": str(tmp_path)})
    assert r.returncode != 0 and "not a checkout of the reference" in r.stderr


def test_launcher_runs_a_script_with_this_tree_first(tmp_path):
    """svs_hip/launch.py: the script's directory must not shadow the drop-in modules, argv is the script's own."""
    probe = tmp_path / "probe.py"
    probe.write_text("import sys, models, volsdf.vsdf\n"
                     "print('ARGV', sys.argv[1:])\nprint('VSDF', volsdf.vsdf.__file__)\nprint('PATH', models.__path__)\n"
                     "assert __name__ == '__main__'\n")
    env = dict(os.environ, SVOLSDF_REFERENCE_ROOT=REFERENCE)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(PKG, "svs_hip", "launch.py"), str(probe), "testlist=scan106"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ARGV ['testlist=scan106']" in r.stdout
    assert f"VSDF {os.path.join(PKG, 'volsdf', 'vsdf.py')}" in r.stdout
    assert os.path.join(REFERENCE, "models") in r.stdout
