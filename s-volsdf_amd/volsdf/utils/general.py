"""`volsdf.utils.general` next to the hot path.  With a reference checkout importable every helper of its
volsdf/utils/general.py (`glob_imgs`, `split_input`, `merge_output`, ...) is re-exported unchanged (svs_hip/refpath.py);
what this repository's `VolOpt` needs itself -- the dotted-name class lookup of the config plug points
(`train.model_class`, `train.loss_class`, `train.dataset_class`; volsdf/vsdf.py:93,98 of the reference) and the
folder helper -- is defined here so that the path also runs without a checkout.  Whole-image chunking lives in
svs_hip/renderer.py, not here."""
import importlib
import os

from svs_hip.refpath import overlay

overlay(globals(), __name__)


def mkdir_ifnotexists(directory):
    os.makedirs(directory, exist_ok=True)


def get_class(kls):
    """'volsdf.model.network.VolSDFNetwork' -> the class object."""
    module, _, attr = kls.rpartition(".")
    return getattr(importlib.import_module(module), attr)
