import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (GOLDEN, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _gpu_ready():
    """A device and the built library: what every `gpu`-marked test needs."""
    try:
        import torch
        if not torch.cuda.is_available():
            return "no GPU visible (run with -m gpu on the MI355X box)"
    except Exception as e:                                   # pragma: no cover
        return f"torch unavailable: {e}"
    lib = os.path.join(ROOT, "s-volsdf_amd", "lib", "libsvolsdf_hip.so")
    return None if os.path.exists(lib) else f"{lib} is not built (python s-volsdf_amd/build.py)"


def pytest_collection_modifyitems(config, items):
    """On a box without a GPU the `gpu`-marked tests are skipped instead of failing in their fixtures; on the GPU box a
    missing library must FAIL (there is no fallback to hide behind), so only the no-device case skips."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason=_gpu_ready() or "needs MI355X")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
