"""The training pixels of a step (volsdf/datasets/scene_dataset.py:275-279, called from volsdf/vsdf.py:234):
`svs_randperm_prefix` -- a host routine of the C-ABI library -- against torch.randperm itself: the same indices AND the same
generator state afterwards, so every later draw of the run (the sampler's jitter, the eikonal points, the next batches) is
unchanged.  No GPU involved."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built_library():
    """the routine is host code of the C-ABI library: build it if this checkout has not yet (hipcc cross-compiles here)"""
    spec = importlib.util.spec_from_file_location("svs_build", os.path.join(ROOT, "s-volsdf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)


def _fast(n, k):
    from svs_hip import lib
    st = torch.get_rng_state()
    out = torch.empty(k, dtype=torch.int64)
    lib.check(lib.load().svs_randperm_prefix(st.data_ptr(), st.numel(), n, k, out.data_ptr()), "svs_randperm_prefix")
    torch.set_rng_state(st)
    return out


def test_randperm_prefix_equals_torch():
    cases = ((576 * 768, 1024), (576 * 768, 256), (768, 768), (5, 3), (1, 1), (1000, 0), (2000, 1999), (24 * 32, 100),
             (1200 * 1600, 2048), (576 * 768, 576 * 768))
    for seed in range(12):
        for n, k in cases:
            runs = []
            for draw in (lambda: torch.randperm(n)[:k], lambda: _fast(n, k)):
                torch.manual_seed(seed)
                torch.rand(seed * 37 % 700)          # an arbitrary position inside the generator's 624-word block
                idx = draw()
                runs.append((idx, torch.rand(7), torch.randperm(13), torch.randn(5), torch.get_rng_state()))
            for a, b in zip(*runs):
                assert torch.equal(a, b), (seed, n, k)


def test_randperm_prefix_rejects_bad_arguments():
    from svs_hip import lib
    L = lib.load()
    st = torch.get_rng_state()
    out = torch.empty(4, dtype=torch.int64)
    assert L.svs_randperm_prefix(st.data_ptr(), 100, 10, 4, out.data_ptr()) != 0            # not a generator state
    assert L.svs_randperm_prefix(st.data_ptr(), st.numel(), 3, 4, out.data_ptr()) != 0       # k > n
    assert L.svs_randperm_prefix(st.data_ptr(), st.numel(), 1 << 31, 4, out.data_ptr()) != 0  # ATen's other algorithm
    assert L.svs_randperm_prefix(None, st.numel(), 10, 4, out.data_ptr()) != 0
    assert torch.equal(st, torch.get_rng_state())


def test_volopt_resample_keeps_the_batches():
    """volsdf.vsdf.change_sampling_idx: the reference's dataset method is recognised and replaced draw for draw; a dataset
    with another method keeps its own."""
    import synthetic_scene
    from volsdf import vsdf
    ds = synthetic_scene.SyntheticSceneDataset(img_res=(48, 64))
    seqs = []
    for use_fast in (False, True):
        torch.manual_seed(5)
        seq = []
        for step in range(6):
            (lambda k: vsdf.change_sampling_idx(ds, k) if use_fast else ds.change_sampling_idx(k))(200 if step != 3 else -1)
            seq.append(None if ds.sampling_idx is None else ds.sampling_idx.clone())
            seq.append(torch.rand(3))
        seqs.append(seq)
    assert vsdf._RESAMPLE_IS_REFERENCE[type(ds)] is True
    for a, b in zip(*seqs):
        assert (a is None and b is None) or torch.equal(a, b)

    class Other:
        total_pixels = 100
        calls = 0

        def change_sampling_idx(self, sampling_size):
            self.calls += 1
            self.sampling_idx = torch.arange(sampling_size)

    o = Other()
    vsdf.change_sampling_idx(o, 7)
    assert o.calls == 1 and torch.equal(o.sampling_idx, torch.arange(7)) and vsdf._RESAMPLE_IS_REFERENCE[Other] is False


def test_randperm_prefix_self_check_and_fallback(monkeypatch):
    """The per-process self-check of the routine against THIS torch's randperm (volsdf.vsdf.randperm_prefix_matches_torch): it
    passes here, leaves the default generator alone, and where it fails -- a torch whose generator layout or shuffle differs,
    simulated by a library call that returns other indices -- the dataset's own method draws the batch."""
    import warnings
    import synthetic_scene
    from svs_hip import lib
    from volsdf import vsdf
    vsdf._PREFIX_CHECK.clear()
    before = torch.get_rng_state()
    assert vsdf.randperm_prefix_matches_torch(576 * 768, 1024) and vsdf.randperm_prefix_matches_torch(48 * 64, 200)
    assert torch.equal(before, torch.get_rng_state())
    ds = synthetic_scene.SyntheticSceneDataset(img_res=(48, 64))
    real = lib.load()

    class Broken:
        def __getattr__(self, name):
            return getattr(real, name)

        def svs_randperm_prefix(self, state, nbytes, n, k, out):
            rc = real.svs_randperm_prefix(state, nbytes, n, k, out)
            torch.frombuffer((ctypes.c_int64 * max(k, 1)).from_address(out), dtype=torch.int64)[:1] += 1
            return rc

    import ctypes
    vsdf._PREFIX_CHECK.clear()
    monkeypatch.setattr(lib, "load", lambda: Broken())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.manual_seed(9)
        vsdf.change_sampling_idx(ds, 200)
        got = ds.sampling_idx.clone()
        after = torch.rand(3)
    assert any("svs_randperm_prefix" in str(x.message) for x in w)
    torch.manual_seed(9)
    ds.change_sampling_idx(200)
    assert torch.equal(got, ds.sampling_idx) and torch.equal(after, torch.rand(3))
    vsdf._PREFIX_CHECK.clear()
