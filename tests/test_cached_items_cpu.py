"""svs_hip.batches.CachedItems: the train items of the reference's DataLoader loop (volsdf/datasets/scene_dataset.py:211-273)
assembled from a pixel grid built once -- the SAME items, batches and random-generator use as the dataset's own method,
which builds the first item of every view and is what every assembled item is checked against.  No GPU involved."""
import os
import random
import sys

import numpy as np
import pytest
import torch

import synthetic_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
from svs_hip.batches import CachedItems          # noqa: E402


def _same_batch(a, b):
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        assert list(x) == list(y)
        for k in x:
            assert x[k].shape == y[k].shape and x[k].dtype == y[k].dtype and torch.equal(x[k], y[k]), k


def _loop(ds, items, epochs=4, num_pixels=37, seed=3):
    """the reference's loop (vsdf.py:351-357): per epoch a fresh iterator, per step change_sampling_idx + next()"""
    torch.manual_seed(seed); random.seed(seed)
    loader = torch.utils.data.DataLoader(items, batch_size=1, shuffle=True, collate_fn=items.collate_fn)
    out = []
    for _ in range(epochs):
        ds.change_sampling_idx(num_pixels)
        for batch in loader:
            out.append(batch)
            ds.change_sampling_idx(num_pixels)
    return out, torch.get_rng_state(), random.getstate()


@pytest.mark.parametrize("data_dir", ["DTU", "BlendedMVS"])
def test_cached_items_are_the_datasets_items(data_dir):
    ds = synthetic_scene.SyntheticSceneDataset(img_res=(24, 32), data_dir=data_dir)
    plain, tp, rp = _loop(ds, ds)
    ci = CachedItems(ds)
    fast, tf, rf = _loop(ds, ci)
    assert ci.reason is None and ci.own_items == 3 and ci.fast_items == len(fast) - 3      # one own item per train view
    assert len(plain) == len(fast) == 20
    for a, b in zip(plain, fast):
        _same_batch(a, b)
    assert torch.equal(tp, tf) and rp == rf
    assert ("near_pose" in fast[0][1]) == (data_dir == "BlendedMVS")
    # the whole image (sampling_idx None) and the plot mode go the same way
    ds.change_sampling_idx(-1)
    random.seed(1); a = ds[0]
    random.seed(1); b = ci[0]
    assert a[0] == b[0] and torch.equal(a[1]["uv"], b[1]["uv"]) and torch.equal(a[2]["rgb"], b[2]["rgb"])
    ds.mode = 'plot'
    n = ci.own_items
    assert ci[0][0] in (1, 3) and ci.own_items == n + 1


def test_cached_items_step_aside_for_another_dataset():
    """a dataset whose item is not the reference's -- here: two draws per item, and one whose uv differs -- is served by its
    own method, with the same items and generator states as without the wrapper"""
    class TwoDraws(synthetic_scene.SyntheticSceneDataset):
        def __getitem__(self, idx):
            random.random()
            return super().__getitem__(idx)

    class Centers(synthetic_scene.SyntheticSceneDataset):
        def __getitem__(self, idx):
            i, s, g = super().__getitem__(idx)
            s["uv"] = s["uv"] + 0.25
            return i, s, g

    for cls in (TwoDraws, Centers):
        ds = cls(img_res=(12, 16))
        plain, tp, rp = _loop(ds, ds, epochs=2)
        ci = CachedItems(ds)
        wrapped, tw, rw = _loop(ds, ci, epochs=2)
        assert ci.reason is not None and ci.fast_items == 0
        for a, b in zip(plain, wrapped):
            _same_batch(a, b)
        assert torch.equal(tp, tw) and rp == rw

    class Bare(torch.utils.data.Dataset):
        mode = 'train'

        def __len__(self):
            return 2

        def __getitem__(self, i):
            return i, {"uv": torch.zeros(3, 2)}, {"rgb": torch.ones(3, 3)}

        def collate_fn(self, b):
            return b
    ci = CachedItems(Bare())
    assert "no " in ci.reason and ci[1][0] == 1


@pytest.mark.skipif(not os.path.isdir("/root/reference/volsdf"), reason="needs the reference checkout (build container)")
@pytest.mark.parametrize("data_dir,pixel_centers", [("DTU", False), ("BlendedMVS", True)])
def test_cached_items_against_the_reference_dataset_class(data_dir, pixel_centers):
    """The REFERENCE's own SceneDataset.__getitem__ / collate_fn / change_sampling_idx (imported from the checkout; the object
    is filled in by hand instead of by __init__, which reads a scan folder): every batch of 4 epochs equal, generators equal."""
    import importlib.util
    import types
    # the module is loaded from the checkout by PATH (this process may already hold the mirror package `volsdf`); the three
    # modules it imports and does not use in __getitem__ / collate_fn / change_sampling_idx are stubbed for the load
    saved = {k: sys.modules.get(k) for k in ("cv2", "volsdf.utils.general", "volsdf.utils.rend_util")}
    try:
        for k in saved:
            if saved[k] is None:
                sys.modules[k] = types.ModuleType(k)
        import volsdf.utils
        had = {k: getattr(volsdf.utils, k, None) for k in ("general", "rend_util")}
        for k in had:
            if had[k] is None:
                setattr(volsdf.utils, k, sys.modules["volsdf.utils." + k])
        spec = importlib.util.spec_from_file_location("ref_scene_dataset", "/root/reference/volsdf/datasets/scene_dataset.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
        for k, v in had.items():
            if v is None and hasattr(volsdf.utils, k):
                delattr(volsdf.utils, k)
    RefDataset = mod.SceneDataset
    H, W, n = 18, 26, 64
    rng = np.random.default_rng(5)
    ds = object.__new__(RefDataset)
    ds.data_dir, ds.scan_id, ds.num_views = data_dir, 4, 3
    ds.img_res, ds.total_pixels = [H, W], H * W
    ds.mode, ds.plot_id, ds.sampling_idx, ds.n_images, ds.use_pixel_centers = 'train', 0, None, n, pixel_centers
    ds.rgb_images = [torch.from_numpy(rng.random((H * W, 3)).astype(np.float32)) for _ in range(n)]
    ds.rgb_smooth = [0.5 * x + 0.1 for x in ds.rgb_images]
    ds.masks = [torch.ones(H * W, 3) for _ in range(n)]
    ds.intrinsics_all = [torch.from_numpy(rng.random((4, 4)).astype(np.float32)) for _ in range(n)]
    ds.pose_all = [torch.from_numpy(rng.random((4, 4)).astype(np.float32)) for _ in range(n)]
    plain, tp, rp = _loop(ds, ds)
    ci = CachedItems(ds)
    fast, tf, rf = _loop(ds, ci)
    assert ci.reason is None and ci.own_items == 3 and ci.fast_items == len(fast) - 3
    for a, b in zip(plain, fast):
        _same_batch(a, b)
    assert torch.equal(tp, tf) and rp == rf
    assert ("near_pose" in fast[0][1]) == (data_dir == "BlendedMVS")
