"""GPU parity of the depth-fusion kernels (csrc/svs_fusion.hip) through the C-ABI, against
the reference-generated fixture and the numpy oracle.  The geometry runs in float64 on both sides: masks must agree
except where a float64 quantity sits within rounding of its threshold (summation order of the 3-term dot products:
numpy's BLAS vs the kernel), depths to 1e-6 relative."""
import os

import numpy as np
import pytest
import torch

import fusion_oracle as forc
import synth

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _masks_agree(a, b, max_frac=2e-3):
    diff = np.asarray(a, bool) != np.asarray(b, bool)
    assert diff.mean() <= max_frac, f"{diff.sum()} of {diff.size} mask pixels differ"
    return ~diff


def test_check_geometric_consistency_golden(dev, golden_dir):
    """The drop-in helpers.utils.check_geometric_consistency (HIP) against the reference's outputs."""
    from helpers.utils import check_geometric_consistency
    g = dict(np.load(os.path.join(golden_dir, "fusion_geo.npz")))
    views = synth.make_fusion_views(int(g["seed"]), hw=tuple(int(v) for v in g["hw"]), n_views=3)
    for key in g:
        if not key.endswith("/mask"):
            continue
        tag = key[:-5]
        ref, src = int(tag[1]), int(tag[3])
        fd, fr = (float(v) for v in tag.split("_")[1:])
        m, d, x, y = check_geometric_consistency(views[ref]["depth"], views[ref]["K"], views[ref]["E"], views[src]["depth"],
                                                 views[src]["K"], views[src]["E"], fd, fr)
        assert m.dtype == bool and d.dtype == F32 and x.dtype == F32
        same = _masks_agree(m, g[tag + "/mask"])
        np.testing.assert_allclose(d[same], g[tag + "/depth"][same], rtol=1e-6, atol=1e-7)
        fin = np.isfinite(g[tag + "/x"])
        # float32 roundings of float64 pixel coordinates: at most 1 ulp apart
        np.testing.assert_allclose(x[fin], g[tag + "/x"][fin], rtol=2.5e-7, atol=1e-6)
        np.testing.assert_allclose(y[fin], g[tag + "/y"][fin], rtol=2.5e-7, atol=1e-6)


@pytest.mark.parametrize("hw,n_views,thres,conf", [((48, 64), 3, 1, 0.3), ((37, 53), 4, 2, 0.0), ((25, 31), 2, 1, 0.9)])
def test_fuse_view_vs_oracle(dev, hw, n_views, thres, conf):
    from svs_hip import fusion
    views = synth.make_fusion_views(5 + n_views, hw=hw, n_views=n_views)
    rng = np.random.default_rng(1)
    extra = (rng.uniform(0, 1, hw) > 0.2).astype(F32) if n_views == 4 else None
    for ref in range(n_views):
        srcs = [views[s] for s in range(n_views) if s != ref]
        want = forc.fuse_view(views[ref], srcs, conf=conf, filter_dist=1, filter_diff=0.01, thres_view=thres, extra_mask=extra)
        got = fusion.fuse_view(views[ref], srcs, conf=conf, filter_dist=1, filter_diff=0.01, thres_view=thres, extra_mask=extra)
        assert np.array_equal(got["photo_mask"].cpu().numpy().astype(bool), want["photo_mask"])
        same = _masks_agree(got["geo_mask"].cpu().numpy(), want["geo_mask"])
        _masks_agree(got["final_mask"].cpu().numpy(), want["final_mask"])
        da = got["depth_avg"].cpu().numpy()
        assert da.dtype == np.float64
        # the average changes by a whole term where a per-source mask flips: compare where the geo masks agree and
        # allow the same tiny fraction of outliers
        close = np.isclose(da, want["depth_avg"], rtol=1e-6, atol=1e-9, equal_nan=True)
        assert (~close & same).mean() <= 2e-3
        if np.array_equal(got["final_mask"].cpu().numpy().astype(bool), want["final_mask"]):
            np.testing.assert_allclose(got["xyz"].cpu().numpy(), want["xyz"], rtol=2e-6, atol=2e-6)
            assert np.array_equal(got["rgb"].cpu().numpy(), want["rgb"])
        assert want["final_mask"].sum() > 20


def test_filter_depth_ply_and_edge_cases(dev, tmp_path):
    """filter_depth end to end (all-vs-all pairs, PLY on disk) against the oracle's vertices; no source views; a view
    fused against itself keeps every pixel with positive depth (size-independent property at a DTU-size image)."""
    from svs_hip import fusion
    views = synth.make_fusion_views(9, hw=(30, 44), n_views=3)
    ids = [0, 1, 2]
    pairs = [(v, [s for s in ids if s != v]) for v in ids]
    ply = str(tmp_path / "scan.ply")
    xyz, rgb, stats = fusion.filter_depth(views, pairs, conf=0.2, thres_view=1, plyfilename=ply, mask_dir=str(tmp_path / "mask"))
    want = [forc.fuse_view(views[v], [views[s] for s in src], conf=0.2, thres_view=1) for v, src in pairs]
    n_want = sum(len(w["xyz"]) for w in want)
    assert abs(len(xyz) - n_want) <= 3 and len(stats) == 3 and n_want > 100
    if len(xyz) == n_want:
        np.testing.assert_allclose(xyz, np.concatenate([w["xyz"] for w in want]), rtol=2e-6, atol=2e-6)
        assert np.array_equal(rgb, np.concatenate([w["rgb"] for w in want]))
        assert open(ply, "rb").read() == forc.ply_bytes(xyz, rgb)
    pts, col = fusion.read_ply_points(ply)
    assert pts.shape == (len(xyz), 3) and np.array_equal(col, rgb)
    assert sorted(os.listdir(tmp_path / "mask"))[0] == "00000000_final.png"

    # no source views: geo_mask_sum = 0 everywhere, depth unchanged, nothing survives thres_view = 1
    out = fusion.fuse_view(views[0], [], conf=0.0, thres_view=1)
    assert int(out["geo_mask"].sum()) == 0 and out["xyz"].shape == (0, 3)
    np.testing.assert_array_equal(out["depth_avg"].cpu().numpy(), views[0]["depth"].astype(np.float64))
    out = fusion.fuse_view(views[0], [], conf=0.0, thres_view=0)
    assert int(out["final_mask"].sum()) == 30 * 44

    # self-consistency at full DTU resolution
    H, W = 1200, 1600
    rng = np.random.default_rng(2)
    big = dict(K=np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1]], F32), E=views[1]["E"],
               depth=rng.uniform(400, 900, (H, W)).astype(F32), confidence=np.ones((H, W), F32))
    big["depth"][::7, ::5] = 0.0
    out = fusion.fuse_view(big, [big, big], conf=0.5, thres_view=2, filter_dist=0.25, filter_diff=1e-4)
    fm = out["final_mask"].cpu().numpy().astype(bool)
    # cv2.remap's 1/32-pixel grid returns the pixel itself at integer coordinates: every positive-depth pixel survives
    assert np.array_equal(fm, big["depth"] > 0)
    np.testing.assert_allclose(out["depth_avg"].cpu().numpy()[fm], big["depth"][fm], rtol=1e-6)
    assert out["xyz"].shape[0] == int(fm.sum())


def test_fuse_argument_errors(dev):
    from svs_hip import fusion, lib
    views = synth.make_fusion_views(3, hw=(8, 8), n_views=2)
    with pytest.raises(RuntimeError, match="n_src"):
        fusion.fuse_view(views[0], [views[1]] * 17)
    bad = dict(views[1]); bad["depth"] = np.zeros((8, 9), F32)
    with pytest.raises(AssertionError):
        fusion.fuse_view(views[0], [bad])
    assert lib.load().svs_fuse_mats_per_src() == 68


def test_filter_depth_folder_vs_reference_end_to_end(dev, golden_dir, tmp_path):
    """`svs_hip.fusion.filter_depth_folder` (files in, PLY out) against the reference's own `filter_depth` run on the same
    scan folder (runner.py:301-404 executed by make_fixtures.fx_filter_depth; fixture filter_depth.npz): the three masks per
    view -- on this fixture no pixel differs (float64 geometry on both sides; none of its quantities sits within rounding of a
    threshold) --, every vertex and colour of every pixel both final masks keep (asserted whatever the masks do), and the
    PLY file byte for byte in the layout the reference hands to plyfile."""
    from datasets.data_io import save_pfm
    from PIL import Image
    from svs_hip import fusion
    from test_fusion_cpu import load_filter_depth
    g, ids, views, conf = load_filter_depth(golden_dir)
    scan, out = tmp_path / "scan24", tmp_path / "out" / "scan24"
    for d in (scan / "cams", scan / "images", out / "depth_est", out / "confidence"):
        d.mkdir(parents=True)
    for v in ids:
        (scan / "cams" / "{:0>8}_cam.txt".format(v)).write_bytes(g[f"cam_{v}"].tobytes())
        (scan / "images" / "{:0>8}.jpg".format(v)).write_bytes(g[f"jpg_{v}"].tobytes())
        save_pfm(str(out / "depth_est" / "{:0>8}.pfm".format(v)), views[v]["depth"])
        save_pfm(str(out / "confidence" / "{:0>8}.pfm".format(v)), views[v]["confidence"])
    ply = str(tmp_path / "scan24.ply")
    xyz, rgb, stats = fusion.filter_depth_folder(str(scan), str(out), ply, ids, **conf)
    n_diff = 0
    got_rows, ref_rows = [], []          # vertex rows (ours / the reference's) of the pixels BOTH final masks keep
    got_base = ref_base = 0
    for v in ids:
        for tag in ("photo", "geo", "final"):
            got = np.array(Image.open(str(out / "mask" / "{:0>8}_{}.png".format(v, tag)))) > 0
            ref = np.asarray(g[f"mask_{v}_{tag}"], bool)
            if tag == "photo":
                assert np.array_equal(got, ref)
            else:
                n_diff += int((~_masks_agree(got, ref)).sum())
            if tag == "final":
                # vertices are the kept pixels of a view in row-major order, views in the order of `ids` (runner.py:392-403)
                both = (got & ref).reshape(-1)
                got_rows.append(got_base + (np.cumsum(got.reshape(-1)) - 1)[both])
                ref_rows.append(ref_base + (np.cumsum(ref.reshape(-1)) - 1)[both])
                got_base += int(got.sum()); ref_base += int(ref.sum())
    assert got_base == len(xyz) and ref_base == len(g["vertex_xyz"])
    got_rows, ref_rows = np.concatenate(got_rows), np.concatenate(ref_rows)
    assert len(got_rows) >= len(g["vertex_xyz"]) - 3
    np.testing.assert_allclose(xyz[got_rows], g["vertex_xyz"][ref_rows], rtol=2e-6, atol=2e-6)
    assert np.array_equal(rgb[got_rows], g["vertex_rgb"][ref_rows])
    # the file: the reference's record layout (the oracle's ply_bytes is pinned to it by test_fusion_cpu) of OUR vertices
    assert open(ply, "rb").read() == forc.ply_bytes(xyz, rgb)
    pts, col = fusion.read_ply_points(ply)
    assert pts.shape == (len(xyz), 3) and np.array_equal(col, rgb)
    print(f"filter_depth: {len(xyz)} vertices (reference {len(g['vertex_xyz'])}), {n_diff} mask pixels differ")
    # on this fixture the masks are the reference's pixel for pixel, hence so are the vertex list and the file
    assert n_diff == 0 and len(xyz) == len(g["vertex_xyz"])
