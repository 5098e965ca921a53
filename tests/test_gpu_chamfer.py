"""GPU parity of the Chamfer-evaluator kernels (csrc/svs_cloud.hip) through the C-ABI:
against the reference script's outputs (fixture chamfer_ref.npz) and sklearn (the oracle) on other shapes.
Bars: keep masks and neighbour indices exact; distances bit-equal to the kd-tree's (same float64 arithmetic) for
neighbours closer than the search radius; means to 1e-12 relative (summation order)."""
import os

import numpy as np
import pytest
import torch

import chamfer_oracle as corc
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_evaluate_scan_matches_reference_script(dev, golden_dir):
    from evals import eval_dtu
    g = dict(np.load(os.path.join(golden_dir, "chamfer_ref.npz")))
    sc = synth.make_dtu_scan(int(g["seed"]))
    (acc, comp, overall), d = eval_dtu.evaluate_scan(sc["data_pcd"], sc["stl"], sc["ObsMask"], sc["BB"], sc["Res"], sc["P"],
                                                     shuffle_rng=np.random.default_rng(int(g["shuffle_seed"])), details=True)
    np.testing.assert_array_equal(d["data_pcd"][:64], g["data_pcd_shuffled_head"])
    assert np.array_equal(d["keep"].cpu().numpy(), np.unpackbits(g["keep"])[:len(sc["data_pcd"])].astype(bool))
    assert (len(d["data_down"]), len(d["data_in"]), len(d["data_in_obs"]), len(d["stl_above"])) == \
        (int(g["n_down"]), int(g["n_in"]), int(g["n_in_obs"]), int(g["n_stl_above"]))
    for k in ("dist_d2s", "dist_s2d"):
        got, want = d[k].cpu().numpy(), g[k]
        near = want < 20
        np.testing.assert_array_equal(got[near], want[near])
        assert (got[~near] >= 20).all()
    np.testing.assert_allclose([acc, comp, overall], [float(g["mean_d2s"]), float(g["mean_s2d"]), float(g["over_all"])], rtol=1e-12)


def test_mesh_mode_matches_reference_script(dev, golden_dir, tmp_path):
    """--mode mesh end to end: the sampled cloud is the script's point for point (fixture chamfer_mesh_ref.npz), and so are
    the protocol's results on it; once on arrays, once through the file layout and the command line."""
    from evals import eval_dtu
    from scipy.io import savemat
    g = dict(np.load(os.path.join(golden_dir, "chamfer_mesh_ref.npz")))
    sc = synth.make_dtu_scan(int(g["scan_seed"]))
    vertices, triangles = synth.make_dtu_mesh(int(g["mesh_seed"]))
    cloud = eval_dtu.sample_mesh(vertices, triangles, 0.2).cpu().numpy()
    want, per_tri = corc.sample_mesh(vertices, triangles, 0.2)
    np.testing.assert_array_equal(cloud, want)                                   # HIP == oracle, bit for bit
    new_pts = cloud[len(vertices):]
    assert len(new_pts) == int(g["n_new_pts"]) == int(per_tri.sum())
    np.testing.assert_array_equal(new_pts[::61], g["new_pts_every_61"])          # == the reference script's points
    np.testing.assert_array_equal(new_pts.sum(0), g["new_pts_sum"])
    (acc, comp, overall), d = eval_dtu.evaluate_scan(cloud, sc["stl"], sc["ObsMask"], sc["BB"], sc["Res"], sc["P"],
                                                     shuffle_rng=np.random.default_rng(int(g["shuffle_seed"])), details=True)
    np.testing.assert_array_equal(d["data_pcd"][:64], g["data_pcd_shuffled_head"])
    assert np.array_equal(d["keep"].cpu().numpy(), np.unpackbits(g["keep"])[:len(cloud)].astype(bool))
    assert (len(d["data_down"]), len(d["data_in"]), len(d["data_in_obs"]), len(d["stl_above"])) == \
        (int(g["n_down"]), int(g["n_in"]), int(g["n_in_obs"]), int(g["n_stl_above"]))
    for k in ("dist_d2s", "dist_s2d"):
        got, ref = d[k].cpu().numpy(), g[k]
        near = ref < 20
        np.testing.assert_array_equal(got[near], ref[near])
        assert (got[~near] >= 20).all()
    np.testing.assert_allclose([acc, comp, overall], [float(g["mean_d2s"]), float(g["mean_s2d"]), float(g["over_all"])], rtol=1e-12)

    # the file layout of the script (mesh PLY with float64 vertices, so that nothing is lost on the way)
    scan = 24
    ds = tmp_path / "root" / "DTU" / "DTU_MVS_Data"
    (ds / "ObsMask").mkdir(parents=True); (ds / "Points" / "stl").mkdir(parents=True); (tmp_path / "pred").mkdir()
    savemat(str(ds / "ObsMask" / f"ObsMask{scan}_10.mat"), dict(ObsMask=sc["ObsMask"], BB=sc["BB"], Res=sc["Res"]))
    savemat(str(ds / "ObsMask" / f"Plane{scan}.mat"), dict(P=sc["P"]))
    with open(ds / "Points" / "stl" / f"stl{scan:03}_total.ply", "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                 "property double z\nend_header\n" % len(sc["stl"])).encode())
        np.ascontiguousarray(sc["stl"], "<f8").tofile(f)
    with open(tmp_path / "pred" / f"mvsnet{scan:03}_l3.ply", "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(vertices), len(triangles))).encode())
        np.ascontiguousarray(vertices, "<f8").tofile(f)
        for t in triangles:
            f.write(b"\x03" + np.asarray(t, "<i4").tobytes())
    res = eval_dtu.main(["--data_dir_root", str(tmp_path / "root"), "--datadir", str(tmp_path / "pred"), "--scan", str(scan), "--mode", "mesh"])
    # the command line shuffles with an unseeded generator like the script: the means move in the last digits only
    np.testing.assert_allclose(res, [float(g["mean_d2s"]), float(g["mean_s2d"]), float(g["over_all"])], rtol=0.02)


@pytest.mark.parametrize("seed,n_tri,thresh", [(0, 300, 0.2), (1, 5000, 0.05), (2, 40, 1.0)])
def test_mesh_sampler_vs_oracle(dev, seed, n_tri, thresh):
    """Random triangle soups (slivers, tiny and large triangles, some with no sample at all) against the oracle's
    restatement of sample_single_tri: same points, same order, bit for bit."""
    from evals import eval_dtu
    rng = np.random.default_rng(seed)
    vertices = rng.normal(0, 1, (n_tri + 2, 3)) * rng.choice([0.05, 0.5, 3.0], (n_tri + 2, 1))
    triangles = np.stack([rng.integers(0, len(vertices), n_tri) for _ in range(3)], 1)
    triangles[::17, 1] = triangles[::17, 0]                                     # degenerate ones in between
    want, per_tri = corc.sample_mesh(vertices, triangles, thresh)
    got = eval_dtu.sample_mesh(vertices, triangles, thresh).cpu().numpy()
    assert (per_tri == 0).any() and per_tri.max() > 5
    np.testing.assert_array_equal(got, want)
    # no triangle at all: the vertices alone
    np.testing.assert_array_equal(eval_dtu.sample_mesh(vertices, np.zeros((0, 3), np.int64), thresh).cpu().numpy(), vertices)


@pytest.mark.parametrize("n_ref,n_q,seed", [(5000, 3000, 0), (1, 17, 1), (257, 1, 2), (20000, 20000, 3)])
def test_nearest_neighbor_vs_sklearn(dev, n_ref, n_q, seed):
    from evals import eval_dtu
    rng = np.random.default_rng(seed)
    ref = rng.normal(0, 30, (n_ref, 3))
    q = np.concatenate([ref[rng.integers(0, n_ref, n_q // 2)] + rng.normal(0, 0.5, (n_q // 2, 3)),       # near the cloud
                        rng.normal(0, 60, (n_q - n_q // 2, 3))], 0)                                       # anywhere
    if n_ref > 100:
        ref[7] = ref[3]                       # exact duplicates: the lower index wins
        q[0] = ref[3]
    want_d, want_i = corc.nn_distance(ref, q, n_jobs=2)
    for cell in (None, 3.0, 11.0):
        got_d, got_i = eval_dtu.nearest_neighbor(ref, q, max_radius=25.0, cell=cell, return_index=True)
        got_d, got_i = got_d.cpu().numpy(), got_i.cpu().numpy()
        near = want_d < 25.0
        np.testing.assert_array_equal(got_d[near], want_d[near])
        assert (got_d[~near] >= 25.0).all()
        ties = ref[got_i[near]] == ref[want_i[near]]            # same point up to duplicates
        assert ties.all()
    if n_ref > 100:
        assert got_d[0] == 0.0 and got_i[0] == 3
    assert eval_dtu.nearest_neighbor(ref, np.zeros((0, 3)), 5.0).shape == (0,)
    with pytest.raises(ValueError):
        eval_dtu.nearest_neighbor(np.zeros((0, 3)), q, 5.0)


@pytest.mark.parametrize("n,radius,seed", [(4000, 0.2, 0), (9000, 1.5, 1), (1, 0.2, 2), (300, 50.0, 3)])
def test_radius_downsample_vs_sklearn_greedy(dev, n, radius, seed):
    from evals import eval_dtu
    rng = np.random.default_rng(seed)
    base = rng.normal(0, 8, (max(n // 3, 1), 3))
    pts = (base[rng.integers(0, len(base), n)] + rng.normal(0, radius * 0.7, (n, 3)))
    if n > 10:
        pts[5] = pts[2]                       # duplicates: the later one is dropped
    want = corc.radius_downsample(pts, radius, n_jobs=2)
    got = eval_dtu.radius_downsample(pts, radius).cpu().numpy()
    assert np.array_equal(got, want)
    if n > 1000:
        assert 0.05 < want.mean() < 0.95
    assert eval_dtu.radius_downsample(np.zeros((0, 3)), radius).shape == (0,)


def test_obs_filter_plane_mean_and_compact(dev):
    """The elementwise stages against the oracle, with points sitting exactly on the box faces and grid-cell centres."""
    import ctypes
    from evals import eval_dtu
    from svs_hip import lib
    from svs_hip.ops import _ptr, _stream
    L = lib.load()
    sc = synth.make_dtu_scan(4, n_pred=6000, n_stl=2000)
    pts = sc["data_pcd"].copy()
    BB = sc["BB"]
    pts[0] = BB[0].astype(np.float64) - 60.0          # on the lower padded face (float32-evaluated threshold)
    pts[1] = BB[1].astype(np.float64) + 120.0         # on the upper padded face: excluded (<)
    pts[2] = BB[0].astype(np.float64) + 2.0 * np.array([3.5, 4.5, 7.5])    # half-way between grid nodes: round half to even
    want_in, want_obs = corc.obs_filter(pts, sc["ObsMask"], BB, sc["Res"], 60.0)
    d = torch.from_numpy(pts).to(dev)
    inb = torch.empty(len(pts), dtype=torch.uint8, device=dev); obs = torch.empty_like(inb)
    om = torch.from_numpy(np.ascontiguousarray(sc["ObsMask"])).to(dev)
    bb = (ctypes.c_float * 6)(*[float(v) for v in BB.reshape(-1)])
    lib.check(L.svs_cloud_obs_filter(_ptr(d), len(pts), bb, 2.0, 60.0, _ptr(om), *om.shape, _ptr(inb), _ptr(obs), _stream()), "obs")
    assert np.array_equal(inb.cpu().numpy().astype(bool), want_in) and np.array_equal(obs.cpu().numpy().astype(bool), want_obs)
    assert 0 < want_obs.sum() < want_in.sum() < len(pts)
    np.testing.assert_array_equal(eval_dtu.compact(d, obs).cpu().numpy(), pts[want_obs])
    assert eval_dtu.compact(d, torch.zeros_like(obs)).shape == (0, 3)

    stl = sc["stl"]
    P = sc["P"]
    above = torch.empty(len(stl), dtype=torch.uint8, device=dev)
    plane = (ctypes.c_double * 4)(*[float(v) for v in P.reshape(-1)])
    sd = torch.from_numpy(stl).to(dev)
    lib.check(L.svs_cloud_plane_side(_ptr(sd), len(stl), plane, _ptr(above), _stream()), "plane")
    want = (P.reshape((1, 4)) * np.concatenate([stl, np.ones_like(stl[:, :1])], -1)).sum(-1) > 0
    assert np.array_equal(above.cpu().numpy().astype(bool), want)

    dist = torch.from_numpy(np.abs(np.random.default_rng(0).normal(0, 15, 100001))).to(dev)
    dn = dist.cpu().numpy()
    np.testing.assert_allclose(eval_dtu.mean_below(dist, 20.0), dn[dn < 20.0].mean(), rtol=1e-13)
    assert np.isnan(eval_dtu.mean_below(dist, -1.0))            # empty selection: numpy's mean gives nan


def test_large_cloud_properties(dev):
    """Size-independent properties at a DTU-size cloud (4 M points): every point is its own nearest neighbour at
    distance 0; against a rigidly shifted copy no distance exceeds the shift; down-sampling leaves no two kept points
    within the radius (down-sampling the result again keeps everything) and every dropped point within the radius of a
    kept one."""
    from evals import eval_dtu
    n = 4_000_000
    g = torch.Generator(device="cpu").manual_seed(0)
    d = torch.randn(n, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=1, keepdim=True) * 150.0                  # sphere of radius 150 mm, spacing ~ 0.27 mm
    pts = d.to(dev)
    dist, idx = eval_dtu.nearest_neighbor(pts, pts, max_radius=20.0, return_index=True)
    assert float(dist.max()) == 0.0 and bool((idx.long() == torch.arange(n, device=dev)).all())
    shift = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64, device=dev)
    dist = eval_dtu.nearest_neighbor(pts, pts + shift, max_radius=20.0)
    assert float(dist.max()) <= float(shift.norm()) * (1 + 1e-12) and float(dist.min()) >= 0.0
    keep = eval_dtu.radius_downsample(pts, 0.2)
    kept = eval_dtu.compact(pts, keep)
    assert 0.3 * n < kept.shape[0] < n
    assert bool(eval_dtu.radius_downsample(kept, 0.2).all())      # no two kept points within the radius: a fixed point
    dropped = eval_dtu.compact(pts, ~keep)
    dk = eval_dtu.nearest_neighbor(kept, dropped, 1.0)
    assert float(dk.max()) <= 0.2


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4095, 4096, 4097, 100001, 5_000_011])
def test_compact_keeps_order_at_any_size(dev, n):
    """The ordered compaction (block-wise mask scan, csrc/svs_scan.h) at sizes around its 16-byte / 4096-byte units, with an
    aligned and an unaligned mask: rows with a non-zero mask byte, in order."""
    from evals import eval_dtu
    g = torch.Generator(device="cpu").manual_seed(n)
    pts = torch.randn(n, 3, generator=g, dtype=torch.float64).to(dev)
    raw = (torch.rand(n + 1, generator=g) < 0.37).to(torch.uint8).to(dev)
    raw[raw != 0] = 7                                           # any non-zero byte counts
    for mask in (raw[:n], raw[1:]):                             # raw[1:] starts one byte off a 16-byte boundary
        got = eval_dtu.compact(pts, mask)
        want = pts[mask != 0]
        assert got.shape == want.shape and torch.equal(got, want)
    assert eval_dtu.compact(pts, torch.zeros(n, dtype=torch.uint8, device=dev)).shape == (0, 3)
    assert torch.equal(eval_dtu.compact(pts, torch.ones(n, dtype=torch.uint8, device=dev)), pts)


@pytest.mark.parametrize("n", [1, 63, 1000, 3_000_001])
def test_cloud_bounds(dev, n):
    from svs_hip import lib
    from svs_hip.ops import _ptr, _stream
    L = lib.load()
    pts = (torch.randn(n, 3, dtype=torch.float64) * torch.tensor([1.0, 50.0, 0.01], dtype=torch.float64)).to(dev)
    ws = torch.empty(L.svs_cloud_bounds_workspace_bytes() // 8, dtype=torch.float64, device=dev)
    box = torch.empty(6, dtype=torch.float64, device=dev)
    lib.check(L.svs_cloud_bounds(_ptr(pts), n, _ptr(ws), _ptr(box), _stream()), "svs_cloud_bounds")
    want = torch.cat([pts.min(0).values, pts.max(0).values])
    assert torch.equal(box, want)
