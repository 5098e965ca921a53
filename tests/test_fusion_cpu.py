"""CPU checks of the depth-fusion row: the oracle against the reference-generated fixture, and
the host-side file formats of the product (PFM, camera text, PLY) -- no GPU work here."""
import os

import numpy as np
import pytest

import fusion_oracle as forc
import synth

F32 = np.float32


def test_oracle_geometric_consistency_matches_reference(golden_dir):
    """check_geometric_consistency of the reference (fixture fusion_geo.npz, cv2.remap bound to the oracle's sampler --
    see make_fixtures.fx_fusion) == the oracle's restatement, bit for bit up to BLAS summation order."""
    g = dict(np.load(os.path.join(golden_dir, "fusion_geo.npz")))
    views = synth.make_fusion_views(int(g["seed"]), hw=tuple(int(v) for v in g["hw"]), n_views=3)
    n = 0
    for key in g:
        if not key.endswith("/mask"):
            continue
        tag = key[:-5]
        ref, src = int(tag[1]), int(tag[3])
        fd, fr = (float(v) for v in tag.split("_")[1:])
        m, d, x, y = forc.check_geometric_consistency(views[ref]["depth"], views[ref]["K"], views[ref]["E"], views[src]["depth"],
                                                      views[src]["K"], views[src]["E"], fd, fr)
        assert np.array_equal(m, g[tag + "/mask"])
        np.testing.assert_array_equal(d, g[tag + "/depth"])
        np.testing.assert_array_equal(x, g[tag + "/x"])
        np.testing.assert_array_equal(y, g[tag + "/y"])
        assert 0.2 < m.mean() < 0.9
        n += 1
    assert n == 8


def test_remap_linear_properties():
    """The cv2.remap restatement: integer coordinates return the pixel, 1/32-pixel steps are exact bilinear blends,
    finer offsets snap to the 1/32 grid, outside and NaN coordinates give the border value 0."""
    rng = np.random.default_rng(0)
    img = rng.uniform(1, 2, (6, 9)).astype(F32)
    yy, xx = np.meshgrid(np.arange(6, dtype=F32), np.arange(9, dtype=F32), indexing="ij")
    np.testing.assert_array_equal(forc.remap_linear(img, xx, yy), img)
    got = forc.remap_linear(img, xx[:, :-1] + F32(0.25), yy[:, :-1])
    np.testing.assert_allclose(got, 0.75 * img[:, :-1] + 0.25 * img[:, 1:], rtol=1e-6)
    snap = forc.remap_linear(img, xx[:, :-1] + F32(0.25 + 0.01), yy[:, :-1])
    np.testing.assert_array_equal(snap, got)
    out = forc.remap_linear(img, np.array([[-1.5, 20.0, np.nan, np.inf]], F32), np.array([[1.0, 1.0, 1.0, 1.0]], F32))
    np.testing.assert_array_equal(out, np.zeros((1, 4), F32))
    edge = forc.remap_linear(img, np.array([[-0.5, 8.5]], F32), np.array([[0.0, 0.0]], F32))      # half inside
    np.testing.assert_allclose(edge, [[0.5 * img[0, 0], 0.5 * img[0, 8]]], rtol=1e-6)


def test_pfm_codec_matches_reference_bytes(golden_dir, tmp_path):
    from datasets.data_io import read_pfm, save_pfm
    g = dict(np.load(os.path.join(golden_dir, "pfm_codec.npz")))
    for tag in ("grey", "colour", "hw1"):
        fn = str(tmp_path / (tag + ".pfm"))
        scale = float(g[tag + "/scale"])
        save_pfm(fn, g[tag + "/image"], int(scale) if scale == int(scale) else scale)
        assert open(fn, "rb").read() == g[tag + "/bytes"].tobytes()
        if tag != "hw1":
            back, sc = read_pfm(fn)
            np.testing.assert_array_equal(back, g[tag + "/read"])
            assert sc == float(g[tag + "/read_scale"])
    with pytest.raises(Exception, match="float32"):
        save_pfm(str(tmp_path / "bad.pfm"), np.zeros((2, 2)))
    with pytest.raises(Exception, match="dimensions"):
        save_pfm(str(tmp_path / "bad.pfm"), np.zeros((2, 2, 2), F32))
    (tmp_path / "junk.pfm").write_bytes(b"P6\n1 1\n255\n")
    with pytest.raises(Exception, match="Not a PFM"):
        read_pfm(str(tmp_path / "junk.pfm"))
    # big-endian files (positive scale line) decode too
    be = tmp_path / "be.pfm"
    be.write_bytes(b"Pf\n2 1\n1.0\n" + np.array([1.5, -2.0], ">f4").tobytes())
    np.testing.assert_array_equal(read_pfm(str(be))[0], np.array([[1.5, -2.0]], F32))


def test_camera_text_round_trip(tmp_path):
    from helpers.utils import read_camera_parameters, write_cam
    rng = np.random.default_rng(3)
    cam = np.zeros((2, 4, 4), F32)
    cam[0] = np.eye(4, dtype=F32); cam[0, :3] = rng.normal(0, 1, (3, 4)).astype(F32)
    cam[1, :3, :3] = np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1]], F32)
    cam[1, 3] = [425.0, 2.5, 192, 902.5]
    fn = str(tmp_path / "00000000_cam.txt")
    write_cam(fn, cam)
    K, E = read_camera_parameters(fn)
    np.testing.assert_array_equal(K, cam[1, :3, :3])
    np.testing.assert_array_equal(E, cam[0])
    lines = open(fn).read().split("\n")
    assert lines[0] == "extrinsic" and lines[6] == "intrinsic" and lines[11].split() == ["425.0", "2.5", "192.0", "902.5"]
    write_cam(fn, cam, cam_near_far=(1.0, 2.0, 3.0, 4.123456))
    assert open(fn).read().split("\n")[11] == "1.0000 2.0000 3.0000 4.1235"


def test_ply_writer_and_reader(tmp_path):
    from svs_hip.fusion import read_ply_points, write_ply
    rng = np.random.default_rng(8)
    xyz = rng.normal(0, 100, (37, 3)).astype(F32)
    rgb = rng.integers(0, 256, (37, 3)).astype(np.uint8)
    fn = str(tmp_path / "c.ply")
    write_ply(fn, xyz, rgb)
    assert open(fn, "rb").read() == forc.ply_bytes(xyz, rgb)
    pts, col = read_ply_points(fn)
    np.testing.assert_array_equal(pts, xyz.astype(np.float64))
    np.testing.assert_array_equal(col, rgb)
    asc = tmp_path / "a.ply"
    asc.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty double x\nproperty double y\n"
                   "property double z\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n1 2 3\n4.5 5 6\n")
    pts, col = read_ply_points(str(asc))
    np.testing.assert_array_equal(pts, [[1, 2, 3], [4.5, 5, 6]])
    assert col is None
    write_ply(fn, np.zeros((0, 3), F32), np.zeros((0, 3), np.uint8))               # empty cloud
    assert read_ply_points(fn)[0].shape == (0, 3)


def load_filter_depth(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "filter_depth.npz")))
    ids = [int(v) for v in g["view_ids"]]
    views = {v: dict(K=g[f"K_{v}"], E=g[f"E_{v}"], depth=g[f"depth_{v}"], confidence=g[f"confidence_{v}"], img=g[f"img_{v}"])
             for v in ids}
    conf = dict(conf=float(g["conf"]), filter_dist=float(g["filter_dist"]), filter_diff=float(g["filter_diff"]),
                thres_view=int(g["thres_view"]))
    return g, ids, views, conf


def test_oracle_filter_depth_matches_reference_end_to_end(golden_dir):
    """`filter_depth` of the reference itself (runner.py:301-404, its source executed by make_fixtures.fx_filter_depth on a
    synthetic scan folder: fixture filter_depth.npz) against the oracle's restatement of its loop: the three masks of every
    view, every vertex, every colour, and the layout of the structured array the function hands to plyfile."""
    g, ids, views, conf = load_filter_depth(golden_dir)
    xyz, rgb = [], []
    for v in ids:
        out = forc.fuse_view(views[v], [views[s] for s in ids if s != v], **conf)
        for tag in ("photo", "geo", "final"):
            assert np.array_equal(out[tag + "_mask"], g[f"mask_{v}_{tag}"]), (v, tag)
        xyz.append(out["xyz"]); rgb.append(out["rgb"])
    xyz, rgb = np.concatenate(xyz), np.concatenate(rgb)
    assert len(xyz) == len(g["vertex_xyz"]) > 1000
    np.testing.assert_array_equal(xyz, g["vertex_xyz"])
    np.testing.assert_array_equal(rgb, g["vertex_rgb"])
    # runner.py:389-400: vertex record = x, y, z (little-endian float32) then red, green, blue (uint8), element 'vertex'
    assert str(g["vertex_descr"]) == "[('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('red', '|u1'), ('green', '|u1'), ('blue', '|u1')]"
    body = forc.ply_bytes(xyz, rgb).split(b"end_header\n", 1)[1]
    rec = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    assert len(rec) == len(xyz) and np.array_equal(rec["z"], g["vertex_xyz"][:, 2]) and np.array_equal(rec["blue"], g["vertex_rgb"][:, 2])


def test_camera_and_image_files_of_the_fixture_read_back(golden_dir, tmp_path):
    """The drop-in readers on the very files the reference read (camera text written by its write_cam, a JPEG)."""
    from helpers.utils import read_camera_parameters, read_img
    g, ids, views, _ = load_filter_depth(golden_dir)
    for v in ids:
        (tmp_path / "cam.txt").write_bytes(g[f"cam_{v}"].tobytes())
        (tmp_path / "im.jpg").write_bytes(g[f"jpg_{v}"].tobytes())
        K, E = read_camera_parameters(str(tmp_path / "cam.txt"))
        np.testing.assert_array_equal(K, views[v]["K"]); np.testing.assert_array_equal(E, views[v]["E"])
        np.testing.assert_array_equal(read_img(str(tmp_path / "im.jpg")), views[v]["img"])
