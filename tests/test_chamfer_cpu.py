"""CPU check of the Chamfer-evaluator oracle against the fixture produced by running the
reference's evals/eval_dtu.py end to end (tests/golden/make_fixtures.py::fx_chamfer)."""
import os

import numpy as np

import chamfer_oracle as corc
import synth


def test_oracle_matches_reference_script(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "chamfer_ref.npz")))
    sc = synth.make_dtu_scan(int(g["seed"]))
    data_pcd = sc["data_pcd"].copy()
    np.random.default_rng(int(g["shuffle_seed"])).shuffle(data_pcd, axis=0)
    np.testing.assert_array_equal(data_pcd[:64], g["data_pcd_shuffled_head"])
    (acc, comp, overall), d = corc.evaluate_scan(data_pcd, sc["stl"], sc["ObsMask"], sc["BB"], sc["Res"], sc["P"], n_jobs=2)
    assert np.array_equal(d["keep"], np.unpackbits(g["keep"])[:len(data_pcd)].astype(bool))
    assert (len(d["data_down"]), len(d["data_in"]), len(d["data_in_obs"]), len(d["stl_above"])) == \
        (int(g["n_down"]), int(g["n_in"]), int(g["n_in_obs"]), int(g["n_stl_above"]))
    np.testing.assert_array_equal(d["dist_d2s"], g["dist_d2s"])
    np.testing.assert_array_equal(d["dist_s2d"], g["dist_s2d"])
    assert (acc, comp, overall) == (float(g["mean_d2s"]), float(g["mean_s2d"]), float(g["over_all"]))
    # the scene exercises every branch of the protocol
    assert int(g["n_down"]) < len(data_pcd) * 0.8 and int(g["n_in"]) < int(g["n_down"]) and int(g["n_in_obs"]) < int(g["n_in"])
    assert (g["dist_d2s"] >= 20).any() and int(g["n_stl_above"]) < len(sc["stl"])


def test_mesh_sampler_oracle_matches_reference_script(golden_dir):
    """--mode mesh: the oracle's triangle sampler against the script's own points (fixture chamfer_mesh_ref.npz), then the
    rest of the protocol on the sampled cloud."""
    g = dict(np.load(os.path.join(golden_dir, "chamfer_mesh_ref.npz")))
    sc = synth.make_dtu_scan(int(g["scan_seed"]))
    vertices, triangles = synth.make_dtu_mesh(int(g["mesh_seed"]))
    data_pcd, per_tri = corc.sample_mesh(vertices, triangles, 0.2)
    new_pts = data_pcd[len(vertices):]
    assert len(new_pts) == int(g["n_new_pts"]) and np.array_equal(per_tri, g["per_tri"])
    assert len(per_tri) == len(triangles) - 3                        # the three degenerate triangles are dropped
    np.testing.assert_array_equal(new_pts[::61], g["new_pts_every_61"])
    np.testing.assert_array_equal(new_pts.sum(0), g["new_pts_sum"])
    np.testing.assert_array_equal(data_pcd[:len(vertices)], vertices)
    np.random.default_rng(int(g["shuffle_seed"])).shuffle(data_pcd, axis=0)
    np.testing.assert_array_equal(data_pcd[:64], g["data_pcd_shuffled_head"])
    (acc, comp, overall), d = corc.evaluate_scan(data_pcd, sc["stl"], sc["ObsMask"], sc["BB"], sc["Res"], sc["P"], n_jobs=2)
    assert np.array_equal(d["keep"], np.unpackbits(g["keep"])[:len(data_pcd)].astype(bool))
    np.testing.assert_array_equal(d["dist_d2s"], g["dist_d2s"])
    np.testing.assert_array_equal(d["dist_s2d"], g["dist_s2d"])
    assert (acc, comp, overall) == (float(g["mean_d2s"]), float(g["mean_s2d"]), float(g["over_all"]))


def _write_mesh_ply(fn, vertices, faces, binary, extra_vertex_prop=False):
    """A PLY with a face element the way mesh tools write it (list uchar int vertex_indices)."""
    with open(fn, "wb") as f:
        head = "ply\nformat %s 1.0\ncomment made by a test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % (
            "binary_little_endian" if binary else "ascii", len(vertices))
        if extra_vertex_prop:
            head += "property uchar quality\n"
        head += "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % len(faces)
        f.write(head.encode())
        for v in vertices:
            if binary:
                f.write(np.asarray(v, "<f4").tobytes() + (b"\x07" if extra_vertex_prop else b""))
            else:
                f.write((" ".join(repr(float(np.float32(x))) for x in v) + (" 7" if extra_vertex_prop else "") + "\n").encode())
        for face in faces:
            if binary:
                f.write(bytes([len(face)]) + np.asarray(face, "<i4").tobytes())
            else:
                f.write((" ".join(str(x) for x in [len(face)] + list(face)) + "\n").encode())


def test_read_ply_mesh(tmp_path):
    """The mesh reader behind --mode mesh: ascii and binary, an extra vertex property, a quad (fanned into two triangles)."""
    from svs_hip.fusion import read_ply_mesh, read_ply_points
    rng = np.random.default_rng(5)
    vertices = rng.normal(0, 10, (9, 3)).astype(np.float32)
    faces = [(0, 1, 2), (2, 3, 4, 5), (6, 7, 8), (8, 0, 4)]
    want = np.array([(0, 1, 2), (2, 3, 4), (2, 4, 5), (6, 7, 8), (8, 0, 4)])
    for binary in (False, True):
        for extra in (False, True):
            fn = str(tmp_path / ("m%d%d.ply" % (binary, extra)))
            _write_mesh_ply(fn, vertices, faces, binary, extra)
            v, t = read_ply_mesh(fn)
            assert v.dtype == np.float64 and t.dtype == np.int64
            np.testing.assert_array_equal(v, vertices.astype(np.float64))
            np.testing.assert_array_equal(t, want)
            np.testing.assert_array_equal(read_ply_points(fn)[0], v)       # the point reader skips the faces
    fn = str(tmp_path / "nofaces.ply")
    _write_mesh_ply(fn, vertices, [], True)
    v, t = read_ply_mesh(fn)
    assert t.shape == (0, 3) and len(v) == 9
