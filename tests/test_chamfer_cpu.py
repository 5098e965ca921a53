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
