"""The analytic scene of the Chamfer parity runs (tests/golden/synth.py, tests/synthetic_scene.py::AnalyticSceneDataset) and
the synthetic MVS prior built from it (tools/chamfer_parity.py::build_prior): what they claim to be.  No GPU involved."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_analytic_scene_is_what_it_says():
    """the sphere tracer's hit points lie on the analytic surface, depths are the z of the camera frame, views overlap"""
    import numpy as np
    import synth
    import synthetic_scene
    ds = synthetic_scene.AnalyticSceneDataset(img_res=(48, 64))
    for i in ds.trains_ids():
        r = ds.renders[i]
        assert 0.1 < r["mask"].mean() < 0.3 and np.abs(synth.analytic_sdf(r["points"])).max() < 1e-5
        pose = ds.pose_all[i].numpy().astype(np.float64)
        zc = ((r["points"] - pose[:3, 3]) @ pose[:3, :3])[:, 2]
        np.testing.assert_allclose(zc, r["depth"][r["mask"]], rtol=1e-6)       # (the poses are float32: orthonormal to 1e-7)
        assert float(ds.rgb_images[i][r["mask"].reshape(-1)].mean()) > 0.2 and float(ds.rgb_images[i][~r["mask"].reshape(-1)].max()) == 0.0



def test_synthetic_prior_is_what_it_says():
    """the prior of chamfer_parity.build_prior: probabilities sum to 1 over the planes on object pixels and vanish elsewhere,
    the expected depth is the analytic depth to a fraction of a plane, hypotheses are in MVS units"""
    import numpy as np
    import chamfer_parity
    import synthetic_scene
    ds = synthetic_scene.AnalyticSceneDataset(img_res=(48, 64))
    outs = chamfer_parity.build_prior(ds)
    assert len(outs) == 3
    for i, o in zip(ds.trains_ids(), outs):
        p, z = o["prob_volume"][0].numpy(), o["depth_values"][0].numpy()
        assert p.shape == z.shape == (chamfer_parity.PRIOR_D, 24, 32)
        s = p.sum(0)
        hit = ds.renders[i]["mask"].reshape(24, 2, 32, 2).any((1, 3))
        assert np.allclose(s[hit], 1.0, atol=1e-5) and np.all(s[~hit] == 0)
        full = ds.renders[i]["mask"].reshape(24, 2, 32, 2).all((1, 3))
        d = ds.renders[i]["depth"].reshape(24, 2, 32, 2).mean((1, 3))
        step = (chamfer_parity.PRIOR_Z[1] - chamfer_parity.PRIOR_Z[0]) / (chamfer_parity.PRIOR_D - 1)
        assert np.abs((p * z).sum(0)[full] / chamfer_parity.MM - d[full]).max() < 0.6 * step
