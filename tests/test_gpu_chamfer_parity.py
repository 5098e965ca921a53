"""Chamfer parity on a scene with known geometry (BASELINE.json's metric: "...; Chamfer parity"), short variant of
tools/chamfer_parity.py: the analytic sphere + box scene optimised through `VolOpt.run` on the HIP kernels and, as comparator,
by plain PyTorch float32 autograd (oracle/torch_ref.py) on the same GPU; both go through render_mvs -> filter_depth ->
evals.eval_dtu.evaluate_scan against the analytic surface.  The long runs (3000 steps, two seeds per path) are
profiles/r05_chamfer_parity.json."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_chamfer_parity_short():
    """600 steps per path WITH the synthetic MVS prior (the reference's regime: MVS term + annealed sparsity + rgb_smooth,
    chamfer_parity.build_prior): the reconstruction goes from the geometric initialisation (a sphere of radius 120 mm, ~10 mm
    off the scene) to ~1 mm, and the HIP path's Chamfer distance sits with the float32 torch path's: inside the larger of the
    paths' seed-to-seed spread and 0.35 mm (a third of the value: at 600 steps two seeds of ONE path differ by that much)."""
    assert torch.cuda.is_available()
    import chamfer_parity
    res = chamfer_parity.measure(steps=600, seeds=(0, 1), paths=("hip", "torch_f32"), rays=512, timeout=900, prior=True)
    for p in ("hip", "torch_f32"):
        assert all("overall_mm" in r for r in res[p]["runs"]), res[p]["runs"]
        assert res[p]["runs"][0]["n_fused"] > 5000
    hip, ref = res["hip"]["overall_mm"], res["torch_f32"]["overall_mm"]
    band = max(res["spread_mm"], 0.35)
    print(f"chamfer parity (600 steps, MVS prior): hip {hip:.3f} mm, torch float32 {ref:.3f} mm, seed spread {res['spread_mm']:.3f} mm")
    assert abs(hip - ref) <= band, (hip, ref, band)
    assert hip < 2.0 and ref < 2.0       # (the untrained initialisation scores ~10 mm, a run without the prior 6-8)
