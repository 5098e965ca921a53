"""Chamfer parity on a scene with known geometry (BASELINE.json's metric: "...; Chamfer parity"), short variant of
tools/chamfer_parity.py: the analytic sphere + box scene optimised through `VolOpt.run` on the HIP kernels and, as comparator,
by plain PyTorch float32 autograd (oracle/torch_ref.py) on the same GPU; both go through render_mvs -> filter_depth ->
evals.eval_dtu.evaluate_scan against the analytic surface.

The parity STATEMENT is the long study, not this file: profiles/r06_chamfer_paired.json -- 3000 steps, seeds paired by their
initial weights, hip - torch_f32 = +0.08 +- 0.07 mm (n = 12).  What runs here, with a fixed amount of work and no retries:
(1) a deterministic run (SVS_DETERMINISTIC=1) repeats EXACTLY, the property a bisection of a Chamfer difference needs;
(2) a regression bound on short runs of fixed seeds."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_chamfer_deterministic_run_repeats():
    """The same seed twice through the whole pipeline in deterministic mode, one process after the other: 300 optimisation
    steps, render, fuse, evaluate -- the fused cloud has the same number of points and the same accuracy / completeness to the
    last bit.  (The default mode gives 6.39 and 7.26 mm for one seed in two runs without a prior: float atomics, NOTES/
    design_history_r01-r05.md section 2.  One at a time: workgroups that spin for their turn and two processes time-sliced on
    one GPU do not mix -- the same two runs side by side took 29 s in one call and 190 s in another.)"""
    assert torch.cuda.is_available()
    import chamfer_parity
    res = chamfer_parity.measure(steps=300, seeds=(0, 0), paths=("hip_det",), rays=512, timeout=900, prior=True, parallel=False)
    a, b = res["hip_det"]["runs"]
    assert "overall_mm" in a and "overall_mm" in b, (a, b)
    print("deterministic runs:", {k: (a[k], b[k]) for k in ("accuracy_mm", "completeness_mm", "n_fused", "beta")})
    for k in ("accuracy_mm", "completeness_mm", "overall_mm", "n_fused", "beta"):
        assert a[k] == b[k], (k, a[k], b[k])


def test_chamfer_parity_short():
    """600 steps per path WITH the synthetic MVS prior (the reference's regime: MVS term + annealed sparsity + rgb_smooth,
    chamfer_parity.build_prior), four fixed seeds per path, all eight runs side by side: the reconstruction goes from the
    geometric initialisation (a sphere of radius 120 mm, ~10 mm off the scene) to ~1 mm.  At 600 steps single runs of EITHER
    path scatter between 0.8 and 2.4 mm (the odd run is still converging), so this is a REGRESSION bound -- every run below
    5 mm, both means below 2 mm, the means within 0.6 mm of each other: a broken backward or prior look-up costs millimetres --
    and the paired difference is printed, not asserted.  No retries, no widening: the work is fixed up front."""
    assert torch.cuda.is_available()
    import chamfer_parity
    seeds = (0, 1, 2, 3)
    res = chamfer_parity.measure(steps=600, seeds=seeds, paths=("hip", "torch_f32"), rays=512, timeout=900, prior=True, parallel=True)
    for p in ("hip", "torch_f32"):
        assert all("overall_mm" in r for r in res[p]["runs"]), res[p]["runs"]
        assert res[p]["runs"][0]["n_fused"] > 5000
    hip = [r["overall_mm"] for r in res["hip"]["runs"]]
    ref = [r["overall_mm"] for r in res["torch_f32"]["runs"]]
    mean = lambda v: sum(v) / len(v)
    pd = res["paired"]["hip_minus_torch_f32"]
    print(f"chamfer (600 steps, MVS prior): hip {[round(v, 3) for v in hip]} mm (mean {mean(hip):.3f}), torch float32 "
          f"{[round(v, 3) for v in ref]} mm (mean {mean(ref):.3f}); paired difference {pd['mean_mm']:+.3f} +- {pd['se_mm']:.3f} mm")
    assert max(hip + ref) < 5.0                            # every single run has left the initialisation far behind
    assert mean(hip) < 2.0 and mean(ref) < 2.0             # (the untrained initialisation scores ~10 mm, a run without the prior 6-8)
    assert abs(mean(hip) - mean(ref)) < 0.6
