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


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def test_chamfer_parity_short():
    """600 steps per path WITH the synthetic MVS prior (the reference's regime: MVS term + annealed sparsity + rgb_smooth,
    chamfer_parity.build_prior): the reconstruction goes from the geometric initialisation (a sphere of radius 120 mm, ~10 mm
    off the scene) to ~1 mm.  At 600 steps single runs of EITHER path scatter between 0.8 and 1.5 mm, the odd one up to 2.4
    (tools/dev/chamfer_600_distribution.py: six seeds per path; the HIP path is not even repeatable for one seed, its float atomics
    order the weight-gradient sums differently from run to run) -- so the statement is about medians of three seeds: the HIP
    path's median lies inside the float32 torch path's own seed-to-seed range widened by 0.35 mm (two more HIP seeds are run
    before the test gives up: a median of three is thrown by two slow runs in ~3 % of the cases).  The 3000-step runs, where the
    scatter is 0.06-0.15 mm, are profiles/r05_chamfer_parity_prior.json / r05_chamfer_prior_seeds.json."""
    assert torch.cuda.is_available()
    import chamfer_parity
    res = chamfer_parity.measure(steps=600, seeds=(0, 1, 2), paths=("hip", "torch_f32"), rays=512, timeout=900, prior=True, parallel=True)
    for p in ("hip", "torch_f32"):
        assert all("overall_mm" in r for r in res[p]["runs"]), res[p]["runs"]
        assert res[p]["runs"][0]["n_fused"] > 5000
    hip = [r["overall_mm"] for r in res["hip"]["runs"]]
    ref = [r["overall_mm"] for r in res["torch_f32"]["runs"]]
    lo, hi = min(ref) - 0.35, max(ref) + 0.35
    if not lo <= _median(hip) <= hi:
        more = chamfer_parity.measure(steps=600, seeds=(3, 4), paths=("hip",), rays=512, timeout=900, prior=True, parallel=True)
        hip += [r["overall_mm"] for r in more["hip"]["runs"]]
    print(f"chamfer parity (600 steps, MVS prior): hip {[round(v, 3) for v in hip]} mm (median {_median(hip):.3f}), torch float32 "
          f"{[round(v, 3) for v in ref]} mm (median {_median(ref):.3f})")
    assert lo <= _median(hip) <= hi, (hip, ref)
    assert _median(hip) < 2.0 and _median(ref) < 2.0       # (the untrained initialisation scores ~10 mm, a run without the prior 6-8)
    assert max(hip + ref) < 5.0                            # every single run has left the initialisation far behind
