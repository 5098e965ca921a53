"""The paired statistics behind the Chamfer parity statement (tools/chamfer_parity.py::paired_differences) and the batch
merger (tools/dev/merge_chamfer.py) on hand-made records -- no GPU, no training."""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _runs(seeds, values):
    return {"runs": [{"seed": s, "overall_mm": v} for s, v in zip(seeds, values)]}


def test_paired_differences_by_seed():
    from chamfer_parity import paired_differences
    res = {"hip": _runs([0, 1, 2, 3], [1.0, 1.2, 0.9, 1.1]),
           "torch_f32": _runs([3, 2, 1, 0], [1.0, 1.0, 1.0, 1.0]),          # another order: pairing is by seed
           "hip_f32": _runs([0, 1, 7], [1.5, 0.5, 9.0]),                     # seed 7 has no partner: dropped
           "prior": True}
    out = paired_differences(res)
    a = out["hip_minus_torch_f32"]
    assert a["n"] == 4 and a["differences_mm"] == [1.0 - 1.0, 1.2 - 1.0, 0.9 - 1.0, 1.1 - 1.0]
    mean = 0.05
    sd = math.sqrt(sum((d - mean) ** 2 for d in a["differences_mm"]) / 3)
    assert abs(a["mean_mm"] - mean) < 1e-12 and abs(a["sd_mm"] - sd) < 1e-12 and abs(a["se_mm"] - sd / 2) < 1e-12
    assert a["within_2_se"] is True
    b = out["hip_f32_minus_torch_f32"]
    assert b["n"] == 2 and abs(b["mean_mm"]) < 1e-12


def test_paired_differences_flags_a_real_offset():
    from chamfer_parity import paired_differences
    res = {"hip": _runs(range(8), [1.30, 1.31, 1.29, 1.30, 1.32, 1.28, 1.30, 1.31]),
           "torch_f32": _runs(range(8), [1.0] * 8)}
    a = paired_differences(res)["hip_minus_torch_f32"]
    assert a["within_2_se"] is False and a["mean_over_se"] > 10


def test_merge_chamfer_batches(tmp_path):
    b1 = {"hip": _runs([0, 1], [1.0, 1.2]), "torch_f32": _runs([0, 1], [1.1, 1.1]), "prior": True, "what": "w"}
    b2 = {"hip": _runs([2, 3], [0.9, 1.3]), "torch_f32": _runs([2, 3], [1.0, 1.2]), "prior": True, "what": "w"}
    p1, p2, out = tmp_path / "b1.json", tmp_path / "b2.json", tmp_path / "out.json"
    p1.write_text(json.dumps(b1)); p2.write_text(json.dumps(b2))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "merge_chamfer.py"), str(out), str(p1), str(p2)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = json.loads(out.read_text())
    assert m["hip"]["n"] == 4 and abs(m["hip"]["mean_overall_mm"] - 1.1) < 1e-12
    a = m["paired"]["hip_minus_torch_f32"]
    assert a["n"] == 4 and abs(a["mean_mm"] - 0.0) < 1e-12
    assert [b["seeds"] for b in m["paired_by_batch"]] == [[0, 1], [2, 3]]
    # a seed in two batches is an error, not a silent double count
    p2.write_text(json.dumps({"hip": _runs([1, 3], [0.9, 1.3]), "torch_f32": _runs([1, 3], [1.0, 1.2])}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "merge_chamfer.py"), str(out), str(p1), str(p2)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
