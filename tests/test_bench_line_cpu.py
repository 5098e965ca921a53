"""bench.py's output contract (host logic, no GPU): the LAST stdout line is one compact strict-JSON object the driver can parse
(round 5 lost its record to a 21 KB line), the full record goes to the extras file, and a roofline fraction outside (0, 1] is
flagged instead of printed as if it were a measurement."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("svs_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _row(i, frac=0.5, bound="hbm"):
    return {"kernel": f"svs::mlp::kernel_{i}", "what": "a sweep " * 12, "bound": bound, "launches_per_step": 1.0, "kernel_ms": 0.7 - 0.01 * i,
            "ms_per_step": 0.7 - 0.01 * i, "points_per_launch": 102400.0, "work_per_point": 34816, "achieved": 8000.0 * frac,
            "unit": "GB/s", "frac": frac, "kernel_ms_in_line": 0.6, "frac_in_line": 0.6}


def _full(n_rows=18, frac=0.5):
    rows = [_row(i, frac if i == 3 else 0.5) for i in range(n_rows)]
    top = dict(rows[0], peak=8000.0, traffic=3.6e9, traffic_source="r05_pmc_traffic.json", timing="x" * 300, peak_note="y" * 300,
               kernels=rows, entry="svs_wgrad_multi")
    return {"metric": "rendered rays/sec (1024-ray batch, 128 samples)", "value": 290811.1664458961, "unit": "rays/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 3.5211852849897696, "host_enqueue_ms_per_step": 1.7, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (fp16x2 ...)", "data": "synthetic",
            "config": {"workload": "configs[1]: " + "w" * 200, "mode": "train", "mlp_precision": "fp16x2: " + "p" * 200,
                       "ray_groups": [[0, 976], [976, 1024]], "ray_group_schedule": {"choice": "split"}, "launch": "eager launches",
                       "launch_plan": None, "rays_per_gpu": 1024, "rays_total": 1024, "settle_steps": 280, "flop_per_ray": 920586240,
                       "model_flops_per_s": 2.67e14},
            "roofline": top,
            "cpu_baseline": {"value": 161.4, "unit": "rays/s", "cores": 32, "kind": "port", "sample": "256 rays ...", "sample_rays": 256,
                             "runs": [{"threads": t, "rays": 256, "rays_per_s": v, "median_s": 1.0, "reps": 3, "warmups": 1}
                                      for t, v in ((1, 84.9), (32, 161.4), (128, 72.0))],
                             "host_cpu": "AMD EPYC", "host_threads": 256, "note": "n" * 500},
            "gpu_torch_baseline": {"value": 26055.7, "unit": "rays/s", "ms_per_step": 39.3, "ratio_value_over_baseline": 11.16, "what": "z" * 400},
            "costvol": {"stage1_ms": 0.87, "roofline": [{"kernel": "CostRegNet (11 launches)", "stage": 1, "bound": "mfma", "frac": 0.14},
                                                        {"kernel": "svs::costvol::warp", "stage": 1, "bound": "hbm", "frac": 0.31}],
                        "workload": "configs[2]"},
            "render_eval": {"image": [576, 768], "render_image_rays_per_s": 707221.6},
            "chamfer_parity": {"runs": {"hip": [{"seed": s, "overall_mm": 1.0} for s in range(64)]}, "note": "c" * 4000}}


def test_final_line_is_compact_strict_json(tmp_path, capsys):
    b = _bench()
    extras = tmp_path / "extras.json"
    b.emit(_full(), str(extras))
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1                                   # ONE line on stdout
    text = out[-1]
    assert len(text) < b.MAX_LINE < 8192
    d = json.loads(text, parse_constant=lambda c: pytest.fail("non-finite constant " + c))
    assert json.loads(json.dumps(d, allow_nan=False)) == d
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert 1 <= len(rf["kernels"]) <= 6
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["cpu_baseline"]["by_threads"] == {"1": 84.9, "32": 161.4, "128": 72.0}
    assert "chamfer_parity" not in d and "self_test_failed" not in d
    full = json.load(open(extras))                         # nothing is lost: the full record is in the extras file
    assert len(full["roofline"]["kernels"]) == 18 and "chamfer_parity" in full


def test_fraction_above_one_is_flagged(tmp_path, capsys):
    b = _bench()
    b.emit(_full(frac=1.494), str(tmp_path / "e.json"))
    cap = capsys.readouterr()
    d = json.loads(cap.out.strip().splitlines()[-1])
    assert "self_test_failed" in d and "1.494" in d["self_test_failed"]
    assert "SELF-TEST FAILED" in cap.err
    with pytest.raises(AssertionError):
        b.check_line(json.dumps(b.compact_line(_full(frac=1.494))))


def test_check_line_rejects_nan():
    b = _bench()
    with pytest.raises(ValueError):
        b.check_line('{"metric": "m", "value": NaN, "roofline": null}')
