"""tools/dev/merge_chamfer.py out.json batch1.json batch2.json ...: the runs of several tools/chamfer_parity.py calls (disjoint
seeds, same settings) as one record with the paired statistics over all seeds and per batch."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from chamfer_parity import paired_differences   # noqa: E402

out_path, batches = sys.argv[1], [json.load(open(p)) for p in sys.argv[2:]]
res = {}
for b in batches:
    for path, v in b.items():
        if isinstance(v, dict) and "runs" in v:
            res.setdefault(path, {"runs": []})["runs"] += v["runs"]
for path, v in res.items():
    seeds = [r["seed"] for r in v["runs"]]
    assert len(seeds) == len(set(seeds)), f"{path}: a seed appears in two batches"
    ok = [r["overall_mm"] for r in v["runs"] if "overall_mm" in r]
    v["n"], v["mean_overall_mm"] = len(ok), sum(ok) / max(len(ok), 1)
    v["median_overall_mm"] = sorted(ok)[len(ok) // 2] if ok else None
res["paired"] = paired_differences(res)
res["paired_by_batch"] = [dict(seeds=sorted({r["seed"] for v in b.values() if isinstance(v, dict) and "runs" in v for r in v["runs"]}),
                               **{k: {q: s[q] for q in ("n", "mean_mm", "se_mm")} for k, s in paired_differences(b).items()})
                          for b in batches]
res["prior"], res["what"] = batches[0].get("prior"), batches[0].get("what")
res["note"] = (f"{len(batches)} GPU calls of 12 seeds (36 runs side by side on one MI355X each; round 6).  A seed fixes the initial "
               "weights of all three paths; batches and draws differ between the HIP paths (VolOpt.run: DataLoader, CPU generator) "
               "and the torch comparator (device generator).")
json.dump(res, open(out_path, "w"), indent=1)
for k, s in res["paired"].items():
    print(f"{k}: n = {s['n']}, mean {s['mean_mm']:+.4f} mm, SE {s['se_mm']:.4f}, within 2 SE: {s['within_2_se']}")
for p in ("hip", "hip_f32", "torch_f32"):
    if p in res: print(p, "mean overall", round(res[p]["mean_overall_mm"], 4), "median", round(res[p]["median_overall_mm"], 4), "n", res[p]["n"])
