"""Turns the rocprofv3 outputs under gpurun_out/<run>/ into the small summaries committed under profiles/:
  <tag>_train_kernel_stats.csv / <tag>_render_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, as written)
  <tag>_train_onegroup_kernel_stats.csv                           (the same step as ONE ray group: a launch = a batch)
  <tag>_{costvol,evalrender,featurenet}_kernel_stats.csv + _result.txt   (tools/bench_costvol.py = config 3, tools/
                           bench_render_eval.py = a 768x576 eval render with fast = -1, tools/bench_featurenet.py)
  <tag>_pmc_mfma.json      matrix-core utilisation + wave-time split from the SQ counters (one-group step)
  <tag>_pmc_traffic.json   per-kernel HBM bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes,
                           corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B
                           request of a wide coalesced stream: doubled; WRITE_SIZE as read; both in KiB).
usage: bash tools/profile_round.sh r01 (on the GPU box), then python tools/summarize_profiles.py gpurun_out/r01 r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the most recent match."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def agg(pattern, counter, by_grid=False):
    """kernel -> [values] (by_grid: kernel -> {grid size -> [values]})"""
    d = collections.defaultdict(lambda: collections.defaultdict(list)) if by_grid else collections.defaultdict(list)
    f = newest(pattern)
    if not f:
        return d
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") == counter:
            if by_grid:
                d[row["Kernel_Name"]][int(row["Grid_Size"])].append(float(row["Counter_Value"]))
            else:
                d[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return d


def main(src, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    for mode in ("train", "render", "train_onegroup", "costvol", "evalrender", "featurenet"):
        f = newest(os.path.join(src, mode, "*", "*kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(out, f"{tag}_{mode}_kernel_stats.csv"))
    # the JSON lines the secondary benches printed under the profiler
    for name in ("costvol", "evalrender", "featurenet"):
        log = os.path.join(src, name + ".log")
        if os.path.exists(log):
            lines = [ln.strip() for ln in open(log) if ln.startswith("{") or ln.startswith("FeatureNet")]
            if lines:
                open(os.path.join(out, f"{tag}_{name}_result.txt"), "w").write("\n".join(lines) + "\n")
    fetch = agg(os.path.join(src, "pmc_fetch", "*", "*counter_collection.csv"), "FETCH_SIZE", by_grid=True)
    write = agg(os.path.join(src, "pmc_write", "*", "*counter_collection.csv"), "WRITE_SIZE", by_grid=True)
    res = {"_note": "bytes per launch; fetch = 2 * FETCH_SIZE KiB (gfx950 wide-stream correction), write = WRITE_SIZE KiB.  "
                    "The default step runs two ray groups, so most kernels are launched in two sizes: hbm_bytes is the "
                    "figure of the LARGEST launch shape (grid size), by_grid_size lists every shape",
           "kernels": {}}
    mean = lambda v: sum(v) / max(1, len(v))
    for k in fetch:
        if not k.startswith(("svs::", "void svs::")):
            continue
        shapes = {}
        for grid, vals in fetch[k].items():
            fb = 2.0 * 1024.0 * mean(vals)
            wb = 1024.0 * mean(write.get(k, {}).get(grid, [0.0]))
            shapes[grid] = {"launches": len(vals), "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
        top = shapes[max(shapes)]
        res["kernels"][k.split("(")[0].replace("void ", "")] = dict(top, grid_size=max(shapes),
                                                                  by_grid_size={str(g): shapes[g] for g in sorted(shapes)})
    json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in res["kernels"].items()}, indent=1))
    mfma_summary(src, out, tag)


MFMA_COUNTERS = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                 "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE")


def mfma_summary(src, out, tag):
    """<tag>_pmc_mfma.json: matrix-core utilisation and the split of wave time per kernel (one-group step)."""
    pat = os.path.join(src, "pmc_mfma", "*", "*counter_collection.csv")
    c = {name: agg(pat, name) for name in MFMA_COUNTERS}
    if not c["GRBM_GUI_ACTIVE"]:
        return
    res = {"_note": "rocprofv3 --pmc " + " ".join(MFMA_COUNTERS) + " -- python3 bench.py --steps 3 --warmup 2 "
           "--no-cpu-baseline --groups none; per-launch averages. kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; mfma_util = "
           "SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024 SIMDs); parked / issue_stall / issuing = SQ_WAIT_ANY / "
           "SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES", "kernels": {}}
    mean = lambda v: sum(v) / max(1, len(v))
    for k, gui in c["GRBM_GUI_ACTIVE"].items():
        if not k.startswith(("svs::mlp", "void svs::mlp", "svs::wgrad", "void svs::wgrad")):
            continue
        cyc = mean(gui) / 8.0
        wave = mean(c["SQ_WAVE_CYCLES"][k]) or 1.0
        res["kernels"][k.split("(")[0].replace("void ", "")] = {
            "launches": len(gui), "kernel_cycles": int(cyc),
            "mfma_util": round(mean(c["SQ_VALU_MFMA_BUSY_CYCLES"][k]) / (cyc * 1024.0), 3),
            "parked": round(mean(c["SQ_WAIT_ANY"][k]) / wave, 3),
            "issue_stall": round(mean(c["SQ_WAIT_INST_ANY"][k]) / wave, 3),
            "issuing": round(mean(c["SQ_ACTIVE_INST_ANY"][k]) / wave, 3),
            "lds_issue_stall": round(mean(c["SQ_WAIT_INST_LDS"][k]) / wave, 3)}
    json.dump(res, open(os.path.join(out, f"{tag}_pmc_mfma.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v["mfma_util"] for k, v in res["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
