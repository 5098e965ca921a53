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


def flat(pattern, counter, kernel):
    """[(grid size, value)] of one kernel's launches in dispatch order"""
    f = newest(pattern)
    if not f:
        return []
    rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == counter and r["Kernel_Name"] == kernel]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(int(r["Grid_Size"]), float(r["Counter_Value"])) for r in rows]


def main(src, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    for mode in ("train", "render", "train_onegroup", "train_half", "train_256", "costvol", "evalrender", "featurenet"):
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
    traffic(src, out, tag, "", "pmc_traffic")
    traffic(src, out, tag, "_half", "half_pmc_traffic")       # SVS_MLP_PRECISION=f16x2_half: one-piece gradient blocks
    mfma_summary(src, out, tag)


def traffic(src, out, tag, suffix, name):
    fetch = agg(os.path.join(src, "pmc_fetch" + suffix, "*", "*counter_collection.csv"), "FETCH_SIZE", by_grid=True)
    write = agg(os.path.join(src, "pmc_write" + suffix, "*", "*counter_collection.csv"), "WRITE_SIZE", by_grid=True)
    if not fetch:
        return
    res = {"_note": "bytes per launch; fetch = 2 * FETCH_SIZE KiB (gfx950 wide-stream correction), write = WRITE_SIZE KiB.  "
                    "The default step runs two ray groups, so most kernels are launched in two sizes, and the weight-gradient "
                    "GEMM four times with (nearly) one grid size (~one workgroup per CU whatever the batch: SDF / radiance x "
                    "two groups).  Launches are therefore grouped by their POSITION in the step (the k-th launch of the kernel in "
                    "every step), not by grid size -- round 2's summary averaged the two ray groups' weight-gradient launches "
                    "together, which is where its 0.965 GB against 1.70 GB algorithmic came from.  hbm_bytes = the launch position "
                    "with the most bytes; by_launch lists every position",
           "kernels": {}}
    mean = lambda v: sum(v) / max(1, len(v))
    steps = 5            # bench.py --steps 3 --warmup 2
    for k in fetch:
        if not k.startswith(("svs::", "void svs::")):
            continue
        # per-launch values in dispatch order (the by_grid dict keeps insertion order within a grid: rebuild the flat order)
        fl = flat(os.path.join(src, "pmc_fetch" + suffix, "*", "*counter_collection.csv"), "FETCH_SIZE", k)
        wl = flat(os.path.join(src, "pmc_write" + suffix, "*", "*counter_collection.csv"), "WRITE_SIZE", k)
        per_step = max(1, len(fl) // steps)
        shapes = {}
        for pos in range(per_step):
            fv = [v for (g, v) in fl[pos::per_step]]
            wv = [v for (g, v) in wl[pos::per_step]] or [0.0]
            fb, wb = 2.0 * 1024.0 * mean(fv), 1024.0 * mean(wv)
            shapes[pos] = {"launches": len(fv), "grid_size": fl[pos][0], "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
        top_pos = max(shapes, key=lambda q: shapes[q]["hbm_bytes"])
        res["kernels"][k.split("(")[0].replace("void ", "")] = dict(shapes[top_pos], launch_position=top_pos,
                                                                  by_launch={str(q): shapes[q] for q in sorted(shapes)})
    json.dump(res, open(os.path.join(out, f"{tag}_{name}.json"), "w"), indent=1, sort_keys=True)
    print(name, json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in res["kernels"].items()}, indent=1))


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
