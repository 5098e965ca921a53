"""Turns the rocprofv3 outputs under gpurun_out/<run>/ into the small summaries committed under profiles/:
  <tag>_train_kernel_stats.csv / <tag>_render_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, as written)
  <tag>_pmc_traffic.json   per-kernel HBM bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes,
                           corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B
                           request of a wide coalesced stream: doubled; WRITE_SIZE as read; both in KiB).
usage: python tools/summarize_profiles.py gpurun_out/r01 r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def agg(pattern, counter):
    d = collections.defaultdict(list)
    files = glob.glob(pattern)
    if not files:
        return d
    for row in csv.DictReader(open(files[0])):
        if row.get("Counter_Name") == counter:
            d[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return d


def main(src, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    for mode in ("train", "render"):
        f = glob.glob(os.path.join(src, mode, "*", "*kernel_stats.csv"))
        if f:
            shutil.copy(f[0], os.path.join(out, f"{tag}_{mode}_kernel_stats.csv"))
    fetch = agg(os.path.join(src, "pmc_fetch", "*", "*counter_collection.csv"), "FETCH_SIZE")
    write = agg(os.path.join(src, "pmc_write", "*", "*counter_collection.csv"), "WRITE_SIZE")
    res = {"_note": "bytes per launch; fetch = 2 * FETCH_SIZE KiB (gfx950 wide-stream correction), write = WRITE_SIZE KiB",
           "kernels": {}}
    for k in fetch:
        if not k.startswith(("svs::", "void svs::")):
            continue
        fb = 2.0 * 1024.0 * sum(fetch[k]) / len(fetch[k])
        wb = 1024.0 * sum(write.get(k, [0.0])) / max(1, len(write.get(k, [])))
        res["kernels"][k.split("(")[0].replace("void ", "")] = {"launches": len(fetch[k]), "fetch_bytes": fb,
                                                              "write_bytes": wb, "hbm_bytes": fb + wb}
    json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in res["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
