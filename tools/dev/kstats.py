"""dev aid: per-step kernel table from a rocprofv3 --stats directory of bench.py:  python tools/dev/kstats.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = [int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]][0]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("steps", steps, "kernel time per step %.3f ms" % (tot / steps / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%-78s %5.1f/step %8.1f us %8.1f us/step" % (r["Name"][:78], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                       float(r["TotalDurationNs"]) / steps / 1e3))
