#!/bin/bash
# counters of the sampler's sdf-only evaluation for the three tilings (dev aid; run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_sdf_only
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
# the selector ops.PackedMlp reads is SVS_SDF_KERNEL = 32 | 16 | pair; the two-wave variants need a library built with
# SVS_BUILD_EXPERIMENTS=1 (python s-volsdf_amd/build.py --force).  The assignment is a shell prefix, the program comes
# directly after `--` (no env / bash -c hop under the profiler), every run is bounded by `timeout`.
SVS_SDF_KERNEL=32 timeout 180 rocprofv3 --pmc $P -d $O/t32 --output-format csv -- python3 $R/tools/dev/time_sdf_only.py > $O/t32.log 2>&1
SVS_SDF_KERNEL=16 timeout 180 rocprofv3 --pmc $P -d $O/t16 --output-format csv -- python3 $R/tools/dev/time_sdf_only.py > $O/t16.log 2>&1
SVS_SDF_KERNEL=pair timeout 180 rocprofv3 --pmc $P -d $O/pair --output-format csv -- python3 $R/tools/dev/time_sdf_only.py > $O/pair.log 2>&1
cd $R
find $O -name '*agent_info.csv' -delete
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_sdf_only")
for v in ("t32", "t16", "pair"):
    f = glob.glob(f"{O}/{v}/*/*counter_collection.csv")
    if not f: print(v, "no csv"); continue
    d = collections.defaultdict(list); dur = []
    for row in csv.DictReader(open(f[0])):
        if "sdf_only" not in row["Kernel_Name"]: continue      # sdf_only_h2_kernel, sdf_only_w16_kernel, sdf_only_kp_kernel
        d[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in row:
            dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    m = {k: sum(x[-100:]) / len(x[-100:]) for k, x in d.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    line = f"{v}: kernel cycles {cyc:.0f}  mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}  "
    w = m["SQ_WAVE_CYCLES"]
    line += f"wave: wait_any {m['SQ_WAIT_ANY'] / w:.2f} wait_inst {m['SQ_WAIT_INST_ANY'] / w:.2f} active {m['SQ_ACTIVE_INST_ANY'] / w:.2f} wait_lds {m['SQ_WAIT_INST_LDS'] / w:.2f}"
    if dur: line += f"  dur_us {sum(dur[-100:]) / len(dur[-100:]) / 1e3:.1f} clock_GHz {cyc / (sum(dur[-100:]) / len(dur[-100:])):.2f}"
    print(line)
PY
