#!/bin/bash
# counters of the fused conv0 (dev aid; run on the GPU box): clock, MFMA busy, LDS bank conflicts
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_conv0
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"
for v in "" $VARIANTS; do
  n=${v:-default}
  if [ -n "$v" ]; then export SVS_LIB_PATH=$R/s-volsdf_amd/lib_ab/libsvolsdf_hip_$v.so; fi
  timeout 180 rocprofv3 --pmc $P1 -d $O/${n}_1 --output-format csv -- python3 $R/tools/dev/time_conv0.py > $O/$n.log 2>&1
  timeout 180 rocprofv3 --pmc $P2 -d $O/${n}_2 --output-format csv -- python3 $R/tools/dev/time_conv0.py > $O/$n.log 2>&1
done
cd $R
find $O -name '*agent_info.csv' -delete
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_conv0")
for d1 in sorted(glob.glob(f"{O}/*_1")):
    v = os.path.basename(d1)[:-2]
    m = {}
    for p in (1, 2):
        f = glob.glob(f"{O}/{v}_{p}/*/*counter_collection.csv")
        if not f: continue
        d = collections.defaultdict(list); dur = []
        for row in csv.DictReader(open(f[0])):
            if "conv3d_pair_kernel<32" not in row["Kernel_Name"]: continue
            d[row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in row:
                dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        for k, x in d.items(): m[(k, p)] = sum(x[-8:]) / len(x[-8:])
        if dur: m[("dur", p)] = sum(dur[-8:]) / len(dur[-8:])
    g = lambda k, p: m.get((k, p), float("nan"))
    cyc = g("GRBM_GUI_ACTIVE", 1) / 8
    w = g("SQ_WAVE_CYCLES", 1)
    print(f"{v}: cycles {cyc:.0f} dur_us {g('dur', 1) / 1e3:.1f} clock_GHz {cyc / g('dur', 1):.2f} mfma_busy {g('SQ_VALU_MFMA_BUSY_CYCLES', 1) / (cyc * 1024):.3f} "
          f"wait_inst {g('SQ_WAIT_INST_ANY', 1) / w:.2f} active {g('SQ_ACTIVE_INST_ANY', 1) / w:.2f} wait_lds {g('SQ_WAIT_INST_LDS', 1) / w:.2f} | "
          f"lds_insts {g('SQ_INSTS_LDS', 2):.3g} idx_active {g('SQ_LDS_IDX_ACTIVE', 2):.3g} bank_conflict {g('SQ_LDS_BANK_CONFLICT', 2):.3g} unaligned {g('SQ_LDS_UNALIGNED_STALL', 2):.3g} "
          f"cycles2 {g('GRBM_GUI_ACTIVE', 2) / 8:.0f}")
PY
