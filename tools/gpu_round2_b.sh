python tools/bench_kernels.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py > $GRAFT_REPO_ROOT/gpurun_out/r02d.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r02d/pmc/*/*counter_collection.csv")[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "wgrad" in r["Kernel_Name"]:
        d[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, c in d.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("grid", g, "n", len(c["GRBM_GUI_ACTIVE"]), "cycles", int(cyc), "mfma_util %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)),
          "lds conflict/active %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(1, m["SQ_LDS_IDX_ACTIVE"])), "lds_active/cycle %.3f" % (m["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)),
          "parked %.3f issue_stall %.3f issuing %.3f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
