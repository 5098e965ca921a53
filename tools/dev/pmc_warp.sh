#!/bin/bash
# counters of the fused warp + variance kernel (dev aid; run on the GPU box).  Every pass under `timeout`: a counter set
# the profiler rejects (FETCH_SIZE with TCC_HIT_sum did) aborts it and leaves it hanging until gpurun's limit.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_warp
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
         "FETCH_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $P -d $O/p$i --output-format csv -- python3 $R/tools/dev/time_warp.py > $O/p$i.log 2>&1
done
cd $R
find $O -name '*agent_info.csv' -delete
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_warp")
for name in ("warp_variance_kernel<32, 2", "warp_variance_kernel<16, 2", "warp_variance_kernel<8, 2"):
    m = {}
    for f in glob.glob(f"{O}/p*/*/*counter_collection.csv"):
        d = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if name not in row["Kernel_Name"]: continue
            d[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, x in d.items(): m[k] = sum(x) / len(x)
    print(name, {k: f"{v:.4g}" for k, v in sorted(m.items())})
PY
