#!/bin/bash
# round 6, GPU call G: FeatureNet's wide layers on the matrix cores: parity, then timing against the float32 kernels
O=gpurun_out/r06g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_costvol.py -x -q -k "conv2d or feature_net or three_stage or config3 or lateral" > $O/pytest_fpn.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_fpn.log; tail -15 $O/pytest_fpn.log
for rep in 1 2; do
  echo "== MFMA"; python tools/bench_featurenet.py 2>/dev/null | tail -1
  echo "== MFMA, lateral as its own launch"; SVS_FPN_FUSE_LATERAL=0 python tools/bench_featurenet.py 2>/dev/null | tail -1
done | tee $O/featurenet_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r06_fpn_tl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_featurenet.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY' | tee gpurun_out/r06g/fpn_timeline_mfma.txt
import csv, glob, os
f = sorted(glob.glob("gpurun_out/r06_fpn_tl/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
              for r in csv.DictReader(open(f)) if "svs::conv2d" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"])
last = rows[-12:]
t0 = last[0][0]
for s, e, n, gx, gy, wx in last:
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  {n.split('(')[0].split('::',1)[-1][:48]:48s} blocks {int(gx) // int(wx)} x {gy}")
print(f"span {(last[-1][1] - t0) / 1e3:.1f} us, sum {sum(e - s for s, e, *_ in last) / 1e3:.1f} us")
PY
find gpurun_out/r06_fpn_tl -name '*kernel_trace.csv' -delete
