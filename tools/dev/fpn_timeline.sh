cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r06_fpn_tl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_featurenet.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/dev/fpn_timeline.py gpurun_out/r06_fpn_tl | tee gpurun_out/r06_fpn_timeline.txt; find gpurun_out/r06_fpn_tl -name '*kernel_trace.csv' -delete
