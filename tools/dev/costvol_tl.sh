cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r06_cv_tl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_costvol.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/dev/costvol_timeline_all.py gpurun_out/r06_cv_tl | tee gpurun_out/r06_costvol_timeline.txt; find gpurun_out/r06_cv_tl -name '*kernel_trace.csv' -delete
