#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
# then, back in the container:  python tools/summarize_profiles.py gpurun_out/r02 r02
# Kernel trace and counter passes are separate runs (a --pmc pass never carries a trace domain besides the kernel
# dispatch records rocprofv3 adds by itself); the program follows "--" directly.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-volopt-loop --no-extras"
rocprofv3 --kernel-trace --stats -d $O/train --output-format csv -- $B --steps 20 --warmup 10 > $O/train.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/render --output-format csv -- $B --mode render --steps 20 --warmup 10 > $O/render.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/train_onegroup --output-format csv -- $B --steps 20 --warmup 10 --groups none > $O/onegroup.log 2>&1
# counter passes: few steps, no settle phase / extra passes (every dispatch is serialised under --pmc)
C="$B --steps 3 --warmup 2 --settle 0 --no-kernel-timing --no-host-timing"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $C > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $C > $O/w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  -d $O/pmc_mfma --output-format csv -- $C --groups none > $O/m.log 2>&1
# the opt-in mixed-precision mode (one-piece gradient blocks): kernel stats and traffic of the same step
export SVS_MLP_PRECISION=f16x2_half
rocprofv3 --kernel-trace --stats -d $O/train_half --output-format csv -- $B --steps 20 --warmup 10 > $O/train_half.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_half --output-format csv -- $C > $O/fh.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_half --output-format csv -- $C > $O/wh.log 2>&1
unset SVS_MLP_PRECISION
# the 256-ray step (config 4's per-GPU share)
rocprofv3 --kernel-trace --stats -d $O/train_256 --output-format csv -- $B --steps 20 --warmup 10 --rays 256 > $O/train_256.log 2>&1
# secondary paths: the cost-volume build at config 3 and a whole-image eval render (fast = -1, 500-ray chunks)
rocprofv3 --kernel-trace --stats -d $O/costvol --output-format csv -- python3 $R/tools/bench_costvol.py > $O/costvol.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/evalrender --output-format csv -- python3 $R/tools/bench_render_eval.py > $O/evalrender.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/featurenet --output-format csv -- python3 $R/tools/bench_featurenet.py > $O/featurenet.log 2>&1
cd $R
# keep what travels back small: the stats files and the counter tables only
find $O -name '*kernel_trace.csv' -delete
find $O -name '*agent_info.csv' -delete
du -sh $O; tail -1 $O/train.log | cut -c1-200
