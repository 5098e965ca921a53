#!/bin/bash
# evidence for the launch-plan / loop numbers of DESIGN.md section 5 (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/plan_evidence; rm -rf $O; mkdir -p $O
bash $R/tools/bench_matrix.sh > $O/bench_matrix.txt 2>&1
MODES="off auto plan" python3 $R/tools/dev/host_vs_gpu.py "--rays 128" "--rays 256" "--rays 128 --model bmvs" "--rays 256 --model bmvs" "--rays 512" "--rays 1024" "--rays 2048 --model bmvs" > $O/host_vs_gpu.txt 2>&1
python3 $R/tools/dev/step_timeline.py "--rays 256 --graph off" eager > $O/timeline_256_eager.txt 2>&1
python3 $R/tools/dev/step_timeline.py "--rays 256 --graph plan" plan > $O/timeline_256_plan.txt 2>&1
python3 $R/tools/dev/volopt_small.py 1024 512 256 128 > $O/volopt_loops.txt 2>&1
cat $O/bench_matrix.txt $O/host_vs_gpu.txt $O/volopt_loops.txt
