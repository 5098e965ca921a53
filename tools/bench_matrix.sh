#!/bin/bash
# The secondary bench lines quoted in DESIGN.md section 5 (one GPU):  gpurun -- 'bash tools/bench_matrix.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/matrix; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-kernel-timing --no-volopt-loop --no-extras --steps 100"
run() { name=$1; shift; $B "$@" 2>/dev/null | tail -1 > $O/$name.json; python3 - "$name" "$O/$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:14s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} {d['unit']}  (host enqueue {(d.get('host_enqueue_ms_per_step') or 0):.2f} ms)")
PY
}
run default
run onegroup --groups none
run render --mode render
run bmvs --model bmvs
run bmvs2048 --model bmvs --rays 2048
run rays2048 --rays 2048
run rays256 --rays 256
run rays512 --rays 512
run graph --graph on
run graph256 --graph on --rays 256
run bmvs256 --model bmvs --rays 256
run bmvs128 --model bmvs --rays 128
# launch mode: the default is `auto` (launch plans below 656 rays, eager launches above)
run eager256 --graph off --rays 256
run eager_bmvs256 --graph off --model bmvs --rays 256
run eager_bmvs128 --graph off --model bmvs --rays 128
run plan1024 --graph plan
run plan1024_1grp --graph plan --groups none
SVS_MLP_PRECISION=f32 run f32
SVS_MLP_PRECISION=f16x2_half run half
SVS_MLP_PRECISION=f16x2_half run half256 --rays 256
SVS_MLP_PRECISION=f16x2_half run half_bmvs --model bmvs
