#!/bin/bash
# A/B of one environment switch on one box: bench.py (quick: no baselines / extras) alternating the two settings.
#   tools/dev/ab_bench.sh VAR A B [reps] [extra bench.py flags...]   -> ms_per_step per run
var=$1; a=$2; b=$3; reps=${4:-2}; shift 4
Q="--no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-extras --no-volopt-loop --no-other-scaling --no-kernel-timing"
for i in $(seq $reps); do
  for v in $a $b; do
    ms=$(env $var=$v python bench.py $Q "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$var=$v  $ms ms/step"
  done
done
