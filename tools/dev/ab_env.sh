#!/bin/bash
# tools/dev/ab_env.sh "<ENV=a ...>" "<ENV=b ...>" [reps] [bench flags]: quick bench alternating two environments on one box
A=$1; B=$2; reps=${3:-2}; shift 3
Q="--no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-extras --no-volopt-loop --no-other-scaling --no-kernel-timing"
for i in $(seq $reps); do
  for v in "$A" "$B"; do
    ms=$(env $v python bench.py $Q "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))")
    echo "$v  $ms"
  done
done
