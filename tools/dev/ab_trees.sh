#!/bin/bash
# A/B of two source trees on one box: bench.py (quick) of the repository against the same of a second checkout (ab_old/, a git
# worktree with its own built library), alternating.   tools/dev/ab_trees.sh [reps] [bench.py flags...]
reps=${1:-3}; shift
Q="--no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-extras --no-volopt-loop --no-other-scaling --no-kernel-timing"
for i in $(seq $reps); do
  for t in ab_old .; do
    ms=$(python $t/bench.py $Q "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$t  $ms ms/step"
  done
done
