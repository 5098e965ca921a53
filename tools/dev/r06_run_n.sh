#!/bin/bash
# round 6, GPU call N: ms per step against the length of the timed region (the driver times 20 steps)
O=gpurun_out/r06n; mkdir -p $O
Q="--no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-extras --no-volopt-loop --no-other-scaling --no-kernel-timing --no-host-timing"
for rep in 1 2; do for n in 10 20 40 100 200; do
  ms=$(python bench.py $Q --steps $n --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "steps $n  $ms"
done; done | tee $O/steps_sweep.txt
