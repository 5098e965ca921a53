#!/bin/bash
# round 6, GPU call M: tile order of pass B / the weight-gradient GEMM (Infinity Cache reuse between consecutive sweeps)
O=gpurun_out/r06m; mkdir -p $O
Q="--no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-extras --no-volopt-loop --no-other-scaling --no-kernel-timing --steps 150"
for rep in 1 2 3; do
  for v in "SVS_X=0" "SVS_BWD_B_REVERSE=1" "SVS_WGRAD_REVERSE=1"; do
    ms=$(env $v python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$v  $ms"
  done
done | tee $O/ab_reverse.txt
for v in "SVS_X=0" "SVS_BWD_B_REVERSE=1"; do env $v python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], [(r['kernel'][:14], round(r['kernel_ms'],4)) for r in d['roofline']['kernels']])"; done | tee -a $O/ab_reverse.txt
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q 2>&1 | tail -2
SVS_BWD_B_REVERSE=1 timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q 2>&1 | tail -2
