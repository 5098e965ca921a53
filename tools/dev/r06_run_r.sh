#!/bin/bash
# round 6, GPU call R: the one-wave-per-SIMD weight-gradient GEMM (SVS_WGRAD_KERNEL=wide): gradient tests, then A/B
O=gpurun_out/r06r; mkdir -p $O
SVS_WGRAD_KERNEL=wide timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_bg.py -x -q > $O/pytest_wide.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_wide.log
bash tools/dev/ab_env.sh "SVS_WGRAD_KERNEL=multi" "SVS_WGRAD_KERNEL=wide" 3 --steps 100 | tee $O/ab_wide.txt
for v in multi wide; do SVS_WGRAD_KERNEL=$v python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], [(r['kernel'][:14], r['what'][-26:], round(r['kernel_ms'],4), round(r['frac'],3)) for r in d['roofline']['kernels']])"; done | tee -a $O/ab_wide.txt
