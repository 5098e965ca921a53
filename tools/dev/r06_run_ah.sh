#!/bin/bash
# round 6, GPU call AH: with several ray groups the sampler runs once for the whole batch (SVS_PRESAMPLE): same forward bits,
# the training tests, step A/B
O=gpurun_out/r06ah; mkdir -p $O
SVS_PRESAMPLE=1 python tools/dev/presample_check.py 2>&1 | grep "^step" > $O/h1.txt; SVS_PRESAMPLE=0 python tools/dev/presample_check.py 2>&1 | grep "^step" > $O/h0.txt
if cmp -s $O/h0.txt $O/h1.txt && [ -s $O/h1.txt ]; then echo "forward BIT-IDENTICAL: $(cat $O/h1.txt)"; else echo "DIFFERENT"; cat $O/h0.txt $O/h1.txt; SVS_PRESAMPLE=1 python tools/dev/presample_check.py 2>&1 | tail -5; fi | tee $O/identity.txt
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_volopt.py tests/test_gpu_bg.py tests/test_gpu_graph.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
bash tools/dev/ab_env.sh "SVS_PRESAMPLE=0" "SVS_PRESAMPLE=1" 3 --steps 100 | tee $O/ab_step.txt
for v in 0 1; do SVS_PRESAMPLE=$v python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('presample $v', d['ms_per_step'], [(r['kernel'][:14], r['what'][-22:], round(r['kernel_ms'],4), round(r['frac'],3)) for r in d['roofline']['kernels']])"; done | tee $O/ab_kernels.txt
