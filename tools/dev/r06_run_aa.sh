#!/bin/bash
# round 6, GPU call AA: the share of workgroups the narrow job (the 16 extra input rows of the radiance network's first layer) gets
O=gpurun_out/r06aa; mkdir -p $O
for rep in 1 2; do for w in 4 5 6 7 8 10; do echo "narrow_work $w: $(SVS_WGRAD_NARROW_WORK=$w python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep '5 jobs' | tr '\n' ' ')"; done; done | tee $O/narrow_alone.txt
bash tools/dev/ab_env.sh "SVS_WGRAD_NARROW_WORK=4" "SVS_WGRAD_NARROW_WORK=7" 3 --steps 100 | tee $O/ab_step.txt
for w in 4 6 7 8; do SVS_WGRAD_NARROW_WORK=$w python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('w=$w', d['ms_per_step'], [(r['kernel'][:14], r['what'][-22:], round(r['kernel_ms'],4), round(r['frac'],3)) for r in d['roofline']['kernels']])"; done | tee $O/ab_kernels.txt
