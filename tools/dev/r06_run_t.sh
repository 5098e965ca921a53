#!/bin/bash
# round 6, GPU call T: the weight-gradient GEMM with its ring slots refilled in two parts (SVS_WGRAD_SCHED=2): gradient tests,
# bit-identity with the kernel in the tree (deterministic-mode hashes), step and kernel A/B
O=gpurun_out/r06t; mkdir -p $O
SVS_WGRAD_SCHED=2 timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_bg.py -x -q > $O/pytest_sched2.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sched2.log
SVS_WGRAD_SCHED=1 python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_1.txt; SVS_WGRAD_SCHED=2 python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_2.txt
if cmp -s $O/hash_1.txt $O/hash_2.txt && [ -s $O/hash_1.txt ]; then echo "BIT-IDENTICAL ($(wc -l < $O/hash_1.txt) lines)"; else echo "DIFFERENT"; diff $O/hash_1.txt $O/hash_2.txt | head -20; fi | tee $O/identity.txt
bash tools/dev/ab_env.sh "SVS_WGRAD_SCHED=1" "SVS_WGRAD_SCHED=2" 3 --steps 100 | tee $O/ab_step.txt
for v in 1 2; do SVS_WGRAD_SCHED=$v python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sched $v', d['ms_per_step'], [(r['kernel'][:14], r['what'][-22:], round(r['kernel_ms'],4), round(r['frac'],3)) for r in d['roofline']['kernels']])"; done | tee $O/ab_kernels.txt
