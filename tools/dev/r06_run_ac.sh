#!/bin/bash
# round 6, GPU call AC: the weight-gradient GEMM's serial part per item shortened (job parameters in registers instead of scalar
# loads per item; the A fragments of both k-steps requested together, the next item's copies issued behind the requests, one wait):
# gradient tests, bit-identity with the previous commit's kernel, the launches alone, step and kernel A/B
O=gpurun_out/r06ad; mkdir -p $O
A=$PWD/s-volsdf_amd/lib_ab
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_bg.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_new.txt; SVS_LIB_PATH=$A/libsvolsdf_hip_head.so python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_old.txt
if cmp -s $O/hash_new.txt $O/hash_old.txt && [ -s $O/hash_new.txt ]; then echo "BIT-IDENTICAL ($(wc -l < $O/hash_new.txt) lines)"; else echo "DIFFERENT"; diff $O/hash_new.txt $O/hash_old.txt | head; fi | tee $O/identity.txt
for rep in 1 2; do
  echo "== new"; python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
  echo "== factor table staged at the top of the item (the previous step of this work)"; SVS_LIB_PATH=$A/libsvolsdf_hip_stagetop.so python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
  echo "== previous commit"; SVS_LIB_PATH=$A/libsvolsdf_hip_head.so python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
done | tee $O/wgrad_alone.txt
bash tools/dev/ab_env.sh "SVS_LIB_PATH=$A/libsvolsdf_hip_head.so" "SVS_NOP=1" 3 --steps 100 | tee $O/ab_step.txt
for v in "SVS_LIB_PATH=$A/libsvolsdf_hip_head.so" "SVS_NOP=1"; do env $v python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v'[-10:], d['ms_per_step'], [(r['kernel'][:14], r['what'][-22:], round(r['kernel_ms'],4), round(r['frac'],3)) for r in d['roofline']['kernels']])"; done | tee $O/ab_kernels.txt
