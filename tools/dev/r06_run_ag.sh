#!/bin/bash
# round 6, GPU call AG: the weight-gradient GEMM's atomic flush with the column tiles walked from a workgroup-dependent start
O=gpurun_out/r06ag; mkdir -p $O
A=$PWD/s-volsdf_amd/lib_ab
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_bg.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_new.txt; SVS_LIB_PATH=$A/libsvolsdf_hip_inorder.so python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_old.txt
if cmp -s $O/hash_new.txt $O/hash_old.txt && [ -s $O/hash_new.txt ]; then echo "BIT-IDENTICAL ($(wc -l < $O/hash_new.txt) lines)"; else echo "DIFFERENT"; diff $O/hash_new.txt $O/hash_old.txt | head -4; fi | tee $O/identity.txt
for rep in 1 2 3; do
  echo "== staggered"; python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
  echo "== in order"; SVS_LIB_PATH=$A/libsvolsdf_hip_inorder.so python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
done | tee $O/wgrad_alone.txt
bash tools/dev/ab_env.sh "SVS_LIB_PATH=$A/libsvolsdf_hip_inorder.so" "SVS_NOP=1" 3 --steps 100 | tee $O/ab_step.txt
