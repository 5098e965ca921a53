#!/bin/bash
# round 6, GPU call AI: the loss's final reduction with 1024 threads (one load per thread instead of five dependent round trips)
O=gpurun_out/r06ai; mkdir -p $O
A=$PWD/s-volsdf_amd/lib_ab
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_bg.py tests/test_gpu_volopt.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
python tools/dev/det_hash.py 3 2>&1 | grep "^dtu\|^bmvs" > $O/hash_new.txt; SVS_LIB_PATH=$A/libsvolsdf_hip_head.so python tools/dev/det_hash.py 3 2>&1 | grep "^dtu\|^bmvs" > $O/hash_old.txt
if cmp -s $O/hash_new.txt $O/hash_old.txt && [ -s $O/hash_new.txt ]; then echo "BIT-IDENTICAL incl. the loss values ($(wc -l < $O/hash_new.txt) lines)"; else echo "DIFFERENT"; diff $O/hash_new.txt $O/hash_old.txt | head -6; fi | tee $O/identity.txt
bash tools/dev/ab_env.sh "SVS_LIB_PATH=$A/libsvolsdf_hip_head.so" "SVS_NOP=1" 4 --steps 100 | tee $O/ab_step.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06ai/prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-volopt-loop --no-extras --steps 20 --warmup 10 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; grep -h "loss_reduce\|loss_rays\|composite" gpurun_out/r06ai/prof/*/*kernel_stats.csv | cut -c1-160; find gpurun_out/r06ai/prof -name '*kernel_trace.csv' -delete
