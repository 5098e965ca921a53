#!/bin/bash
# round 6, GPU call Z: the weight-gradient GEMM with its B fragments requested one tile ahead (counted lgkmcnt): gradient tests,
# bit-identity with the unpipelined form (deterministic-mode hashes), the launches alone, step and kernel A/B
O=gpurun_out/r06z; mkdir -p $O
A=$PWD/s-volsdf_amd/lib_ab
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_bg.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_new.txt; SVS_LIB_PATH=$A/libsvolsdf_hip_nobpipe.so python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_old.txt
if cmp -s $O/hash_new.txt $O/hash_old.txt && [ -s $O/hash_new.txt ]; then echo "BIT-IDENTICAL ($(wc -l < $O/hash_new.txt) lines)"; else echo "DIFFERENT"; diff $O/hash_new.txt $O/hash_old.txt | head; fi | tee $O/identity.txt
for rep in 1 2; do
  echo "== pipelined (product)"; python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
  echo "== unpipelined"; SVS_LIB_PATH=$A/libsvolsdf_hip_nobpipe.so python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
  for d in 4 8; do echo "== DIAG=$d"; SVS_LIB_PATH=$A/libsvolsdf_hip_wgdiag$d.so python tools/dev/time_wgrad.py 1024 2>/dev/null | grep wgrad; done
done | tee $O/wgrad_alone.txt
bash tools/dev/ab_env.sh "SVS_LIB_PATH=$A/libsvolsdf_hip_nobpipe.so" "SVS_NOP=1" 3 --steps 100 | tee $O/ab_step.txt
for v in "SVS_LIB_PATH=$A/libsvolsdf_hip_nobpipe.so" "SVS_NOP=1"; do env $v python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v'[-12:], d['ms_per_step'], [(r['kernel'][:14], r['what'][-22:], round(r['kernel_ms'],4), round(r['frac'],3)) for r in d['roofline']['kernels']])"; done | tee $O/ab_kernels.txt
