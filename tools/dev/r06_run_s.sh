#!/bin/bash
# round 6, GPU call S: the operand split by v_fma_mix{lo,hi}_f16 (12 instructions per 8 values instead of 20): bit-identity
# against the C++ form (deterministic-mode hashes), the MLP parity tests, kernel times, step A/B
O=gpurun_out/r06s; mkdir -p $O
P=$PWD/s-volsdf_amd/lib_ab/libsvolsdf_hip_mix.so
python tools/dev/det_hash.py 4 2>&1 | grep -v Warn > $O/hash_default.txt; SVS_LIB_PATH=$P python tools/dev/det_hash.py 4 2>&1 | grep -v Warn > $O/hash_variant.txt
tail -n +2 $O/hash_default.txt > $O/a.txt; tail -n +2 $O/hash_variant.txt > $O/b.txt
if cmp -s $O/a.txt $O/b.txt && [ -s $O/a.txt ]; then echo "BIT-IDENTICAL ($(wc -l < $O/a.txt) lines)"; else echo "DIFFERENT"; diff $O/a.txt $O/b.txt | head -20; fi | tee $O/identity.txt
tail -3 $O/hash_default.txt
for i in 1 2; do for v in "SVS_NOP=1" "SVS_LIB_PATH=$P"; do env $v python tools/bench_kernels.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v'[:12], {k: round(v,4) for k,v in d.items() if isinstance(v,float) and k.endswith('_ms')})"; done; done | tee $O/ab_isolated.txt
bash tools/dev/ab_env.sh "SVS_LIB_PATH=$P" "SVS_NOP=1" 3 --steps 100 | tee $O/ab_step.txt
for v in "SVS_LIB_PATH=$P" "SVS_NOP=1"; do env $v python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v'[:12], d['ms_per_step'], [(r['kernel'][:14], r['what'][-22:], round(r['kernel_ms'],4), round(r['frac'],3)) for r in d['roofline']['kernels']])"; done | tee $O/ab_kernels.txt
