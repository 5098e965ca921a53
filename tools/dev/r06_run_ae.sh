#!/bin/bash
# round 6, GPU call AE: the commit's kernel once more -- the whole gpu suite, bit-identity with the kernel of commit beda521
# (deterministic-mode hashes, also in the one-piece and float32 modes), smoke, the driver's bench command, 200 steps
O=gpurun_out/r06ae; mkdir -p $O
A=$PWD/s-volsdf_amd/lib_ab
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_gpu.log
for prec in "" f16x2_half; do
  SVS_MLP_PRECISION=$prec python tools/dev/det_hash.py 3 2>&1 | grep "^dtu\|^bmvs" > $O/hash_new_$prec.txt; SVS_MLP_PRECISION=$prec SVS_LIB_PATH=$A/libsvolsdf_hip_head.so python tools/dev/det_hash.py 3 2>&1 | grep "^dtu\|^bmvs" > $O/hash_old_$prec.txt
  if cmp -s $O/hash_new_$prec.txt $O/hash_old_$prec.txt && [ -s $O/hash_new_$prec.txt ]; then echo "precision '$prec': BIT-IDENTICAL ($(wc -l < $O/hash_new_$prec.txt) lines)"; else echo "precision '$prec': DIFFERENT"; diff $O/hash_new_$prec.txt $O/hash_old_$prec.txt | head -4; fi
done | tee $O/identity.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep real $O/bench_driver_cmd.err; wc -c $O/bench_driver_cmd.json
for i in 1 2 3; do python3 bench.py --steps 200 --no-cpu-baseline --no-gpu-torch --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200 steps:', d['ms_per_step'], d['value'])"; done | tee $O/bench_200x3.txt
python - <<PY
import json
d=json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('self_test_failed'))
for r in d['roofline']['kernels']: print(r['kernel'][:20], r['what'][-24:], r['kernel_ms'], r['frac'], r.get('mfma_frac'), r.get('hbm_frac'))
PY
