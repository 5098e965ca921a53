#!/bin/bash
# round 6, GPU call AB: after the narrow job's new share and the GEMM rows' second roofline: gradient tests, the driver's bench command
O=gpurun_out/r06ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_bg.py tests/test_gpu_volopt.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep real $O/bench_driver_cmd.err; wc -c $O/bench_driver_cmd.json
python - <<PY
import json
d=json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('self_test_failed'))
print({k: d['roofline'][k] for k in ('frac','mfma_frac','kernel_ms') if k in d['roofline']})
for r in d['roofline']['kernels']: print(r)
PY
python3 bench.py --steps 200 --no-cpu-baseline --no-gpu-torch --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200 steps:', d['ms_per_step'], d['value'])"
