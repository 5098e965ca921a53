#!/bin/bash
# round 6, GPU call U: the driver's bench command after the sdf_full row got its second roofline
O=gpurun_out/r06u; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep real $O/bench_driver_cmd.err; wc -c $O/bench_driver_cmd.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06u/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('self_test_failed'))
for r in d['roofline']['kernels']: print(r)
print(d.get('sdf_full_note'))
PY
