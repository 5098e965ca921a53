#!/bin/bash
# round 6, GPU call J: bench.py's N > 1 path with two ranks sharing the one GPU (collectives over gloo): the line must parse
O=gpurun_out/r06j; mkdir -p $O
SVS_DIST_SHARE_GPU=1 SVS_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks.out 2> $O/bench_2ranks.err
echo "rc $?"; tail -3 $O/bench_2ranks.err | cut -c1-300
python - <<PY
import json
t=open("$O/bench_2ranks.out").read().strip().splitlines()[-1]
d=json.loads(t); print(len(t), d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d.get("other_scaling"), d["roofline"]["frac"], list(d))
PY
