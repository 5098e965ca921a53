#!/bin/bash
# round 6, GPU call H: the whole gpu suite (with durations), smoke, the driver's bench command
O=gpurun_out/r06h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log; tail -22 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep real $O/bench_driver_cmd.err; wc -c $O/bench_driver_cmd.json; cut -c1-700 $O/bench_driver_cmd.json
