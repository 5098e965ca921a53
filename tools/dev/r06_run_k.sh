#!/bin/bash
# round 6, GPU call K: the gpu suite twice (flakiness check), smoke, FeatureNet timeline
O=gpurun_out/r06k; mkdir -p $O
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_$i.log 2>&1; echo "run $i pytest rc $?"; tail -2 $O/pytest_gpu_$i.log | head -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
