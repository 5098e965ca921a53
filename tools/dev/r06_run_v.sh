#!/bin/bash
# round 6, GPU call V: conv1 of the U-Net with its weight fragments held in registers over TPW tiles per wave (SVS_S2C8_TPW)
O=gpurun_out/r06v; mkdir -p $O
for t in 2 4 8; do SVS_S2C8_TPW=$t timeout 600 python -m pytest tests/test_gpu_costvol.py -x -q -k "stride2_from_8 or costreg or rows_from_split" > $O/pytest_tpw$t.log 2>&1; echo "tpw $t pytest rc $?"; tail -1 $O/pytest_tpw$t.log; done
for rep in 1 2; do for t in 1 2 4 8; do for st in 1 2 3; do echo "tpw $t stage $st: $(SVS_S2C8_TPW=$t python tools/bench_conv.py $st 2>/dev/null | grep conv1)"; done; done; done | tee $O/ab_conv1.txt
