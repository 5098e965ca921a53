#!/bin/bash
# round 6, GPU call W: a third batch of 12 paired Chamfer seeds (24-35) at 3000 steps: hip, hip_f32, torch_f32 side by side
O=gpurun_out/r06w; mkdir -p $O
timeout 1700 python3 tools/chamfer_parity.py --steps 3000 --seeds 24,25,26,27,28,29,30,31,32,33,34,35 --paths hip,hip_f32,torch_f32 --prior --parallel --out $O/chamfer_paired_24-35.json > $O/chamfer_paired_24-35.log 2>&1
echo "rc $?"; tail -3 $O/chamfer_paired_24-35.log | cut -c1-300
