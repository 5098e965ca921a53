#!/bin/bash
# round 6, GPU call F: the data-parallel path with two REAL ranks on the one GPU (collectives over gloo): bucketed all-reduce in
# eager steps and the default (first step eager, then launch plans); FeatureNet's wide layers as one-plane 3-D convolutions
O=gpurun_out/r06f; mkdir -p $O
for M in dtu bmvs; do
  export DP_MODEL=$M
  timeout 300 python tools/dev/dp_two_ranks.py single /tmp/dp_ref_$M.pt 2>&1 | tail -1
  for G in auto 0; do
    echo "== model $M, SVS_TRAIN_GRAPH=$G"
    SVS_TRAIN_GRAPH=$G SVS_DIST_SHARE_GPU=1 SVS_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dev/dp_two_ranks.py dp /tmp/dp_ref_$M.pt 2>&1 | grep -v -i "warning\|warn(" | tail -4
  done
done 2>&1 | tee $O/dp_two_ranks.txt
unset DP_MODEL
python tools/dev/time_fpn_as3d.py 2>&1 | grep -v -i warn | tee $O/fpn_as3d.txt
