set -x
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_volopt.py tests/test_gpu_train.py tests/test_gpu_bg.py -q 2>&1 | grep -E "passed|failed" 
for M in dtu bmvs; do
export DP_MODEL=$M
timeout 300 python tools/dev/dp_two_ranks.py single /tmp/dp_ref_$M.pt 2>&1 | tail -1
SVS_DIST_SHARE_GPU=1 SVS_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dev/dp_two_ranks.py dp /tmp/dp_ref_$M.pt 2>&1 | grep -v Warning | tail -3
done
unset DP_MODEL
timeout 900 python tools/dev/volopt_small.py 256 128 1024 2>&1 | tail -8
