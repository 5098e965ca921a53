python -m pytest tests/test_gpu_graph.py tests/test_gpu_train.py tests/test_gpu_volopt.py -x -q 2>&1 | tail -25 > gpurun_out/r2_tests_b.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r2_bench_b.log 2>&1
python bench.py --steps 100 --warmup 10 --rays 256 --no-cpu-baseline > gpurun_out/r2_bench_b256.log 2>&1
python bench.py --steps 100 --warmup 10 --rays 256 --no-cpu-baseline --no-graph --no-kernel-timing > gpurun_out/r2_bench_b256_eager.log 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph --no-kernel-timing > gpurun_out/r2_bench_b_eager.log 2>&1
python bench.py --steps 100 --warmup 10 --rays 256 --model bmvs --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_bench_b256_bg.log 2>&1
tail -5 gpurun_out/r2_tests_b.log; for f in gpurun_out/r2_bench_b*.log; do echo $f; tail -1 $f | cut -c1-260; done
