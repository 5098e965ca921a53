python -X faulthandler -m pytest tests/test_gpu_graph.py tests/test_gpu_train.py tests/test_gpu_volopt.py tests/test_gpu_bg.py -x -q > gpurun_out/r2_tests_b.log 2>&1
for g in off on linear; do
python bench.py --steps 100 --warmup 10 --rays 256 --no-cpu-baseline --no-kernel-timing --graph $g > gpurun_out/r2_bench_b256_$g.log 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --graph $g > gpurun_out/r2_bench_b1024_$g.log 2>&1
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --groups none > gpurun_out/r2_bench_b1024_onegroup.log 2>&1
tail -5 gpurun_out/r2_tests_b.log; for f in gpurun_out/r2_bench_b*_*.log; do echo $f; tail -1 $f | cut -c1-200; done
