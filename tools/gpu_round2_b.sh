python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r2_tests_e.log 2>&1
tail -3 gpurun_out/r2_tests_e.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2_bench_e.log 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays 256 --no-kernel-timing > gpurun_out/r2_bench_e256.log 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --model bmvs --no-kernel-timing > gpurun_out/r2_bench_e_bg.log 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --mode render --no-kernel-timing > gpurun_out/r2_bench_e_render.log 2>&1
for f in gpurun_out/r2_bench_e*.log; do echo $f; tail -1 $f | cut -c1-220; done
