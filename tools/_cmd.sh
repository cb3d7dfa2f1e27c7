GSDF_PERSIST=1 python -m pytest tests -m gpu -x -q -k "track or smoke or stream or golden or host" 2>&1 | tail -8 > gpurun_out/r3_t5.log
GSDF_PERSIST=1 python bench.py --steps 20 --warmup 5 --only-main > gpurun_out/r3_bench_p1.json 2> gpurun_out/r3_bench_p1.err
GSDF_PERSIST=0 python bench.py --steps 20 --warmup 5 --only-main > gpurun_out/r3_bench_p0.json 2> gpurun_out/r3_bench_p0.err
cat gpurun_out/r3_t5.log; tail -2 gpurun_out/r3_bench_p1.err
