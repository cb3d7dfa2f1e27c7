python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_t6.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err
GSDF_NORMALS_STREAM=0 python bench.py --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r3_bench4_ns0.json 2> gpurun_out/r3_bench4_ns0.err
cat gpurun_out/r3_t6.log; tail -2 gpurun_out/r3_bench4.err
