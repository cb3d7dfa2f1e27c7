python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3_t11.log
python bench.py --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r3_bench9.json 2> gpurun_out/r3_bench9.err
GSDF_PERSIST=0 python tools/track_trace.py 20 2>&1 | tail -6
cat gpurun_out/r3_t11.log
