python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_t9.log
python bench.py --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r3_bench6.json 2> gpurun_out/r3_bench6.err
python tools/raycast_bench.py > gpurun_out/r3_raycast11.json 2>/dev/null
cat gpurun_out/r3_t9.log; tail -2 gpurun_out/r3_bench6.err
