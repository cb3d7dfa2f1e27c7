python bench.py --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r3_bench8.json 2> gpurun_out/r3_bench8.err
python tools/raycast_bench.py > gpurun_out/r3_raycast13.json 2>/dev/null
GSDF_PERSIST=0 python tools/track_trace.py 20 2>&1 | tail -6
