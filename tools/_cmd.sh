python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 > gpurun_out/r3_t7.log
python tools/run_c4.py --frames 2000 --force-exchange > gpurun_out/r3_c4.json 2> gpurun_out/r3_c4.err
cat gpurun_out/r3_t7.log; tail -2 gpurun_out/r3_c4.json
