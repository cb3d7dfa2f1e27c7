# kernel trace of the frame loop with / without the pass-0 role (k_fuse<.., P0>): per-kernel medians
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  rm -rf /tmp/ktp$m
  GSDF_P0_RIDERS=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp$m -o bench -- python $root/bench.py --gpus 1 --steps 20 --warmup 5 --only-main > $root/gpurun_out/p0_trace_$m.json 2> /dev/null
  echo "== GSDF_P0_RIDERS=$m"; python $root/tools/trace_summary.py /tmp/ktp$m | head -12
done
