#!/bin/bash
# tools/ab_bench.sh OUT "LABEL|ENV (may be empty)|EXTRA ARGS (may be empty)" ...
# A/B of bench.py --only-main, alternating, 3 rounds; prints the headline value, fused-only rate, k_fuse and tracker-pass durations
out=$1; shift
: > $out
for round in 1 2 3; do
  for spec in "$@"; do
    IFS='|' read -r label envs extra <<< "$spec"
    env $envs python bench.py --gpus 1 --only-main $extra 2>/dev/null | grep "^{" | \
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label', j['value'], 'conv', j['config']['converged_frames'], 'passes', j['config']['mean_tracker_passes'], 'fused_only', j['config']['fused_only_fps'], 'k_fuse_us', j['roofline']['avg_launch_us'], 'trk_pass_us', j['roofline']['tracker']['pass_launch_us_median'])" >> $out
  done
done
cat $out
