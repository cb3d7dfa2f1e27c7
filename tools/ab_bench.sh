#!/bin/bash
# tools/ab_bench.sh OUT label1:"env / args" ...   -- A/B of bench.py --only-main on the driver's window, alternating, 2 rounds
# each entry: LABEL|ENV (may be empty)|EXTRA ARGS (may be empty)
out=$1; shift
: > $out
for round in 1 2 3; do
  for spec in "$@"; do
    IFS='|' read -r label envs extra <<< "$spec"
    env $envs python bench.py --gpus 1 --steps 20 --warmup 5 --only-main $extra 2>/dev/null | grep "^{" | \
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label', j['value'], 'fused_only', j['config']['fused_only_fps'], 'k_fuse_us', j['roofline']['avg_launch_us'], 'frac', j['roofline']['frac'], 'trk_pass_us', j['roofline']['tracker']['pass_launch_us_median'])" >> $out
  done
done
cat $out
