#!/bin/bash
# Collects the evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>       -> gpurun_out/prof_<tag>/...   (copy what is to be judged into profiles/)
# 1. plain bench lines: the driver's window (--steps 20 --warmup 5) and the default window;
# 2. rocprofv3 --kernel-trace --stats over the driver's command (per-kernel table + min/median/p90 + timeline classes);
# 3. rocprofv3 --pmc passes over the SAME command, one counter group per pass (FETCH_SIZE and WRITE_SIZE cannot share a pass).
#   tools/collect_profiles.sh <tag> c3    -> the same for BASELINE configs[2] (1280x960, 5 mm voxels, capacity 2^25: the map is
#                                            1 GiB of records, outside the Infinity Cache): bench line, kernel trace, PMC passes
set -u
tag=${1:-r02}
cfg=${2:-c2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
DRV="python $root/bench.py --gpus 1 --steps 20 --warmup 5"
PMCJSON=pmc_latest.json
if [ "$cfg" = "c3" ]; then
  tag=${tag}_c3
  DRV="$DRV --width 1280 --height 960 --voxel-size 0.005 --hash-capacity-log2 25"
  PMCJSON=pmc_latest_c3.json
  SKIP_EXTRAS=1
fi
out=$root/gpurun_out/prof_$tag
mkdir -p $out
if [ -z "${SKIP_BENCH:-}" ]; then
$DRV > $out/bench_driver_window.json 2> $out/bench_driver_window.err
[ "$cfg" = "c3" ] || GSDF_BENCH_DEBUG=1 python $root/bench.py > $out/bench_default.json 2> $out/bench_default.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- $DRV --cpu-frames 0 > $out/bench_profiled.json 2> $out/rocprof_kernel_trace.err
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $out/bench_kernel_stats.csv 2>/dev/null
python $root/tools/trace_summary.py /tmp/kt > $out/bench_kernel_summary.txt 2>&1
python $root/tools/trace_timeline.py /tmp/kt > $out/bench_kernel_timeline.txt 2>&1
# the same trace over the default 200-step window (a quarter of its frames does not converge: 25 passes, no fusion)
[ "$cfg" = "c3" ] || { rm -rf /tmp/ktd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktd -o bench -- python $root/bench.py --cpu-frames 0 > $out/bench_default_profiled.json 2> $out/rocprof_kernel_trace_default.err
python $root/tools/trace_summary.py /tmp/ktd > $out/bench_default_kernel_summary.txt 2>&1; }
fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  # --only-main: the fused+tracked windows and the roofline replays.  (The sharded flavour creates an RCCL communicator, and
  # rocprofv3's counter collection, which serialises dispatches, does not get along with RCCL's kernels: the pass hung.)
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -o pmc -- $DRV --only-main > /dev/null 2> $out/rocprof_pmc$i.err
done
python $root/tools/pmc_summary.py /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4 /tmp/pmc5 > $out/pmc_counters.txt 2> $out/pmc_summary.err
python $root/tools/pmc_summary.py --json "$DRV --only-main" --source "profiles/${tag}_pmc_counters.txt" /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 > $out/$PMCJSON 2>> $out/pmc_summary.err
if [ -n "${SKIP_EXTRAS:-}" ]; then ls -la $out; exit 0; fi
# 4. the raycaster: timing against the sample-at-a-time kernel (test build), per-workgroup lifetimes, PMC counters of both
python $root/tools/raycast_bench.py > $out/raycast_bench.json 2> $out/raycast_bench.err
GRAFT_REPO_ROOT=$root bash $root/tools/raycast_pmc.sh $tag > /dev/null 2>&1
cp $root/gpurun_out/raycast_pmc_$tag.txt $out/raycast_pmc_counters.txt 2>/dev/null
# 5. the tracker: per-workgroup phase traces of one frame, one launch per pass (default) and the one-launch optimize() (GSDF_PERSIST=1)
GSDF_PERSIST=0 python $root/tools/track_trace.py 20 > $out/track_trace_per_pass.txt 2>&1
GSDF_PERSIST=1 python $root/tools/track_all_trace.py 20 > $out/track_trace_one_launch.txt 2>&1
GSDF_PERSIST=1 python $root/bench.py --steps 20 --warmup 5 --only-main > $out/bench_one_launch_tracker.json 2> /dev/null
# 6. k_fuse and the number of dispatch rounds: frame sizes with 1040 / 1200 / 1536 / 2048 tiles on the 512 workgroup slots
for wh in "640 416" "640 480" "768 512" "1024 512"; do set -- $wh
  python $root/bench.py --steps 20 --warmup 5 --only-main --repeats 2 --width $1 --height $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; t=(($1+15)//16)*(($2+15)//16)
        print('%4d x %3d: %4d tiles = %.2f rounds of 512 workgroups | k_fuse %.1f us = %.1f ns per tile | %d updates -> frac %.3f' % ($1, $2, t, t/512.0, r['avg_launch_us'], 1e3*r['avg_launch_us']/t, d['config']['n_upd_per_frame'], r['frac']))
"
done > $out/fuse_rounds.txt
# 7. what WRITE_SIZE / FETCH_SIZE count for the flush's access pattern (tools/write_calib.hip)
GRAFT_REPO_ROOT=$root bash $root/tools/write_calib.sh > /dev/null 2>&1
cp $root/gpurun_out/write_calib.txt $out/write_calibration.txt 2>/dev/null
ls -la $out
