#!/bin/bash
# Collects the evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>       -> gpurun_out/prof_<tag>/...   (copy what is to be judged into profiles/)
# 1. plain bench lines: the driver's window (--steps 20 --warmup 5) and the default window;
# 2. rocprofv3 --kernel-trace --stats over the driver's command (per-kernel table + min/median/p90 + timeline classes);
# 3. rocprofv3 --pmc passes over the SAME command, one counter group per pass (FETCH_SIZE and WRITE_SIZE cannot share a pass).
set -u
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
DRV="python $root/bench.py --gpus 1 --steps 20 --warmup 5"
$DRV > $out/bench_driver_window.json 2> $out/bench_driver_window.err
GSDF_BENCH_DEBUG=1 python $root/bench.py > $out/bench_default.json 2> $out/bench_default.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- $DRV --cpu-frames 0 > $out/bench_profiled.json 2> $out/rocprof_kernel_trace.err
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $out/bench_kernel_stats.csv 2>/dev/null
python $root/tools/trace_summary.py /tmp/kt > $out/bench_kernel_summary.txt 2>&1
python $root/tools/trace_timeline.py /tmp/kt > $out/bench_kernel_timeline.txt 2>&1
# the same trace over the default 200-step window (a quarter of its frames does not converge: 25 passes, no fusion)
rm -rf /tmp/ktd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktd -o bench -- python $root/bench.py --cpu-frames 0 > $out/bench_default_profiled.json 2> $out/rocprof_kernel_trace_default.err
python $root/tools/trace_summary.py /tmp/ktd > $out/bench_default_kernel_summary.txt 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -o pmc -- $DRV --cpu-frames 0 > /dev/null 2> $out/rocprof_pmc$i.err
done
python $root/tools/pmc_summary.py /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4 /tmp/pmc5 > $out/pmc_counters.txt 2> $out/pmc_summary.err
python $root/tools/pmc_summary.py --json "$DRV --cpu-frames 0" /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 > $out/pmc_latest.json 2>> $out/pmc_summary.err
ls -la $out
