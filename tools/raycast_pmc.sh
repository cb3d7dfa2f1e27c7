#!/bin/bash
# PMC counters of the raycaster (k_raycast, and k_raycast_v1 of the test build) over tools/raycast_bench.py:
#   tools/raycast_pmc.sh <tag>  -> gpurun_out/raycast_pmc_<tag>.txt
set -u
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
CMD="python $root/tools/raycast_bench.py 26 5"
i=0
dirs=""
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rm -rf /tmp/rpmc$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/rpmc$i -o pmc -- $CMD > /dev/null 2> $root/gpurun_out/raycast_pmc_$tag.err$i
  dirs="$dirs /tmp/rpmc$i"
done
python $root/tools/pmc_summary.py $dirs > $root/gpurun_out/raycast_pmc_$tag.txt 2>&1
cat $root/gpurun_out/raycast_pmc_$tag.txt
