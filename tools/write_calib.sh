#!/bin/bash
# tools/write_calib.sh -> gpurun_out/write_calib.txt : WRITE_SIZE / FETCH_SIZE per launch of tools/write_calib.hip against the bytes it moves
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/write_calib.txt
: > $out
for mode in 0 1 2 3 4 5; do
  for ctr in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/wc
    rocprofv3 --pmc $ctr --output-format csv -d /tmp/wc -o pmc -- $root/tools/bin/write_calib $mode > /tmp/wc.log 2>&1
    python3 - "$mode" "$ctr" >> $out <<'PY'
import csv, glob, sys, statistics
mode, ctr = sys.argv[1], sys.argv[2]
vals = []
for f in glob.glob("/tmp/wc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void k_calib") or "k_calib" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
n = 1260000 * 32
if vals:
    m = statistics.median(vals)
    print("mode %s %-10s median %12.1f KB per launch = %6.3f x the %0.2f MB the kernel %s" % (mode, ctr, m, m * 1024 / n, n / 1e6, "writes" if ctr == "WRITE_SIZE" else "reads (modes 4, 5) / writes"))
else:
    print("mode %s %s: no samples" % (mode, ctr))
PY
  done
done
cat $out
