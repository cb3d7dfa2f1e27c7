"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / min / median / p90 / max / sum (us)."""
import collections, csv, glob, sys
root = sys.argv[1] if len(sys.argv) > 1 else "."
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + root)
d = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-50s %6s %8s %8s %8s %8s %10s" % ("kernel", "n", "min", "med", "p90", "max", "sum_us"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-50s %6d %8.1f %8.1f %8.1f %8.1f %10.1f" % (k, len(v), v[0], v[len(v) // 2], v[int(len(v) * 0.9)], v[-1], sum(v)))
