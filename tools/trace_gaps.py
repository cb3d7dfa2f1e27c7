"""gaps between consecutive kernels of a rocprofv3 kernel trace: histogram, and the time lost in gaps > 3 us by the kind of launch they follow"""
import collections, csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
prev = None
hist = collections.Counter(); lost = collections.defaultdict(float); cnt = collections.Counter()
busy = 0.0
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:20]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None:
        gap = (s - prev[1]) / 1e3
        if gap < 2000:
            b = "<2" if gap < 2 else "2-4" if gap < 4 else "4-8" if gap < 8 else "8-16" if gap < 16 else "16-50" if gap < 50 else ">50"
            hist[b] += 1
            if gap > 3 and name.startswith("k_track") and prev[0].startswith("k_"):
                lost[prev[0] + " -> " + name] += gap; cnt[prev[0] + " -> " + name] += 1
    busy += (e - s) / 1e3
    prev = (name, e)
print("gap histogram (us):", dict(hist))
for k, v in sorted(lost.items(), key=lambda kv: -kv[1])[:8]:
    print("%-50s n %5d  lost %9.1f us  mean %6.2f" % (k, cnt[k], v, v / cnt[k]))
