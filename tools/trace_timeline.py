"""Timeline view of a rocprofv3 --kernel-trace CSV: for every k_track_pass launch its grid size, duration and the gap to the
previous kernel; summary per (kernel, grid) class.  usage: python tools/trace_timeline.py <dir> [first_n]"""
import collections, csv, glob, sys
root = sys.argv[1]
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
cls = collections.defaultdict(list)
prev_end = None
dump = int(sys.argv[2]) if len(sys.argv) > 2 else 0
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t_origin = int(rows[0]["Start_Timestamp"]) if rows else 0
for ri, r in enumerate(rows):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    cls[(name, grid)].append(((e - s) / 1e3, gap))
    if skip <= ri < skip + dump:
        print("%6d %-22s grid %7d start %10.2f us dur %7.2f gap %6.2f" % (ri, name, grid, (s - t_origin) / 1e3, (e - s) / 1e3, gap))
    prev_end = e
print("%-26s %9s %6s %8s %8s %8s %9s" % ("kernel", "grid", "n", "med_us", "p90_us", "max_us", "med_gap"))
for (name, grid), v in sorted(cls.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    d = sorted(x for x, _ in v); g = sorted(x for _, x in v)
    print("%-26s %9d %6d %8.2f %8.2f %8.2f %9.2f" % (name, grid, len(d), d[len(d) // 2], d[int(len(d) * 0.9)], d[-1], g[len(g) // 2]))
