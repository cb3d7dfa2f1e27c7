"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv)."""
import collections, csv, glob, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not (k.startswith("k_fuse") or k.startswith("k_track") or k.startswith("k_normals")):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
