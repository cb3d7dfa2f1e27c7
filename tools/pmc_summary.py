"""Per-kernel means of rocprofv3 --pmc counters (counter_collection.csv of one or more passes).

Launches that did no work are excluded per kernel and counter: bench.py queues k_fuse speculatively behind every batch of
tracker passes and the launch exits at once when optimize() has not ended or did not converge, and k_track_pass launches
behind the end of optimize() exit at once too.  A dispatch counts as EXECUTED when its value is at least 20 % of the median of
the non-zero values of that kernel and counter.
usage: pmc_summary.py [--json "<command>" [--source profiles/<file>]] dir [dir ...]"""
import collections, csv, glob, json, statistics, sys
args = sys.argv[1:]
as_json = None
source = "profiles/<tag>_pmc_counters.txt"
if args and args[0] == "--json":
    as_json = args[1]
    args = args[2:]
if args and args[0] == "--source":       # the committed file the numbers can be recomputed from
    source = args[1]
    args = args[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for root in args:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not (k.startswith("k_fuse") or k.startswith("k_track") or k.startswith("k_normals") or k.startswith("k_raycast")):
        continue
    res[k] = {}
    for c, v in sorted(d.items()):
        nz = [x for x in v if x > 0]
        thr = 0.2 * statistics.median(nz) if nz else 0.0
        ex = [x for x in v if x >= thr and x > 0] or [0.0]
        res[k][c] = (sum(ex) / len(ex), len(ex), len(v))
if as_json is None:
    for k, d in res.items():
        print(k)
        for c, (m, n, nt) in d.items():
            print("   %-28s mean %16.1f  (executed launches: %d of %d)" % (c, m, n, nt))
else:
    # The fusion kernel exists in two table sizes (k_fuse<2048, ..>, k_fuse<2560, ..>) and with / without the roles it can carry
    # (normals of the next frame, closing head of optimize()).  `roofline` is quoted on the fusion work alone -- bench.py's replay
    # launches the plain instantiation -- so the traffic figure is the PLAIN instantiation's: of those, the one with the most
    # executed launches (any instantiation if no plain one ran).
    def plain(k):
        return "true" not in k
    cand = [k for k in res if k.startswith("k_fuse") and not k.startswith("k_fuse_resolve")]
    fuse = sorted(cand, key=lambda k: (not plain(k), -res[k].get("FETCH_SIZE", (0, 0, 0))[1]))
    kf = res[fuse[0]] if fuse else {}
    fetch_kb = kf.get("FETCH_SIZE", (0, 0, 0))[0]
    write_kb = kf.get("WRITE_SIZE", (0, 0, 0))[0]
    out = {"source": source, "command": as_json,
           "unit_note": "FETCH_SIZE / WRITE_SIZE in KB as reported by rocprofv3 (TCC_EA0_RDREQ / WRREQ based); on gfx950 FETCH_SIZE "
                        "reports half the bytes of wide coalesced streams and is uncalibrated for the scattered 16-byte accesses of this "
                        "kernel (MI355X_MICROARCH.md, HBM section): the figure is an estimate, ratios between kernel versions are exact",
           "kernel": fuse[0] if fuse else None,
           "k_fuse": {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
                      "TCC_HIT": kf.get("TCC_HIT_sum", (0,))[0], "TCC_MISS": kf.get("TCC_MISS_sum", (0,))[0],
                      "TCC_ATOMIC": kf.get("TCC_ATOMIC_sum", (0,))[0],
                      "executed_launches": kf.get("FETCH_SIZE", (0, 0, 0))[1]},
           "traffic_bytes_per_fusion": round((fetch_kb + write_kb) * 1024)}
    print(json.dumps(out, indent=1))
