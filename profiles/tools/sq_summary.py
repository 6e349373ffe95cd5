"""Per-kernel means of the SQ counters collected by sq_counters.sh, as fractions of SQ_WAVE_CYCLES."""
import collections, csv, sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        agg[(path, k)][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(path, k)].add(r["Dispatch_Id"])
for (path, k), v in sorted(agg.items(), key=lambda t: (t[0][1], t[0][0])):
    if not k.startswith("k_"):
        continue
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    n = len(launches[(path, k)])
    parts = ", ".join(f"{c[3:]} {val / wc:.3f}" if wc else f"{c[3:]} {val / n:.0f}" for c, val in sorted(v.items()) if c != "SQ_WAVE_CYCLES")
    print(f"{k:44s} n={n:3d} wave_cycles/launch {wc / n:12.0f} | {parts}")
