"""Condense a tools/m4prof.sh (or any rocprofv3 stats + pmc) output directory: per kernel calls / avg us, FETCH_SIZE x2 and WRITE_SIZE per launch (bytes)."""
import csv, glob, os, re, statistics, sys
d = sys.argv[1]
def find(sub, pat):
    fs = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return fs[0] if fs else None
def short(n):
    m = re.search(r"\b(k_\w+)", n)
    return m.group(1) if m else n[:40]
stats = find("stats", "*kernel_stats.csv")
if stats:
    print("kernel,calls,avg_us,total_us,pct")
    for r in csv.DictReader(open(stats)):
        print(f'{short(r["Name"])},{r["Calls"]},{float(r["AverageNs"])/1e3:.1f},{float(r["TotalDurationNs"])/1e3:.1f},{r["Percentage"]}')
for sub, ctr, mul in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    print(f"{ctr} per launch (bytes; FETCH x2 gfx950 correction applied):")
    for k, v in sorted(acc.items()):
        print(f"  {k}: n={len(v)} median={statistics.median(v)*mul:.4g} min={min(v)*mul:.4g} max={max(v)*mul:.4g}")
