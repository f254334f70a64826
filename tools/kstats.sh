#!/bin/bash
# per-kernel time of any tool script under rocprofv3.  usage: tools/kstats.sh <tag> <script.py> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ks_$tag; mkdir -p $O
script=$R/$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $script "$@" > $O/stats.log 2>&1 || { echo "stats pass failed"; tail -3 $O/stats.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, re
fs = glob.glob("gpurun_out/ks_$tag/stats/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(fs[0])))[:14]:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Name"])
    print((m.group(1) if m else r["Name"][:40]).ljust(34), "calls", r["Calls"].rjust(4), "avg_us=%.1f" % (float(r["AverageNs"]) / 1e3))
PY
