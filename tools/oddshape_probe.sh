#!/bin/bash
# Kernel time (rocprofv3 kernel stats, not host-side events: at 35 - 60 us a call the host's launch rate shows in those) of the
# 90-degree kernels per shape and tuning.  usage: bash tools/oddshape_probe.sh "<shape> <tune>" ...   -> gpurun_out/oddprobe/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/oddprobe; rm -rf $O; mkdir -p $O
i=0
: > $O/summary.txt
for cfg in "$@"; do
    set -- $cfg; sh=$1; tune=${2:-}
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$i -- python3 $R/tools/shapebench.py --shapes $sh --tune "$tune" > $O/r$i.jsonl 2> $O/r$i.err
    python3 - "$sh" "$tune" $O/r$i <<'PY' >> $O/summary.txt
import csv, glob, sys, re
fs = glob.glob(sys.argv[3] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if any(s in r["Name"] for s in ("rot90", "part90", "carve90")):
        m = re.search(r"(k_\w+)", r["Name"])
        print("%-14s %-12s %-22s calls %3s avg %7.1f us min %7.1f" % (sys.argv[1], sys.argv[2] or "-", m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
cat $O/summary.txt
