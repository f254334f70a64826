#!/bin/bash
# Kernel-trace of the labelling kernels, split by the number of colours of the call: tools/ccltrace.sh <tag> [big]
tag=$1; wl=${2:-taj}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ccltrace_$tag
mkdir -p $O
if [ $wl = big ]; then prog="python3 $R/tools/opbench.py --ops N2 --reps 2"; else prog="python3 $R/tools/cclbench.py"; fi
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- $prog > $O/prog.log 2>&1 || { echo "trace failed"; tail -5 $O/prog.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, re, collections, statistics
f = glob.glob("gpurun_out/ccltrace_$tag/t/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_ccl" not in n: continue
    m = re.search(r"(k_ccl_\w+(<[^>]*>)?)", n)
    acc[(m.group(1), r.get("Grid_Size_Y") or r.get("Grid_Size_y"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    print("%-34s colours %s calls %3d  median %7.1f us  min %7.1f  max %7.1f" % (k[0], k[1], len(v), statistics.median(v), min(v), max(v)))
PY
grep "^{" $O/prog.log
