#!/bin/bash
# kernel trace of the notebook-1 chain (Taj @ 512): where the resident chain's milliseconds go.  usage: tools/nb1prof.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/nb1_$tag
python3 $R/tools/notebook1_bench.py > $R/gpurun_out/nb1_$tag/plain.json 2>$R/gpurun_out/nb1_$tag/plain.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nb1_$tag/s -- python3 $R/tools/notebook1_bench.py > $R/gpurun_out/nb1_$tag/prof.json 2>$R/gpurun_out/nb1_$tag/s.err || exit 1
cd $R
python3 - <<PY
import csv, glob
fs = glob.glob("gpurun_out/nb1_$tag/s/**/*kernel_stats.csv", recursive=True)
tot = 0
rows = list(csv.DictReader(open(fs[0])))
for r in rows: tot += float(r["TotalDurationNs"])
print("total kernel ms over the whole script:", round(tot / 1e6, 2))
for r in rows[:25]:
    print(r["Name"][:80].ljust(80), r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 1))
PY
