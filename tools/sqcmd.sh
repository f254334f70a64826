#!/bin/bash
# SQ counters (two passes) + kernel trace of any tool script.  usage: tools/sqcmd.sh <tag> <kernel-substring> <script.py> [args...]
tag=$1; ksub=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_$tag; mkdir -p $O
script=$R/$1; shift
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $script "$@" > $O/stats.log 2>&1 || { echo "stats pass failed"; tail -3 $O/stats.log; exit 1; }
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -- python3 $script "$@" > $O/a.log 2>&1 || { echo "pass a failed"; exit 1; }
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $script "$@" > $O/b.log 2>&1 || { echo "pass b failed"; exit 1; }
cd $R
python3 - <<PY
import csv, glob, statistics, re
fs = glob.glob("gpurun_out/sq_$tag/stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if "$ksub" in r["Name"]:
        print(re.search(r"(k_\w+(<[^>]*>)?)", r["Name"]).group(1).ljust(30), "calls", r["Calls"], "avg_us=%.1f" % (float(r["AverageNs"]) / 1e3))
for sub in ("a", "b"):
    fs = glob.glob("gpurun_out/sq_$tag/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print("no csv for", sub); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if "$ksub" not in r["Kernel_Name"]: continue
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k.ljust(30), c.ljust(22), "median=%.4g" % statistics.median(v), "n=%d" % len(v))
PY
