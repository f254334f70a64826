#!/bin/bash
# SQ counters of the kernels of one opbench op.  usage: tools/sqop.sh <tag> <op> <kernel-substring>
tag=$1; op=$2; ksub=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sqop_$tag
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/sqop_$tag/a -- python3 $R/tools/opbench.py --ops $op --reps 2 > /dev/null 2>$R/gpurun_out/sqop_$tag/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $R/gpurun_out/sqop_$tag/b -- python3 $R/tools/opbench.py --ops $op --reps 2 > /dev/null 2>$R/gpurun_out/sqop_$tag/b.err
cd $R
python3 - <<PY
import csv, glob, statistics, re
for sub in ("a", "b"):
    fs = glob.glob("gpurun_out/sqop_$tag/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print("no csv for", sub); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if "$ksub" not in r["Kernel_Name"]: continue
        m = re.search(r"\b(k_\w+(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "median=%.4g" % statistics.median(v), "n=%d" % len(v))
PY
