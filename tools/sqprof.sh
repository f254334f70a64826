#!/bin/bash
# SQ counter pass (where do the waves' cycles go) for an m4bench command line.  usage: tools/sqprof.sh <tag> <m4bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/sq_$tag/a -- python3 $R/tools/m4bench.py --no-check --reps 3 "$@" > $R/gpurun_out/sq_$tag.a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/sq_$tag/b -- python3 $R/tools/m4bench.py --no-check --reps 3 "$@" > $R/gpurun_out/sq_$tag.b.log 2>&1
cd $R
python3 - <<PY
import csv, glob, statistics, re
for sub in ("a", "b"):
    fs = glob.glob("gpurun_out/sq_$tag/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print("no csv for", sub); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"\b(k_\w+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        if "rotate_bits" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "median=%.4g" % statistics.median(v), "n=%d" % len(v))
PY
