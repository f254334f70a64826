#!/bin/bash
# SQ counter passes (where do the waves' cycles go) for the kernels of the bit-sliced chain.  usage: tools/slicedsq.sh <tag> [slicedbench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ssq_$tag
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/ssq_$tag/a -- python3 $R/tools/slicedbench.py --rounds 1 --reps 1 "$@" > $R/gpurun_out/ssq_$tag.a.log 2>&1 || { echo "pass a failed"; tail -3 $R/gpurun_out/ssq_$tag.a.log; exit 1; }
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/ssq_$tag/b -- python3 $R/tools/slicedbench.py --rounds 1 --reps 1 "$@" > $R/gpurun_out/ssq_$tag.b.log 2>&1 || { echo "pass b failed"; tail -3 $R/gpurun_out/ssq_$tag.b.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, statistics, re
for sub in ("a", "b"):
    fs = glob.glob("gpurun_out/ssq_$tag/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print("no csv for", sub); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"\b(k_\w+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        if "k_s32" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "median=%.4g" % statistics.median(v), "n=%d" % len(v))
PY
