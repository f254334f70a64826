#!/bin/bash
# Point extraction (M7) under rocprofv3: per-kernel time, FETCH_SIZE / WRITE_SIZE (separate passes) and the SQ counters of the count and
# fill kernels.  usage: tools/m7prof.sh <tag> [opbench args, e.g. --tune points_fill=1]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/m7_$tag; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/opbench.py --ops M7 "$@" > $O/stats.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/opbench.py --ops M7 --reps 2 "$@" > $O/fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/opbench.py --ops M7 --reps 2 "$@" > $O/write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sqa -- python3 $R/tools/opbench.py --ops M7 --reps 2 "$@" > $O/sqa.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $O/sqb -- python3 $R/tools/opbench.py --ops M7 --reps 2 "$@" > $O/sqb.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, statistics, re
fs = glob.glob("gpurun_out/m7_$tag/stats/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(fs[0])))[:8]:
    print(r["Name"][:70].ljust(70), r["Calls"], "avg_us=%.1f" % (float(r["AverageNs"]) / 1e3))
for sub in ("fetch", "write", "sqa", "sqb"):
    fs = glob.glob("gpurun_out/m7_$tag/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print("no csv for", sub); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if "k_points" not in r["Kernel_Name"]: continue
        m = re.search(r"\b(k_\w+(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        x = statistics.median(v)
        note = "  (KiB; x2 gfx950 correction -> %.3f GB)" % (x * 2 * 1024 / 1e9) if c == "FETCH_SIZE" else ("  (KiB -> %.3f GB)" % (x * 1024 / 1e9) if c == "WRITE_SIZE" else "")
        print(k, c, "median=%.5g" % x, "n=%d" % len(v), note)
PY
