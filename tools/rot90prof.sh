#!/bin/bash
# SQ / traffic counters of the 90-degree kernels on one shape.  usage: tools/rot90prof.sh <tag> <shape> [tune]
tag=$1; shape=$2; tune=${3:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r90_$tag
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r90_$tag/s -- python3 $R/tools/shapebench.py --shapes $shape --tune "$tune" > $R/gpurun_out/r90_$tag/bench.jsonl 2>$R/gpurun_out/r90_$tag/s.err
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/r90_$tag/a -- python3 $R/tools/shapebench.py --shapes $shape --tune "$tune" > /dev/null 2>$R/gpurun_out/r90_$tag/a.err
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $R/gpurun_out/r90_$tag/b -- python3 $R/tools/shapebench.py --shapes $shape --tune "$tune" > /dev/null 2>$R/gpurun_out/r90_$tag/b.err
# FETCH_SIZE, WRITE_SIZE and the TCP counters in SEPARATE passes: the TCC has four counter slots (FETCH_SIZE takes three, WRITE_SIZE two);
# asked for together, rocprofiler aborts in the first launch ("Request exceeds the capabilities of the hardware to collect") and its
# signal handler then sits in finalisation until the box's silence watchdog kills the call (round 2, gpurun_out/r90_a/c.err).  Every
# pass writes its own log under gpurun_out/ and is bounded by `timeout`, so a stall is visible and short.
for pass in "c FETCH_SIZE" "d WRITE_SIZE" "e TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    set -- $pass; sub=$1; shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/r90_$tag/$sub -- python3 $R/tools/shapebench.py --shapes $shape --tune "$tune" > $R/gpurun_out/r90_$tag/$sub.out 2>$R/gpurun_out/r90_$tag/$sub.err || { echo "pass $sub failed: see gpurun_out/r90_$tag/$sub.err"; exit 1; }
done
cd $R
python3 - <<PY
import csv, glob, statistics, re
for sub in ("a", "b", "c", "d", "e"):
    fs = glob.glob("gpurun_out/r90_$tag/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print("no csv for", sub); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"\b(k_\w+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        if not any(s in k for s in ("rot90", "rotate_bits", "part90")): continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "median=%.4g" % statistics.median(v), "n=%d" % len(v))
fs = glob.glob("gpurun_out/r90_$tag/s/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if any(s in r["Name"] for s in ("rot90", "rotate_bits", "part90", "carve90")): print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
