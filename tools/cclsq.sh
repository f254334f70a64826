#!/bin/bash
# Where the connected-component kernels' time goes: kernel-trace statistics + two SQ counter passes of the k_ccl_* family on
# (a) the carved 1024^3 colour grid (opbench N2) and (b) the four part colours of the notebook-1 chain at Taj 512 (cclbench).
# usage: tools/cclsq.sh <tag>      (separate passes: --pmc is never combined with a trace domain)
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cclsq_$tag
mkdir -p $O
run() {  # name, rocprof args..., -- program
    local name=$1; shift
    timeout -k 10 300 rocprofv3 "$@" > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; exit 1; }
}
PMC_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
PMC_B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_LDS"
for wl in ${WORKLOADS:-big taj}; do
    if [ $wl = big ]; then prog="python3 $R/tools/opbench.py --ops N2 --reps 2"; else prog="python3 $R/tools/cclbench.py"; fi
    run ${wl}_trace --kernel-trace --stats --output-format csv -d $O/${wl}_trace -- $prog
    run ${wl}_a --pmc $PMC_A --output-format csv -d $O/${wl}_a -- $prog
    run ${wl}_b --pmc $PMC_B --output-format csv -d $O/${wl}_b -- $prog
    run ${wl}_f --pmc FETCH_SIZE --output-format csv -d $O/${wl}_f -- $prog
    run ${wl}_w --pmc WRITE_SIZE --output-format csv -d $O/${wl}_w -- $prog
done
cd $R
python3 - <<PY > $O/summary.txt
import csv, glob, statistics, re
for wl in "${WORKLOADS:-big taj}".split():
    print("==== workload", wl, "(big = carved 1024^3 colour grid, colour full_building; taj = Taj 512 part colours)")
    fs = glob.glob("gpurun_out/cclsq_$tag/%s_trace/**/*kernel_stats.csv" % wl, recursive=True)
    if fs:
        for r in csv.DictReader(open(fs[0])):
            if "k_ccl" in r["Name"] or "k_fin" in r["Name"]:
                m = re.search(r"\b(k_\w+(<[^>]*>)?)", r["Name"])
                print("trace", m.group(1), "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "total_us %.1f" % (float(r["TotalDurationNs"]) / 1e3))
    for sub in ("a", "b", "f", "w"):
        fs = glob.glob("gpurun_out/cclsq_$tag/%s_%s/**/*counter_collection.csv" % (wl, sub), recursive=True)
        if not fs: print("no csv for", wl, sub); continue
        acc = {}
        for r in csv.DictReader(open(fs[0])):
            if "k_ccl" not in r["Kernel_Name"]: continue
            m = re.search(r"\b(k_\w+(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
            acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(k, c, "median=%.5g" % statistics.median(v), "max=%.5g" % max(v), "n=%d" % len(v))
PY
cat $O/summary.txt
