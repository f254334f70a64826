#!/bin/bash
# rocprofv3 evidence for the bit-sliced chain (csrc/sliced.hip): per-kernel time, then FETCH_SIZE and WRITE_SIZE in SEPARATE counter
# passes (one TCC counter set per pass).  usage: tools/slicedprof.sh <tag> [slicedbench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$tag
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/stats -- python3 $R/tools/slicedbench.py "$@" > $R/gpurun_out/prof_$tag.stats.log 2>&1 || { echo "stats pass failed"; tail -5 $R/gpurun_out/prof_$tag.stats.log; exit 1; }
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/fetch -- python3 $R/tools/slicedbench.py --rounds 1 --reps 1 "$@" > $R/gpurun_out/prof_$tag.fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $R/gpurun_out/prof_$tag.fetch.log; exit 1; }
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/write -- python3 $R/tools/slicedbench.py --rounds 1 --reps 1 "$@" > $R/gpurun_out/prof_$tag.write.log 2>&1 || { echo "write pass failed"; tail -5 $R/gpurun_out/prof_$tag.write.log; exit 1; }
cd $R
python3 tools/profsum.py gpurun_out/prof_$tag > gpurun_out/prof_$tag.summary.txt 2>&1
cat gpurun_out/prof_$tag.summary.txt
