#!/bin/bash
# rocprofv3 evidence for the generic-angle step: kernel stats + FETCH_SIZE / WRITE_SIZE in separate passes.  usage: tools/m4prof.sh <tag> <m4bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/stats -- python3 $R/tools/m4bench.py --no-check --reps 5 "$@" > $R/gpurun_out/prof_$tag.stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/fetch -- python3 $R/tools/m4bench.py --no-check --reps 3 "$@" > $R/gpurun_out/prof_$tag.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/write -- python3 $R/tools/m4bench.py --no-check --reps 3 "$@" > $R/gpurun_out/prof_$tag.write.log 2>&1
cd $R
python3 tools/profsum.py gpurun_out/prof_$tag > gpurun_out/prof_$tag.summary.txt 2>&1
cat gpurun_out/prof_$tag.summary.txt
