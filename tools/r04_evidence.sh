#!/bin/bash
# Round-4 evidence, produced on the GPU box in one go from the tree it is run on: GPU tests, the bench line + rocprofv3 kernel stats + PMC
# traffic of the same command, every op (opbench), the bit-sliced chains, global_carve chains, the real shapes, the notebook-1 chain and
# labelling per part colour.   usage (repo root, GPU box): bash tools/r04_evidence.sh [tag]   -> files under gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gputests.log; fi
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 900 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_prof.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bench_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_write.log 2>&1
cd $R
echo "== opbench"; python3 tools/opbench.py > $O/opbench.jsonl 2> $O/opbench.err
echo "== sliced"; python3 tools/slicedbench.py > $O/slicedbench.jsonl 2> $O/slicedbench.err
python3 tools/gcbench.py > $O/gcbench.jsonl 2> $O/gcbench.err
python3 tools/shapebench.py > $O/shapebench.jsonl 2> $O/shapebench.err
python3 tools/tybench.py --op part --shapes 512x278x512,355x512x355,1024x1024x1024 --variants ";" --rounds 3 --reps 15 > $O/tybench_part.jsonl 2> $O/tybench.err
echo "== notebook 1"; python3 tools/notebook1_bench.py > $O/notebook1.json 2> $O/notebook1.err
python3 tools/cclbench.py > $O/cclbench.jsonl 2> $O/cclbench.err
bash tools/nb1prof.sh $TAG > $O/nb1prof.txt 2>&1
echo "== slab PMC"; bash tools/slab_pmc.sh $TAG > $O/slab_pmc.txt 2>&1; tail -4 $O/slab_pmc.txt | cut -c1-200
bash tools/slicedprof.sh $TAG --shapes 1024x1024x1024 --intervals 45,5 > $O/slicedprof.txt 2>&1
echo done
