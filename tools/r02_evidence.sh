#!/bin/bash
# Round-2 evidence, produced on the GPU box in one go: bench line + rocprofv3 kernel stats + PMC traffic of the same command,
# every op of SURVEY 8(d), the real shapes, the generic-angle step (time, FETCH/WRITE, SQ counters), the camera objective latency,
# the workgroups-per-CU sweep of the 90-degree kernels, the cold-chain prefetch figures and the notebook-1 chain.
# usage (from the repo root on the GPU box): bash tools/r02_evidence.sh   -> files under gpurun_out/r02/
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bench_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 tools/opbench.py --ops M1,M2,M3,M4,M5,M6,M7,M8,A2,A6,A9 > $O/opbench.jsonl 2> $O/opbench.err
python3 tools/shapebench.py > $O/shapebench.jsonl 2> $O/shapebench.err
python3 tools/m4bench.py --angles 45,30,5,60 --variants 128,256,256:0:0:0:0:0:1 --reps 10 > $O/m4bench_1024.jsonl 2>&1
python3 tools/m4bench.py --size 512 --angles 45 --variants 64,128,256 --reps 10 > $O/m4bench_512.jsonl 2>&1
python3 tools/m4bench.py --size 512x278x512 --angles 45 --variants 64,128,256 --reps 10 >> $O/m4bench_512.jsonl 2>&1
python3 tools/objbench.py > $O/objective_latency.jsonl 2> $O/objbench.err
python3 tools/tybench.py --fills 0,4,6,8,12 > $O/tybench.jsonl 2> $O/tybench.err
python3 tools/chainbench.py > $O/chainbench.jsonl 2> $O/chainbench.err
python3 tools/notebook1_bench.py > $O/notebook1.json 2> $O/notebook1.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/opbench_stats -- python3 $R/tools/opbench.py --ops M3,M4,M5,M6,M7,M8,A9 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/opbench_fetch -- python3 $R/tools/opbench.py --ops M3,M4,M5,M6,M7,M8 --reps 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/opbench_write -- python3 $R/tools/opbench.py --ops M3,M4,M5,M6,M7,M8 --reps 2 > /dev/null 2>&1
cd $R
bash tools/sqprof.sh r02 --angles 45 --variants 256 > $O/m4_sq_counters.txt 2>&1
echo done
