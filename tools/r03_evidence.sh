#!/bin/bash
# HISTORICAL: this script produced the round-3 profiles on the round-3 tree (it uses the numbered misc knobs and kernels that round 4 removed);
# the current one is tools/r04_evidence.sh.
# Round-3 evidence, produced on the GPU box in one go from the FINAL tree: the bench line + rocprofv3 kernel stats + PMC traffic of the
# same command, every op incl. the N2 kernels, the bit-sliced chain (time, per-kernel stats, FETCH / WRITE of a middle step), the
# global_carve chains, the 256-tile 90-degree kernel against the 128-tile one, the notebook-1 chain (host API, resident, kernel
# breakdown, stages) and labelling + statistics per part colour.
# usage (from the repo root on the GPU box): bash tools/r03_evidence.sh   -> files under gpurun_out/r03/
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 700 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_prof.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bench_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_write.log 2>&1
cd $R
echo "== opbench"; python3 tools/opbench.py > $O/opbench.jsonl 2> $O/opbench.err
python3 tools/opbench.py --size 512 --ops M3,M5,N2 > $O/opbench_512.jsonl 2>> $O/opbench.err
echo "== sliced"; python3 tools/slicedbench.py > $O/slicedbench.jsonl 2> $O/slicedbench.err
python3 tools/gcbench.py > $O/gcbench.jsonl 2> $O/gcbench.err
python3 tools/tybench.py --shapes 1024x1024x1024,512x512x512,512x278x512,512x318x512 --fills 0,4,8 --wide 0,2 --rounds 3 > $O/tybench.jsonl 2> $O/tybench.err
python3 tools/tybench.py --shapes 352x512x352,355x512x355,437x512x437,500x400x500,512x278x512,512x512x512,1024x1024x1024 --variants ";misc5=16;rot90_wide=2" --rounds 3 --reps 40 > $O/tybench_maskblock.jsonl 2>> $O/tybench.err
python3 tools/tybench.py --op part --shapes 512x278x512,512x512x512,1024x1024x1024 --variants ";misc5=16" --rounds 3 --reps 15 > $O/tybench_part.jsonl 2>> $O/tybench.err
python3 tools/shapebench.py > $O/shapebench.jsonl 2> $O/shapebench.err
echo "== notebook 1"; python3 tools/notebook1_bench.py > $O/notebook1.json 2> $O/notebook1.err
python3 tools/nb1stages.py > $O/nb1stages.json 2> $O/nb1stages.err
python3 tools/cclbench.py > $O/cclbench.jsonl 2> $O/cclbench.err
bash tools/nb1prof.sh r03 > $O/nb1prof.txt 2>&1
bash tools/slicedprof.sh r03 --shapes 1024x1024x1024 --intervals 5 > $O/slicedprof.txt 2>&1
echo done
