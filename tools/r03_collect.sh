#!/bin/bash
# Copy what tools/r03_evidence.sh left under gpurun_out/r03/ into profiles/ under the names profiles/README.md lists (run here, after
# the GPU call has merged its outputs back), and re-make profiles/pmc_traffic.json from the PMC passes of that run.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03; P=profiles
newest() { ls -t $1 2>/dev/null | head -1; }
cp $O/bench.json $P/r03_bench.json
cp "$(newest "$O/bench_stats/runc/*_kernel_stats.csv")" $P/r03_bench_kernel_stats.csv
cp "$(newest "$O/bench_fetch/runc/*_counter_collection.csv")" $P/r03_bench_pmc_fetch_size.csv
cp "$(newest "$O/bench_write/runc/*_counter_collection.csv")" $P/r03_bench_pmc_write_size.csv
cp $O/opbench.jsonl $P/r03_opbench_all_ops.jsonl; cp $O/opbench_512.jsonl $P/r03_opbench_512.jsonl
cp $O/slicedbench.jsonl $P/r03_slicedbench.jsonl; cp $O/slicedprof.txt $P/r03_sliced_chain_kernel_stats_and_pmc.txt
cp $O/gcbench.jsonl $P/r03_gcbench.jsonl; cp $O/tybench.jsonl $P/r03_tybench_tile256_vs_tile128.jsonl
[ -s $O/tybench_maskblock.jsonl ] && cp $O/tybench_maskblock.jsonl $P/r03_tybench_maskblock_and_odd_shapes.jsonl
[ -s $O/tybench_part.jsonl ] && cp $O/tybench_part.jsonl $P/r03_tybench_part_carve_jobbits.jsonl
cp $O/shapebench.jsonl $P/r03_shapebench_real_shapes.jsonl
cp $O/notebook1.json $P/r03_notebook1_taj512_host_api.json; cp $O/nb1prof.txt $P/r03_notebook1_taj512_kernel_stats.txt
cp $O/nb1stages.json $P/r03_notebook1_stages.json; cp $O/cclbench.jsonl $P/r03_cclbench.jsonl
python tools/pmc_summary.py $P/r03_bench_pmc_fetch_size.csv $P/r03_bench_pmc_write_size.csv 1073741824 "r03 bench.py --steps 5, 1024^3 M1" | tail -1
