#!/bin/bash
# Copy what tools/r04_evidence.sh left under gpurun_out/<tag>/ (default r04) into profiles/ under the names profiles/README.md lists (run
# here, after the GPU call has merged its outputs back), and re-make the 1024-plane record of profiles/pmc_traffic.json from that run's PMC passes.
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r04}
O=gpurun_out/$TAG; P=profiles
newest() { ls -t $1 2>/dev/null | head -1; }
cp $O/gputests.log $P/r04_gpu_test_suite.log
cp $O/bench.json $P/r04_bench.json
cp "$(newest "$O/bench_stats/runc/*_kernel_stats.csv")" $P/r04_bench_kernel_stats.csv
cp "$(newest "$O/bench_fetch/runc/*_counter_collection.csv")" $P/r04_bench_pmc_fetch_size.csv
cp "$(newest "$O/bench_write/runc/*_counter_collection.csv")" $P/r04_bench_pmc_write_size.csv
cp $O/opbench.jsonl $P/r04_opbench_all_ops.jsonl
cp $O/slicedbench.jsonl $P/r04_slicedbench.jsonl
cp $O/gcbench.jsonl $P/r04_gcbench.jsonl
cp $O/shapebench.jsonl $P/r04_shapebench_real_shapes.jsonl
cp $O/tybench_part.jsonl $P/r04_tybench_part_carve.jsonl
cp $O/notebook1.json $P/r04_notebook1_taj512_host_api.json; cp $O/nb1prof.txt $P/r04_notebook1_taj512_kernel_stats.txt
cp $O/cclbench.jsonl $P/r04_cclbench.jsonl
# (profiles/pmc_traffic.json: copied from gpurun_out/slabpmc_<tag>/pmc_traffic.json, made on the GPU box by tools/slab_pmc.sh with the tree that ran)
cp gpurun_out/slabpmc_$TAG/pmc_traffic.json $P/pmc_traffic.json
cp $O/slicedprof.txt $P/r04_sliced_chain_kernel_stats_and_pmc.txt
