#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the headline kernel on the slab sizes rank 0 carves at N = 1, 2, 4, 8:
# bench.py --slab-planes P under rocprofv3 -> profiles/pmc_traffic.json, so that roofline.traffic of an N > 1 bench line is not null.
# usage: tools/slab_pmc.sh <tag> [size]
tag=$1; S=${2:-1024}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/slabpmc_$tag
mkdir -p $O
for P in $S $((S / 2)) $((S / 4)) $((S / 8)); do
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $O/p${P}_$c -- python3 $R/bench.py --size $S --slab-planes $P --steps 5 --warmup 2 > $O/p${P}_$c.log 2>&1 || { echo "pass $P $c failed"; tail -3 $O/p${P}_$c.log; exit 1; }
    done
    f=$(find $O/p${P}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $O/p${P}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    (cd $R && python3 tools/pmc_summary.py $f $w $((P * S * S)) "$tag bench.py --slab-planes $P --steps 5, ${S}^3 M1 (rank 0's slab at N = $((S / P)))") | cut -c1-260
done
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
