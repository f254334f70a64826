#!/bin/bash
for sz in 512x512x512 355x512x355 437x512x437 512x278x512 256x139x256; do
  echo "== $sz split on"; timeout -k 10 100 python tools/m4bench.py --size $sz --angles 45 --variants 256 --reps 20 --no-check | cut -c1-150 || exit 1
  echo "== $sz split off"; PB3D_TUNE5=4 timeout -k 10 100 python tools/m4bench.py --size $sz --angles 45 --variants 256 --reps 20 --no-check | cut -c1-150 || exit 1
done
