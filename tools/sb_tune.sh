#!/bin/bash
# development: shapebench lines under tuning knobs.  usage: tools/sb_tune.sh "<tune1>" "<tune2>" ...
for t in "$@"; do
  echo "== tune [$t]"
  timeout -k 10 120 python tools/shapebench.py --shapes 512x278x512,512x512x512,256x139x256 --tune "$t" 2>&1 | grep -E "process_voxel_grid|part_carve" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['op'], d['ms'], d['alg_GB_s'])
" || exit 1
  timeout -k 10 200 python tools/opbench.py --ops M3 --tune "$t" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['op'], d['name'][:50], d['ms'], d.get('frac_of_8TBs'))
" || exit 1
done
