#!/bin/bash
# development: shapebench lines under tuning knobs.  usage: tools/sb_tune.sh "<tune1>" "<tune2>" ...
for t in "$@"; do
  echo "== tune [$t]"
  timeout -k 10 120 python tools/shapebench.py --shapes 355x512x355,437x512x437,512x278x512 --tune "$t" 2>&1 | grep -E "part_carve\(6" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['op'], d['ms'], d['alg_GB_s'])
" || exit 1
done
