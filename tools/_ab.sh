for v in 32 64 128 256; do echo TYW $v; PB3D_TYW=$v timeout -k 10 120 python tools/opbench.py --ops M4 2>&1 | tail -2; done
