#!/bin/bash
# scan kernel time, v_mfma 16x16x32 form (default) vs 32x32x16 form (MMF_SCAN_SHAPE=32), alternating on one box
cd "$(dirname "$0")/.."
one() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['roofline']['kernel_ms'],3), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['config']['fallback_rows'])"; }
for rep in 1 2; do
  one s16 "$@"
  MMF_SCAN_SHAPE=32 one s32 "$@"
done
