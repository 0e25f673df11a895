#!/bin/bash
# scripts/ab_run.sh name1 name2 ... : scan kernel time of the default library and of each A/B variant, twice, same box
cd "$(dirname "$0")/.."
one() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['roofline']['kernel_ms'],3), round(d['ms_per_step'],3), d['config']['fallback_rows'])"; }
for rep in 1 2; do
  one base
  for n in "$@"; do MMF_HG_LIBRARY=$PWD/multimodal-fusion_amd/libmmf_hg_$n.so one $n; done
done
