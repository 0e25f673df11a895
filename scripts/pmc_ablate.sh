#!/bin/bash
# PMC of the instrumented scan under a timing-only ablation: scripts/pmc_ablate.sh <MMF_SCAN_DEBUG bits> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bits=$1; tag=$2
export MMF_SCAN_DEBUG=$bits
mkdir -p gpurun_out/pmc_$tag
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_$tag -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bf16-leg > gpurun_out/pmc_$tag/log.txt 2>&1
echo "rc=$?"
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = 0
for f in glob.glob('gpurun_out/pmc_$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'scan_b16x' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
print('$tag', dict(acc))
g = acc.get('GRBM_GUI_ACTIVE', 0) / 8.0
if g:
    print('$tag mfma busy / (1024 SIMDs * cycles) = %.3f   wait_any/wave_cycles = %.3f  active/wave_cycles = %.3f' % (
        acc['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * g), acc['SQ_WAIT_ANY'] / acc['SQ_WAVE_CYCLES'], acc['SQ_ACTIVE_INST_ANY'] / acc['SQ_WAVE_CYCLES']), 'cycles', g)
PY
grep -o '"kernel_ms": [0-9.]*' gpurun_out/pmc_$tag/log.txt | tail -1
